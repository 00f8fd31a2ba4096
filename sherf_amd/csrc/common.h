// Shared host/device helpers for libsherf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sherf_hip.h"

extern char g_sherf_err[256];
extern int g_sherf_debug;   // ablation switches for profiling (sherf_set_debug); 0 in production
// Scheduling experiments of tools/frame_ab.py --exps (environment SHERF_EXPERIMENT, read per frame; unset in production): launch ORDER / stream
// placement only, never arithmetic.  bit 0: level-0 rows scattered before the level builds are queued; bit 1: level builds on the encoder's own
// stream; bit 2: cross-stream events created with hipEventReleaseToDevice (read once, at the first frame); bit 3: encoder queued before the ray side; bit 6: the backward's general tall GEMM kernel instead of the streaming one; bit 7: round 2's weight-gradient GEMM kernel; bit 8: round 3's tap scatter (binned by the coarsest cell); bit 9: the gather walks its voxel corners one at a time as in round 5 instead of requesting the next corner's rows ahead (gather_tokens_h8_kernel<.., PF>; same bits); bit 10: the eight-channel-per-lane gather instead of the sixteen-channel one (same bits); bit 11: the compaction with one lane per ray and whole waves for the hit rays only (same records; measured slower); bit 12: the two 96-column single-product sparse convolutions as three 32-column workgroups per row tile (sconv3_kernel: CS; same bits; measured slower); bit 13: the compaction with sixteen lanes per ray, four rays per wave (same records; no gain); bit 14: the list search one pipeline stage deeper (cand_search_lists2_kernel; same results; the frame measured slower), bit 15 with it: that kernel held to 80 registers.  Bits 16-19: the warp kernel as that many persistent workgroups per CU (0: one workgroup per 256 samples; measured slower).  Bit 20: the compositing walks a ray's samples one per trip as in rounds 1-5 instead of in batches of four (same arithmetic in the same order).
#include <stdlib.h>
// bit 4: host clock (us, CLOCK_MONOTONIC) of the frame driver's enqueue points on stderr
#include <time.h>
static inline double sherf_host_us() { timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); return t_.tv_sec * 1e6 + t_.tv_nsec * 1e-3; }
#define SHERF_HOST_STAMP(xp_, what_) do { if ((xp_) & 16) fprintf(stderr, "[host] %.1f %s\n", sherf_host_us(), what_); } while (0)
// (same bit: a host function queued IN the stream prints the host clock when the GPU gets there)
static void sherf_gpu_stamp_fn(void* what_) { fprintf(stderr, "[host] %.1f gpu: %s\n", sherf_host_us(), (const char*)what_); }
#define SHERF_GPU_STAMP(xp_, strm_, what_) do { if ((xp_) & 16) (void)hipLaunchHostFunc(strm_, sherf_gpu_stamp_fn, (void*)what_); } while (0)
// SHERF_FRAME_GRAPH_DEBUG=1: while a frame is being captured into a hipGraph (csrc/frame.hip) the enqueue points are traced on stderr
extern int g_sherf_cap_trace;
#define SHERF_CAP_TRACE(what_) do { if (g_sherf_cap_trace) { fprintf(stderr, "[sherf] capture: %s\n", what_); fflush(stderr); } } while (0)
static inline int sherf_experiment() { const char* s_ = getenv("SHERF_EXPERIMENT"); return s_ ? atoi(s_) : 0; }

#define SHERF_CHECK_ARG(cond)                                                                     \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            snprintf(g_sherf_err, sizeof(g_sherf_err), "%s: bad argument: %s", __func__, #cond);  \
            return SHERF_EINVAL;                                                                  \
        }                                                                                         \
    } while (0)

#define SHERF_LAUNCH_CHECK()                                                                      \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) {                                                                   \
            snprintf(g_sherf_err, sizeof(g_sherf_err), "%s: launch failed: %s", __func__,         \
                     hipGetErrorString(e_));                                                      \
            return SHERF_ELAUNCH;                                                                 \
        }                                                                                         \
        return SHERF_OK;                                                                          \
    } while (0)

#define SHERF_HIP_CHECK(expr)                                                                     \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            snprintf(g_sherf_err, sizeof(g_sherf_err), "%s: %s: %s", __func__, #expr,             \
                     hipGetErrorString(e_));                                                      \
            return SHERF_ELAUNCH;                                                                 \
        }                                                                                         \
    } while (0)
#define SHERF_RUN(x) do { const int rc_ = (x); if (rc_ != SHERF_OK) return rc_; } while (0)

static inline hipStream_t as_stream(sherf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int n_cus() {             // compute units of the current device (256 on an MI355X); persistent kernels size their grids by it
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n;
}

// grid header layout (12 x 32-bit): origin.xyz (f32), cell (f32), inv_cell (f32), nx, ny, nz (i32), sub (i32: the near
// mask has sub x sub x sub bits per cell), 3 unused
constexpr int kGridHdr = 12;
struct CellGrid {
    float ox, oy, oz, cell, inv_cell;
    int nx, ny, nz, sub;
};

__device__ __forceinline__ CellGrid load_grid(const float* hdr) {
    CellGrid g;
    g.ox = hdr[0]; g.oy = hdr[1]; g.oz = hdr[2]; g.cell = hdr[3]; g.inv_cell = hdr[4];
    g.nx = __float_as_int(hdr[5]); g.ny = __float_as_int(hdr[6]); g.nz = __float_as_int(hdr[7]);
    g.sub = __float_as_int(hdr[8]);
    return g;
}

// Exact fp32 squared distance, evaluated as ((dx*dx)+(dy*dy))+(dz*dz) with NO fma contraction: the same
// expression the oracle uses for the role of pytorch3d.knn_points (renderer.py:315).
__device__ __forceinline__ float dist2_exact(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// sin / cos of an angle `a` [rad] with the range reduction done right: the phase in revolutions is a two-term product with
// 1 / (2 pi) = kHi + kLo (the fma recovers the rounding error of a * kHi exactly), v_fract keeps its fraction, v_sin / v_cos take
// revolutions.  Phase error < 1e-8 revolutions for |a| <= 2^8, against 6e-8 * |a| / (2 pi) of a plain a * (1 / 2 pi).
// Shared by the network kernel (csrc/mlp.hip: pe_frags) and the gather (csrc/gather.hip: the encodings as ready-made MFMA operand fragments,
// round 6) -- ONE definition, so that the two produce the same bits.  SHERF_MLP_FASTMATH=0 (a profiling build of mlp.hip): libm's sincosf.
#ifndef SHERF_MLP_FASTMATH
#define SHERF_MLP_FASTMATH 1
#endif
__device__ __forceinline__ void sherf_sincos_exact_phase(float a, float* s, float* c) {
#if SHERF_MLP_FASTMATH
    const float kHi = 0.15915494f, kLo = 6.4206383e-09f;
    const float p = a * kHi;
    const float e = __builtin_fmaf(a, kHi, -p);
    const float r = __builtin_amdgcn_fractf(p) + __builtin_fmaf(a, kLo, e);
    *s = __builtin_amdgcn_sinf(r); *c = __builtin_amdgcn_cosf(r);
#else
    sincosf(a, s, c);
#endif
}

// Examine every vertex whose cell overlaps the ball (x, r); keep the lexicographic minimum of (d2, id).
// All vertices with distance <= r are guaranteed to be examined (cell range widened by a safety margin).
__device__ __forceinline__ void nn_search(const CellGrid& g, const int32_t* __restrict__ cell_start,
                                          const float4* __restrict__ pts, float x, float y, float z, float r,
                                          float& best_d2, int& best_id) {
    float rr = r + 1e-4f * g.cell;
    int x0 = (int)floorf((x - rr - g.ox) * g.inv_cell), x1 = (int)floorf((x + rr - g.ox) * g.inv_cell);
    int y0 = (int)floorf((y - rr - g.oy) * g.inv_cell), y1 = (int)floorf((y + rr - g.oy) * g.inv_cell);
    int z0 = (int)floorf((z - rr - g.oz) * g.inv_cell), z1 = (int)floorf((z + rr - g.oz) * g.inv_cell);
    x0 = max(x0, 0); y0 = max(y0, 0); z0 = max(z0, 0);
    x1 = min(x1, g.nx - 1); y1 = min(y1, g.ny - 1); z1 = min(z1, g.nz - 1);
    if (x0 > x1 || y0 > y1 || z0 > z1) return;
    for (int cz = z0; cz <= z1; ++cz)
        for (int cy = y0; cy <= y1; ++cy) {
            int row = (cz * g.ny + cy) * g.nx;
            int s = cell_start[row + x0], e = cell_start[row + x1 + 1];   // cells along x are contiguous
            for (int i = s; i < e; ++i) {
                float4 p = pts[i];
                float d2 = dist2_exact(x, y, z, p.x, p.y, p.z);
                int id = __float_as_int(p.w);
                if (d2 < best_d2 || (d2 == best_d2 && id < best_id)) { best_d2 = d2; best_id = id; }
            }
        }
}

// The same search with the memory accesses batched (one thread per query, nothing else to hide a load behind): when the ball
// reaches at most 3 x 3 rows of cells (always, for a radius <= the cell size) the 18 row bounds are fetched together, then the points of
// the concatenated segments four at a time.  Same candidate set and the same lexicographic minimum as nn_search -- the result does
// not depend on the order of examination.  A wider ball takes the plain loops.
// SHERF_NN_ROWS_VARIANT (build-time, A/B libraries: tools/build_variants.sh nnrows): 2 = the row segments one after the other (round 6 experiment, slower)
#ifndef SHERF_NN_ROWS_VARIANT
#define SHERF_NN_ROWS_VARIANT 0
#endif
constexpr int sherf_nn_rows_variant = SHERF_NN_ROWS_VARIANT;
__device__ __forceinline__ void nn_search_batched(const CellGrid& g, const int32_t* __restrict__ cell_start,
                                                  const float4* __restrict__ pts, float x, float y, float z, float r,
                                                  float& best_d2, int& best_id) {
    const float rr = r + 1e-4f * g.cell;
    int x0 = (int)floorf((x - rr - g.ox) * g.inv_cell), x1 = (int)floorf((x + rr - g.ox) * g.inv_cell);
    int y0 = (int)floorf((y - rr - g.oy) * g.inv_cell), y1 = (int)floorf((y + rr - g.oy) * g.inv_cell);
    int z0 = (int)floorf((z - rr - g.oz) * g.inv_cell), z1 = (int)floorf((z + rr - g.oz) * g.inv_cell);
    x0 = max(x0, 0); y0 = max(y0, 0); z0 = max(z0, 0);
    x1 = min(x1, g.nx - 1); y1 = min(y1, g.ny - 1); z1 = min(z1, g.nz - 1);
    if (x0 > x1 || y0 > y1 || z0 > z1) return;
    if (y1 - y0 > 2 || z1 - z0 > 2) { nn_search(g, cell_start, pts, x, y, z, r, best_d2, best_id); return; }
    int s[9], cum[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int cz = z0 + i / 3, cy = y0 + i % 3;
        const bool ok = cz <= z1 && cy <= y1;
        const int row = ok ? (cz * g.ny + cy) * g.nx : 0;
        s[i] = cell_start[row + x0];
        cum[i] = ok ? cell_start[row + x1 + 1] : s[i];          // (end of the segment for now)
    }
    if (sherf_nn_rows_variant == 2) {
        // Round 6 EXPERIMENT (build with -DSHERF_NN_ROWS_VARIANT=2; MEASURED SLOWER, profiles/r06_call_p_*): the (up to) nine row segments walked ONE AFTER THE
        // OTHER, four points per step.  The loop below indexes the concatenation of the segments and finds point t's address with an eight-way select chain -- ~24 of
        // its ~37 VALU per point -- and here a point costs its distance and its comparison; but a wave then runs the SUM over the nine rows of its lanes' longest
        // segment instead of the longest concatenation: warp_geom 150 -> 176 us, the frame +30 us.  Same points, order-free minimum: identical results.
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            for (int p = s[i]; p < cum[i]; p += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = pts[p + u < cum[i] ? p + u : s[i]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d2 = dist2_exact(x, y, z, v[u].x, v[u].y, v[u].z);
                    const int id = __float_as_int(v[u].w);
                    if (p + u < cum[i] && (d2 < best_d2 || (d2 == best_d2 && id < best_id))) { best_d2 = d2; best_id = id; }
                }
            }
        }
        return;
    }
    int n = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { n += cum[i] - s[i]; cum[i] = n; }
    for (int base = 0; base < n; base += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = base + u;
            int p = s[0] + t;
#pragma unroll
            for (int i = 1; i < 9; ++i) p = (t >= cum[i - 1]) ? s[i] + (t - cum[i - 1]) : p;
            v[u] = pts[t < n ? p : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d2 = dist2_exact(x, y, z, v[u].x, v[u].y, v[u].z);
            const int id = __float_as_int(v[u].w);
            if (base + u < n && (d2 < best_d2 || (d2 == best_d2 && id < best_id))) { best_d2 = d2; best_id = id; }
        }
    }
}

// order-preserving float <-> int map for atomicMin/atomicMax on floats of any sign
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// A frame's compact tiles cut into `nparts` contiguous parts (whole groups of 8 tiles = 256 samples: an MLP workgroup's four tiles and the
// gather's tile pairs never straddle a boundary), so that the gather of part k + 1 and the MLP of part k can be in flight together on
// two streams (csrc/frame.hip).  nparts <= 1: the whole range.
__device__ __forceinline__ void sherf_part_range(int64_t n_tiles, int part, int nparts, int64_t& lo, int64_t& hi) {
    if (nparts <= 1) { lo = 0; hi = n_tiles; return; }
    const int64_t units = (n_tiles + 7) / 8;
    lo = units * part / nparts * 8;
    hi = units * (part + 1) / nparts * 8;
    if (hi > n_tiles) hi = n_tiles;
    if (lo > hi) lo = hi;
}
