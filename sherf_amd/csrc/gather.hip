// Hierarchical feature gather (gfx950): rows a10 (pixel-aligned), a11 (voxel trilinear) and a12 (tri-plane)
// of SURVEY.md section 8, fused into one pass that emits the transformer's input tokens.
//
// Interpolation is linear, so the 1x1 convolutions the reference applies AFTER sampling
// (conv1d_projection 192->96, renderer.py:350; conv1d_reprojection 96->32 per slot, renderer.py:423-424) are
// applied ONCE per frame to the tables instead (sherf_amd/renderer.py: fold_tables): the taps then sum
// straight into the three 32-d slot tokens.  Only the rgb positional encoding (non-linear) is left for the
// MLP kernel.  Tables are channel-last fp32 so every tap is one 16-byte load per lane: 8 lanes own the
// 8 channel quads of a slot, 8 samples per wave, 32 samples (= one MFMA column tile) per workgroup.
#include "common.h"

#ifndef SHERF_GATHER_BRANCHLESS
#define SHERF_GATHER_BRANCHLESS 0
#endif

namespace {

struct Levels { sherf_vox_level l[3]; };

__device__ __forceinline__ void axpy4(float4& a, float w, const float4 v) {
    a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// one channel quad of a table: 16 bytes of fp32, or (HT) 8 bytes of fp16 widened in registers -- `idx` counts quads either way
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
template <bool HT>
__device__ __forceinline__ float4 ldq(const void* __restrict__ base, size_t idx) {
    if constexpr (HT) {
        const h16x4 v = reinterpret_cast<const h16x4*>(base)[idx];
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    } else {
        return reinterpret_cast<const float4*>(base)[idx];
    }
}

// BL: voxel-row loads issued unconditionally (see phase 2 below); same taps, same sums -- a launch-time variant
// (`mode | 4` of sherf_gather_tokens) so that it can be timed against the branching form on the device.
// MINW: waves per SIMD the kernel is compiled for (register cap 512 / MINW): the unconditional form wants 160 VGPRs (3 waves / SIMD);
// `mode | 12` asks for it squeezed into 128 (4 waves / SIMD) -- more loads per wave AND more waves, if the compiler finds the registers.
// HT: fp16 tables (sherf_fold_tables(out_half) / the encoder's half fold rows): the kernel is bound by the bytes it pulls through L2
// (74 % of its wave cycles wait, profiles/r03_pmc_mlp_gather_sampler.txt), so halving them is its speed-up; same taps, fp32 sums.
template <bool BL, int MINW, bool HT>
__global__ void __launch_bounds__(256, MINW) gather_tokens_kernel(const int32_t* __restrict__ counters, const float* __restrict__ geom,
                                                            const void* __restrict__ planes_f, int P,
                                                            const void* __restrict__ feat_f, int Hf, int Wf,
                                                            const float4* __restrict__ img4, int H, int W, Levels lv,
                                                            const float4* __restrict__ tok_bias, const float* __restrict__ bounds,
                                                            const float* __restrict__ vox_min, int3 vox_sh, int64_t capacity,
                                                            float4* __restrict__ tokens, float* __restrict__ extras, int dbg, int mode) {
    const int64_t nv = min((int64_t)counters[0], capacity);
    int64_t t_lo, n_tiles;                          // this launch's part of the tiles (mode bits 8-15: part, 16-23: number of parts)
    sherf_part_range((nv + 31) / 32, (mode >> 8) & 255, (mode >> 16) & 255, t_lo, n_tiles);
    n_tiles -= t_lo;
    mode &= 255;
    const int l = threadIdx.x & 7;                 // channel quad within a slot
    const int j = threadIdx.x >> 3;                // sample within the tile (0..31)
    // XCD-banded tile order (dbg bit 10 turns it off): workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), each with its own 4 MiB L2,
    // and the tables (28-60 MB) fit none of them.  The compact samples are ray-major, so a CONTIGUOUS range of tiles is a band of image
    // rows = a slab of the body; giving XCD k the k-th eighth of the tiles (instead of every eighth tile) makes each L2 serve one slab
    // of the feature map, of the voxel rows and of the (x, y) / (z, y) planes instead of all of them.
    const bool banded = !(dbg & 1024) && gridDim.x % 8 == 0;
    const int64_t per_xcd = (n_tiles + 7) / 8, slots = gridDim.x / 8;
    for (int64_t it = banded ? blockIdx.x / 8 : blockIdx.x; it < (banded ? per_xcd : n_tiles); it += banded ? slots : gridDim.x) {
        int64_t tile = banded ? (int64_t)(blockIdx.x % 8) * per_xcd + it : it;
        if (tile >= n_tiles) break;
        tile += t_lo;
        const int64_t c = tile * 32 + j;
        float4 acc[3];
        if (mode == 2) acc[0] = acc[1] = acc[2] = make_float4(0.f, 0.f, 0.f, 0.f);     // voxel pass adds onto the stored tokens
        else { acc[0] = tok_bias[l]; acc[1] = tok_bias[8 + l]; acc[2] = tok_bias[16 + l]; }
        // extras[tile][12][j]: rows 0-5 = x_c, v_c (copied straight from geom by lanes 0-5), 6-8 = tapped rgb, 9-11 = 0.
        // They are stored as soon as they are known: carried to the end of the tile they cost 9 VGPRs and the kernel
        // one wave of occupancy (100 -> <= 96 VGPRs: 4 -> 5 waves/SIMD).
        float4 rgb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nv) {
            const float* gm = geom + c * 8;
            const float xc[3] = {gm[0], gm[1], gm[2]};
            if (mode != 2 && l < 6) extras[(tile * 12 + l) * 32 + j] = gm[l];
            // ---- tri-plane: renderer.py:234-243, align_corners=False, zeros padding ----
            float n[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) n[a] = 2.f * (xc[a] - bounds[a]) / (bounds[3 + a] - bounds[a]) - 1.f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if ((dbg & 8) || mode == 2) break;
                const float ga = p == 2 ? n[2] : n[0];                 // planes (x,y), (x,z), (z,y)
                const float gb = p == 1 ? n[2] : n[1];
                float px = clampf(((ga + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
                float py = clampf(((gb + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
                float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
                int xi = (int)x0, yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < P && yy >= 0 && yy < P) {
                            float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                            axpy4(acc[p], w, ldq<HT>(planes_f, ((size_t)(p * P + yy) * P + xx) * 8 + l));
                        }
                    }
            }
            // ---- pixel-aligned feature + rgb: renderer.py:330-336, align_corners=True ----
            if (!(dbg & 16) && mode != 2) {
                float gx = 2.0f * gm[6] / (float)W - 1.0f, gy = 2.0f * gm[7] / (float)H - 1.0f;
                float px = clampf((gx + 1.f) * 0.5f * (Wf - 1), -2.f, (float)Wf + 1.f);
                float py = clampf((gy + 1.f) * 0.5f * (Hf - 1), -2.f, (float)Hf + 1.f);
                float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
                int xi = (int)x0, yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < Wf && yy >= 0 && yy < Hf) {
                            float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                            const size_t t = ((size_t)yy * Wf + xx) * 16;
                            axpy4(acc[0], w, ldq<HT>(feat_f, t + l));
                            axpy4(acc[1], w, ldq<HT>(feat_f, t + 8 + l));
                        }
                    }
                px = clampf((gx + 1.f) * 0.5f * (W - 1), -2.f, (float)W + 1.f);
                py = clampf((gy + 1.f) * 0.5f * (H - 1), -2.f, (float)H + 1.f);
                x0 = floorf(px); y0 = floorf(py); fx = px - x0; fy = py - y0;
                xi = (int)x0; yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                            float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                            axpy4(rgb, w, img4[(size_t)yy * W + xx]);
                        }
                    }
            }
            if (mode != 2) {
                if (l == 6) extras[(tile * 12 + 6) * 32 + j] = rgb.x;
                if (l == 7) extras[(tile * 12 + 7) * 32 + j] = rgb.y;
                if (l < 4) extras[(tile * 12 + 8 + l) * 32 + j] = (l == 0) ? rgb.z : 0.f;
            }
            // ---- sparse voxel levels: renderer.py:544-556 + 762-782, align_corners=True ----
            if (!(dbg & 4) && mode != 1) {
                float gz = ((xc[2] - vox_min[2]) / 0.005f) / (float)vox_sh.x * 2.f - 1.f;   // vox_sh = (D,H,W) = (z,y,x)
                float gy = ((xc[1] - vox_min[1]) / 0.005f) / (float)vox_sh.y * 2.f - 1.f;
                float gx = ((xc[0] - vox_min[0]) / 0.005f) / (float)vox_sh.z * 2.f - 1.f;
#pragma unroll 1
                for (int L = 0; L < 3; ++L) {      // not unrolled: keeps the register count (occupancy) down
                    const sherf_vox_level& lev = lv.l[L];
                    float px = clampf((gx + 1.f) * 0.5f * (lev.W - 1), -2.f, (float)lev.W + 1.f);
                    float py = clampf((gy + 1.f) * 0.5f * (lev.H - 1), -2.f, (float)lev.H + 1.f);
                    float pz = clampf((gz + 1.f) * 0.5f * (lev.D - 1), -2.f, (float)lev.D + 1.f);
                    float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
                    float fx = px - x0, fy = py - y0, fz = pz - z0;
                    int xi = (int)x0, yi = (int)y0, zi = (int)z0;
                    // phase 1: the 8 occupancy records, issued back to back (independent loads)
                    uint2 rec[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int xx = xi + (t & 1), yy = yi + ((t >> 1) & 1), zz = zi + (t >> 2);
                        const bool inb = xx >= 0 && xx < lev.W && yy >= 0 && yy < lev.H && zz >= 0 && zz < lev.D;
                        const int key = inb ? (zz * lev.H + yy) * lev.W + xx : 0;
                        const uint2 rr = reinterpret_cast<const uint2*>(lev.wp)[key >> 5];
                        const uint32_t bit = 1u << (key & 31);
                        rec[t] = make_uint2((inb && (rr.x & bit)) ? 1u : 0u, rr.y + __popc(rr.x & (bit - 1u)));
                    }
                    // phase 2: rows of the occupied corners
                    if constexpr (BL) {
                    // variant to be measured: every corner's three row loads issued unconditionally (absent corners read row 0
                    // with weight 0) so that all 24 loads of a level are in flight together instead of one divergent branch at
                    // a time -- trades cached redundant loads for memory-level parallelism in a kernel that is 74 % wait
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float w = rec[t].x ? ((t & 1) ? fx : 1.f - fx) * (((t >> 1) & 1) ? fy : 1.f - fy) * ((t >> 2) ? fz : 1.f - fz) : 0.f;
                        const size_t r = (size_t)(rec[t].x ? rec[t].y : 0u) * 24;
                        axpy4(acc[0], w, ldq<HT>(lev.rows, r + l));
                        axpy4(acc[1], w, ldq<HT>(lev.rows, r + 8 + l));
                        axpy4(acc[2], w, ldq<HT>(lev.rows, r + 16 + l));
                    }
                    } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (rec[t].x) {
                            const float w = ((t & 1) ? fx : 1.f - fx) * (((t >> 1) & 1) ? fy : 1.f - fy) * ((t >> 2) ? fz : 1.f - fz);
                            const size_t r = (size_t)rec[t].y * 24;
                            axpy4(acc[0], w, ldq<HT>(lev.rows, r + l));
                            axpy4(acc[1], w, ldq<HT>(lev.rows, r + 8 + l));
                            axpy4(acc[2], w, ldq<HT>(lev.rows, r + 16 + l));
                        }
                    }
                    }
                }
            }
        } else {
            acc[0] = acc[1] = acc[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mode != 2) {                               // padding columns of the last tile
                extras[(tile * 12 + l) * 32 + j] = 0.f;
                if (l < 4) extras[(tile * 12 + 8 + l) * 32 + j] = 0.f;
            }
        }
        // tokens[tile][slot][quad][j] (float4), extras[tile][12][j]
        if (mode == 2) {
            if (c < nv)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    float4 t = tokens[((tile * 3 + s) * 8 + l) * 32 + j];
                    t.x += acc[s].x; t.y += acc[s].y; t.z += acc[s].z; t.w += acc[s].w;
                    tokens[((tile * 3 + s) * 8 + l) * 32 + j] = t;
                }
            continue;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) tokens[((tile * 3 + s) * 8 + l) * 32 + j] = acc[s];
    }
}

// fp16 tables, EIGHT channels per lane (round 3).  The counters of the kernel above (profiles/r03_pmc_gather_ta.txt) read: texture
// addresser busy 85 % of the launch, 1.44e8 cache accesses -- the SAME with fp32 and with fp16 tables, same 370 us.  The addresser
// takes a wave's load four lanes per cycle whether a lane asks for 8 or for 16 bytes: the kernel's time is its number of wave-level
// load instructions x 16 cycles, not its bytes.  So with fp16 tables a lane takes 16 bytes = EIGHT channels, FOUR lanes own a
// 32-channel slot and a wave covers 16 samples instead of 8: half the load instructions for the same taps.  Same stencils, same
// weights, fp32 sums; the tokens / extras come out in the same tile-major layout (a lane stores two quads).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int kPeStride = 120;          // halfs per sample of the encodings' LDS exchange: 112 + 8 (240 B: 16-byte aligned, 16 samples on 16 bank groups)
// the first pair of tiles workgroup `b` of `g` takes (the loop header of gather_tokens_h8_kernel): n_pairs or more = none
__device__ __forceinline__ int64_t banded_first_pair(int b, int g, int64_t n_pairs, int dbg) {
    const bool banded = !(dbg & 1024) && g % 8 == 0;
    if (!banded) return b;
    const int64_t per_xcd = (n_pairs + 7) / 8;
    return (int64_t)(b / 8) < per_xcd ? (int64_t)(b % 8) * per_xcd + b / 8 : n_pairs;
}
struct f8 { float4 a, b; };
// (idx8: a 32-bit index of 16-byte units -- the tables of this kernel are far below 4 GiB (sherf_gather_tokens checks), and a uniform base + a 32-bit lane
//  offset is ONE address instruction where a 64-bit index is a multiply-add pair per load: round 6, this kernel turned out to be bound by its VALU count)
__device__ __forceinline__ void axpy8(f8& acc, float w, const void* __restrict__ base, uint32_t idx8) {
    const h16x8 v = *reinterpret_cast<const h16x8*>(static_cast<const char*>(base) + (size_t)(idx8 * 16u));
    acc.a.x += w * (float)v[0]; acc.a.y += w * (float)v[1]; acc.a.z += w * (float)v[2]; acc.a.w += w * (float)v[3];
    acc.b.x += w * (float)v[4]; acc.b.y += w * (float)v[5]; acc.b.z += w * (float)v[6]; acc.b.w += w * (float)v[7];
}

// STAGE (round 5): the (word, prefix) occupancy records of the two COARSER tapped levels (2.3 K + 0.3 K words at the
// bench subject's grid: 20 KiB) are copied to LDS once per workgroup, and a workgroup then walks several pairs of tiles -- 16 of a sample's 24 first-hop
// look-ups become LDS reads.  The counters of the round (profiles/r05_call_g_pmc_*) say the kernel waits on dependent gathers (waves parked 65 % of their
// cycles, addresser 62 % busy): this shortens two of its three look-up -> row chains.  Same arithmetic: bit-identical tokens.
// PE (round 6): the network kernel is bound by the board's power cap, this one by the latency of dependent gathers (its
// VALU sits idle 85 % of the time) -- so the positional encodings PE6(x_c), PE4(v_c), PE5(rgb) (renderer.py:875-916, 423) are evaluated HERE, from
// the values this kernel holds anyway, and handed to sherf_nerf_mlp3_pe as ready-made fp16 MFMA B-operand fragments:
//     pefrag[tile][q][lane] (16 bytes: features 16 kb + 8 h .. + 7 of column j, lane = 32 h + j),  q = 0-2 PE6 kb, 3-4 PE4 kb, 5-6 PE5 kb,
// in the SAME natural feature order, from the SAME sin / cos sequence (common.h) and rounding (nearest even) the kernel's own pe_frags<2> uses:
// bit-identical operands.  Lane l < 3 of a sample's quad evaluates axis l of the three vectors (15 sin / cos pairs instead of the 63 every lane
// of the network kernel computes, twice for PE6), the quad transposes through 240 bytes of LDS per sample into 8-feature groups, lane l stores
// groups l, l + 4, l + 8, l + 12.  224 bytes per sample more to write, ~180 VALU per 16 samples.
// PF (round 6): the voxel rows of corner t + 1 are requested BEFORE corner t's 24 multiply-adds instead of after them.  The kernel's time is its chain of dependent
// load -> wait -> use rounds (~50 per 16 samples: 24 voxel corners, 12 plane and 8 pixel corners, the look-ups), each a full L1 / L2 round trip that only the other
// waves of the SIMD cover.  Absent corners are read at row 0 with weight 0 (exact: rows are finite, the buffers start zeroed) so that the loads carry no branch -- a
// load inside `if (present)` is waited for on the spot (round 5's trap) -- and a scheduling barrier per corner keeps the compiler from hoisting all 24 requests at once
// (round 5's unconditional variant: 176 registers, two waves per SIMD, slower).  Same sums in the same order: bit-identical tokens.
template <bool STAGE, bool PE = false, bool PF = false>
__global__ void __launch_bounds__(256) gather_tokens_h8_kernel(const int32_t* __restrict__ counters, const float* __restrict__ geom,
                                                               const void* __restrict__ planes_f, int P, const void* __restrict__ feat_f,
                                                               int Hf, int Wf, const float4* __restrict__ img4, int H, int W, Levels lv,
                                                               const float4* __restrict__ tok_bias, const float* __restrict__ bounds,
                                                               const float* __restrict__ vox_min, int3 vox_sh, int64_t capacity,
                                                               float4* __restrict__ tokens, float* __restrict__ extras, int dbg, int mode,
                                                               uint4* __restrict__ pefrag) {
    const int64_t nv = min((int64_t)counters[0], capacity);
    int64_t t_lo, t_hi;                             // this launch's part of the tiles (mode bits 8-15: part, 16-23: number of parts)
    sherf_part_range((nv + 31) / 32, (mode >> 8) & 255, (mode >> 16) & 255, t_lo, t_hi);
    mode &= 255;
    const int64_t n_tiles = t_hi - t_lo, n_pairs = (n_tiles + 1) / 2;          // a workgroup step = two tiles = 64 samples (t_lo is even)
    extern __shared__ __attribute__((aligned(16))) uint2 s_wp[];              // STAGE: level 1's records, then level 2's
    int s_n1 = 0;                                    // level 2's records start behind level 1's
    if constexpr (STAGE) {
        const int n1 = (lv.l[1].D * lv.l[1].H * lv.l[1].W + 31) / 32, n2 = (lv.l[2].D * lv.l[2].H * lv.l[2].W + 31) / 32;
        s_n1 = n1;
        if ((banded_first_pair(blockIdx.x, gridDim.x, n_pairs, dbg)) < n_pairs && mode != 1) {     // (a workgroup beyond the data stages nothing)
            for (int i = threadIdx.x; i < n1; i += 256) s_wp[i] = reinterpret_cast<const uint2*>(lv.l[1].wp)[i];
            for (int i = threadIdx.x; i < n2; i += 256) s_wp[n1 + i] = reinterpret_cast<const uint2*>(lv.l[2].wp)[i];
        }
        __syncthreads();
    }
    const int l = threadIdx.x & 3;                 // channel octet within a slot (quads 2l, 2l + 1)
    const int js = threadIdx.x >> 2;               // sample within the pair of tiles (0..63)
    const bool banded = !(dbg & 1024) && gridDim.x % 8 == 0;                  // XCD-banded order, as in gather_tokens_kernel
    const int64_t per_xcd = (n_pairs + 7) / 8, slots = gridDim.x / 8;
    for (int64_t it = banded ? blockIdx.x / 8 : blockIdx.x; it < (banded ? per_xcd : n_pairs); it += banded ? slots : gridDim.x) {
        const int64_t pair = banded ? (int64_t)(blockIdx.x % 8) * per_xcd + it : it;
        if (pair >= n_pairs) break;
        int64_t tile = pair * 2 + (js >> 5);
        const int j = js & 31;
        if (tile >= n_tiles) continue;
        tile += t_lo;
        const int64_t c = tile * 32 + j;
        f8 acc[3];
        if (mode == 2) { for (int s_ = 0; s_ < 3; ++s_) acc[s_].a = acc[s_].b = make_float4(0.f, 0.f, 0.f, 0.f); }
        else { for (int s_ = 0; s_ < 3; ++s_) { acc[s_].a = tok_bias[8 * s_ + 2 * l]; acc[s_].b = tok_bias[8 * s_ + 2 * l + 1]; } }
        float4 rgb = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool valid = c < nv;
        float xv[3] = {0.f, 0.f, 0.f};                 // x_c for the voxel taps below (outside the divergent branch)
        if (valid) {
            const float* gm = geom + c * 8;
            const float xc[3] = {gm[0], gm[1], gm[2]};
            xv[0] = xc[0]; xv[1] = xc[1]; xv[2] = xc[2];
            if (mode != 2) {                                                  // extras rows 0-5 = x_c, v_c: lanes 0-2 write two each
                if (l < 3) { extras[(tile * 12 + 2 * l) * 32 + j] = gm[2 * l]; extras[(tile * 12 + 2 * l + 1) * 32 + j] = gm[2 * l + 1]; }
            }
            float n[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) n[a] = 2.f * (xc[a] - bounds[a]) / (bounds[3 + a] - bounds[a]) - 1.f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if ((dbg & 8) || mode == 2) break;
                const float ga = p == 2 ? n[2] : n[0];
                const float gb = p == 1 ? n[2] : n[1];
                float px = clampf(((ga + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
                float py = clampf(((gb + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
                float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
                int xi = (int)x0, yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < P && yy >= 0 && yy < P)
                            axpy8(acc[p], (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy), planes_f, (uint32_t)(((p * P + yy) * P + xx) * 4 + l));
                    }
            }
            if (!(dbg & 16) && mode != 2) {
                float gx = 2.0f * gm[6] / (float)W - 1.0f, gy = 2.0f * gm[7] / (float)H - 1.0f;
                float px = clampf((gx + 1.f) * 0.5f * (Wf - 1), -2.f, (float)Wf + 1.f);
                float py = clampf((gy + 1.f) * 0.5f * (Hf - 1), -2.f, (float)Hf + 1.f);
                float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
                int xi = (int)x0, yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < Wf && yy >= 0 && yy < Hf) {
                            const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                            const uint32_t t = (uint32_t)((yy * Wf + xx) * 8);
                            axpy8(acc[0], w, feat_f, t + l);
                            axpy8(acc[1], w, feat_f, t + 4 + l);
                        }
                    }
                px = clampf((gx + 1.f) * 0.5f * (W - 1), -2.f, (float)W + 1.f);
                py = clampf((gy + 1.f) * 0.5f * (H - 1), -2.f, (float)H + 1.f);
                x0 = floorf(px); y0 = floorf(py); fx = px - x0; fy = py - y0;
                xi = (int)x0; yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < W && yy >= 0 && yy < H) axpy4(rgb, (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy), img4[(size_t)yy * W + xx]);
                    }
            }
            if (mode != 2) {                                                  // rows 6-8 = tapped rgb, 9-11 = 0
                if (l == 3) { extras[(tile * 12 + 6) * 32 + j] = rgb.x; extras[(tile * 12 + 7) * 32 + j] = rgb.y; }
                extras[(tile * 12 + 8 + l) * 32 + j] = (l == 0) ? rgb.z : 0.f;
            }
        } else {
            for (int s_ = 0; s_ < 3; ++s_) acc[s_].a = acc[s_].b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mode != 2) {                               // padding columns of the last tile
                if (l < 3) { extras[(tile * 12 + 2 * l) * 32 + j] = 0.f; extras[(tile * 12 + 2 * l + 1) * 32 + j] = 0.f; }
                if (l == 3) { extras[(tile * 12 + 6) * 32 + j] = 0.f; extras[(tile * 12 + 7) * 32 + j] = 0.f; }
                extras[(tile * 12 + 8 + l) * 32 + j] = 0.f;
            }
        }
        // voxel taps (a11's three tapped levels).  A sample's four lanes need the same eight corner look-ups per level ((word, prefix) of
        // the level's bitmap -> present?, row): lane l makes TWO of them (corners 2l, 2l + 1) and the quad exchanges the results -- a
        // third of this kernel's load instructions were those look-ups repeated four times over (round 4).  Outside the `valid` branch:
        // the exchange is a wave-level operation (invalid lanes look nothing up and add nothing).
        if (!(dbg & 4) && mode != 1) {
            const float gz = ((xv[2] - vox_min[2]) / 0.005f) / (float)vox_sh.x * 2.f - 1.f;
            const float gy = ((xv[1] - vox_min[1]) / 0.005f) / (float)vox_sh.y * 2.f - 1.f;
            const float gx = ((xv[0] - vox_min[0]) / 0.005f) / (float)vox_sh.z * 2.f - 1.f;
            const int quad0 = (threadIdx.x & 63) & ~3;
#pragma unroll 1
            for (int L = 0; L < 3; ++L) {
                const sherf_vox_level& lev = lv.l[L];
                float px = clampf((gx + 1.f) * 0.5f * (lev.W - 1), -2.f, (float)lev.W + 1.f);
                float py = clampf((gy + 1.f) * 0.5f * (lev.H - 1), -2.f, (float)lev.H + 1.f);
                float pz = clampf((gz + 1.f) * 0.5f * (lev.D - 1), -2.f, (float)lev.D + 1.f);
                float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
                float fx = px - x0, fy = py - y0, fz = pz - z0;
                int xi = (int)x0, yi = (int)y0, zi = (int)z0;
                int mine[2];                               // corner 2l + u: row index, or -1 if the voxel is absent
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int xx = xi + u, yy = yi + (l & 1), zz = zi + (l >> 1);
                    const bool inb = valid && xx >= 0 && xx < lev.W && yy >= 0 && yy < lev.H && zz >= 0 && zz < lev.D;
                    const int key = inb ? (zz * lev.H + yy) * lev.W + xx : 0;
                    uint2 rr;
                    if (STAGE && L > 0) rr = s_wp[(L == 2 ? s_n1 : 0) + (key >> 5)];
                    else rr = reinterpret_cast<const uint2*>(lev.wp)[key >> 5];
                    const uint32_t bit = 1u << (key & 31);
                    mine[u] = (inb && (rr.x & bit)) ? (int)(rr.y + __popc(rr.x & (bit - 1u))) : -1;
                }
                if constexpr (PF) {
                    int rows8[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) rows8[t] = __shfl(mine[t & 1], quad0 | (t >> 1));
                    h16x8 buf[2][3];
                    const char* rb = reinterpret_cast<const char*>(lev.rows);
                    auto request = [&](int t, h16x8 (&b)[3]) {
                        const uint32_t r = ((uint32_t)max(rows8[t], 0) * 12u + l) * 16u;
                        b[0] = *reinterpret_cast<const h16x8*>(rb + (size_t)r);
                        b[1] = *reinterpret_cast<const h16x8*>(rb + (size_t)(r + 64u));
                        b[2] = *reinterpret_cast<const h16x8*>(rb + (size_t)(r + 128u));
                    };
                    request(0, buf[0]);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (t < 7) request(t + 1, buf[(t + 1) & 1]);
                        const float w0 = ((t & 1) ? fx : 1.f - fx) * (((t >> 1) & 1) ? fy : 1.f - fy) * ((t >> 2) ? fz : 1.f - fz);
                        const float w = rows8[t] >= 0 ? w0 : 0.f;
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) {
                            const h16x8 v = buf[t & 1][s_];
                            acc[s_].a.x += w * (float)v[0]; acc[s_].a.y += w * (float)v[1]; acc[s_].a.z += w * (float)v[2]; acc[s_].a.w += w * (float)v[3];
                            acc[s_].b.x += w * (float)v[4]; acc[s_].b.y += w * (float)v[5]; acc[s_].b.z += w * (float)v[6]; acc[s_].b.w += w * (float)v[7];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int row = __shfl(mine[t & 1], quad0 | (t >> 1));
                    if (row >= 0) {
                        const float w = ((t & 1) ? fx : 1.f - fx) * (((t >> 1) & 1) ? fy : 1.f - fy) * ((t >> 2) ? fz : 1.f - fz);
                        const uint32_t r = (uint32_t)row * 12u;
                        axpy8(acc[0], w, lev.rows, r + l);
                        axpy8(acc[1], w, lev.rows, r + 4 + l);
                        axpy8(acc[2], w, lev.rows, r + 8 + l);
                    }
                }
                }
            }
        }
        // tokens[tile][slot][quad][j] (float4): this lane owns quads 2l and 2l + 1 of every slot
        if (mode == 2) {
            if (c < nv)
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    float4* ta = tokens + ((tile * 3 + s_) * 8 + 2 * l) * 32 + j;
                    float4 t0 = ta[0], t1 = ta[32];
                    t0.x += acc[s_].a.x; t0.y += acc[s_].a.y; t0.z += acc[s_].a.z; t0.w += acc[s_].a.w;
                    t1.x += acc[s_].b.x; t1.y += acc[s_].b.y; t1.z += acc[s_].b.z; t1.w += acc[s_].b.w;
                    ta[0] = t0; ta[32] = t1;
                }
            continue;
        }
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            tokens[((tile * 3 + s_) * 8 + 2 * l) * 32 + j] = acc[s_].a;
            tokens[((tile * 3 + s_) * 8 + 2 * l + 1) * 32 + j] = acc[s_].b;
        }
        if constexpr (PE) {
            // the encodings of this sample as B-operand fragments (see the kernel's header).  A wave = 16 whole quads of ONE tile: the `continue`s
            // above are wave-uniform, the LDS exchange stays inside a quad, wave_barrier orders it (the wave runs in lockstep)
            __shared__ __attribute__((aligned(16))) _Float16 s_pe[64 * kPeStride];
            _Float16* sp = s_pe + js * kPeStride;              // [PE6: 48 | PE4: 32 | PE5: 32] features, natural order
            float v3[3] = {0.f, 0.f, 0.f};
            if (valid) { const float* gm = geom + c * 8; v3[0] = gm[3]; v3[1] = gm[4]; v3[2] = gm[5]; }
            if (l < 3) {
                const float u6 = l == 0 ? xv[0] : l == 1 ? xv[1] : xv[2];
                const float u4 = l == 0 ? v3[0] : l == 1 ? v3[1] : v3[2];
                const float u5 = l == 0 ? rgb.x : l == 1 ? rgb.y : rgb.z;
                sp[l] = (_Float16)u6; sp[48 + l] = (_Float16)u4; sp[80 + l] = (_Float16)u5;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    float sn, cs;
                    sherf_sincos_exact_phase(u6 * (float)(1 << q), &sn, &cs);
                    sp[3 + 6 * q + l] = (_Float16)sn; sp[6 + 6 * q + l] = (_Float16)cs;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float sn, cs;
                    sherf_sincos_exact_phase(u4 * (float)(1 << q), &sn, &cs);
                    sp[48 + 3 + 6 * q + l] = (_Float16)sn; sp[48 + 6 + 6 * q + l] = (_Float16)cs;
                }
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    float sn, cs;
                    sherf_sincos_exact_phase(u5 * (float)(1 << q), &sn, &cs);
                    sp[80 + 3 + 6 * q + l] = (_Float16)sn;
                    if (q < 4 || l < 2) sp[80 + 6 + 6 * q + l] = (_Float16)cs;        // (feature 32 = the last cosine is not an input of conv1d_reprojection's [:32])
                }
            } else {                                                                   // the zero padding of the K-blocks: PE6 39..47, PE4 27..31
                sp[39] = (_Float16)0.f;
                *reinterpret_cast<uint4*>(sp + 40) = make_uint4(0u, 0u, 0u, 0u);
                sp[48 + 27] = (_Float16)0.f;
                *reinterpret_cast<uint2*>(sp + 48 + 28) = make_uint2(0u, 0u);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = l + 4 * i;                                               // group g = 2 q + h: features 8 g .. 8 g + 7
                if (g < 14) pefrag[((tile * 7 + (g >> 1)) * 2 + (g & 1)) * 32 + j] = *reinterpret_cast<const uint4*>(sp + 8 * g);
            }
            __builtin_amdgcn_wave_barrier();                                           // every group is read before the next step rewrites the slots
        }
    }
}

// SIXTEEN channels per lane (round 6): two lanes own a sample, a wave covers one whole 32-sample tile.  Why: once its taps ran on v_fma_mix (no SLP: sherf_amd/build.py)
// the eight-channel kernel turned out to be bound by its VALU count as much as by its gathers -- ~1 750 instructions per 16 samples of which only 736 are the taps'
// multiply-adds; the rest (a sample's normalised coordinates, three planes' and three levels' stencils: floors, fractions, 56 corner weights, occupancy look-ups,
// eleven IEEE divisions) is the SAME for every lane of a sample and was computed by all four.  With two lanes per sample that arithmetic is done half as often per
// sample; a lane takes 32 contiguous bytes of a row's slot (two 16-byte loads), the same number of load instructions per sample.  Same taps, same weights, same order
// of accumulation per channel: bit-identical tokens / extras.  One pass over every tap (mode 0); the split / part / staged / encodings forms stay on the
// eight-channel kernel.
__global__ void __launch_bounds__(256, 4) gather_tokens_h16_kernel(const int32_t* __restrict__ counters, const float* __restrict__ geom,
                                                                const void* __restrict__ planes_f, int P, const void* __restrict__ feat_f,
                                                                int Hf, int Wf, const float4* __restrict__ img4, int H, int W, Levels lv,
                                                                const float4* __restrict__ tok_bias, const float* __restrict__ bounds,
                                                                const float* __restrict__ vox_min, int3 vox_sh, int64_t capacity,
                                                                float4* __restrict__ tokens, float* __restrict__ extras, int dbg) {
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t n_tiles = (nv + 31) / 32, n_groups = (n_tiles + 3) / 4;       // a workgroup step = four tiles = 128 samples
    const int l = threadIdx.x & 1;                  // half of a slot's 32 channels: octets 2l, 2l + 1 (quads 4l .. 4l + 3)
    const int js = threadIdx.x >> 1;                // sample within the group of four tiles (0..127): a wave = one tile
    const bool banded = !(dbg & 1024) && gridDim.x % 8 == 0;                    // XCD-banded order, as in gather_tokens_h8_kernel
    const int64_t per_xcd = (n_groups + 7) / 8, slots = gridDim.x / 8;
    for (int64_t it = banded ? blockIdx.x / 8 : blockIdx.x; it < (banded ? per_xcd : n_groups); it += banded ? slots : gridDim.x) {
        const int64_t group = banded ? (int64_t)(blockIdx.x % 8) * per_xcd + it : it;
        if (group >= n_groups) break;
        const int64_t tile = group * 4 + (js >> 5);
        const int j = js & 31;
        if (tile >= n_tiles) continue;
        const int64_t c = tile * 32 + j;
        f8 acc[3][2];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int o = 0; o < 2; ++o) { acc[s_][o].a = tok_bias[8 * s_ + 2 * (2 * l + o)]; acc[s_][o].b = tok_bias[8 * s_ + 2 * (2 * l + o) + 1]; }
        float4 rgb = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool valid = c < nv;
        float xv[3] = {0.f, 0.f, 0.f};
        if (valid) {
            const float* gm = geom + c * 8;
            const float xc[3] = {gm[0], gm[1], gm[2]};
            xv[0] = xc[0]; xv[1] = xc[1]; xv[2] = xc[2];
            if (l == 0) {                                                         // extras rows 0-5 = x_c, v_c
#pragma unroll
                for (int r = 0; r < 6; ++r) extras[(tile * 12 + r) * 32 + j] = gm[r];
            }
            float n[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) n[a] = 2.f * (xc[a] - bounds[a]) / (bounds[3 + a] - bounds[a]) - 1.f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (dbg & 8) break;
                const float ga = p == 2 ? n[2] : n[0];
                const float gb = p == 1 ? n[2] : n[1];
                float px = clampf(((ga + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
                float py = clampf(((gb + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
                float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
                int xi = (int)x0, yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < P && yy >= 0 && yy < P) {
                            const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                            const uint32_t t = (uint32_t)(((p * P + yy) * P + xx) * 4 + 2 * l);
                            axpy8(acc[p][0], w, planes_f, t);
                            axpy8(acc[p][1], w, planes_f, t + 1);
                        }
                    }
            }
            if (!(dbg & 16)) {
                float gx = 2.0f * gm[6] / (float)W - 1.0f, gy = 2.0f * gm[7] / (float)H - 1.0f;
                float px = clampf((gx + 1.f) * 0.5f * (Wf - 1), -2.f, (float)Wf + 1.f);
                float py = clampf((gy + 1.f) * 0.5f * (Hf - 1), -2.f, (float)Hf + 1.f);
                float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
                int xi = (int)x0, yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < Wf && yy >= 0 && yy < Hf) {
                            const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                            const uint32_t t = (uint32_t)((yy * Wf + xx) * 8 + 2 * l);
                            axpy8(acc[0][0], w, feat_f, t);
                            axpy8(acc[0][1], w, feat_f, t + 1);
                            axpy8(acc[1][0], w, feat_f, t + 4);
                            axpy8(acc[1][1], w, feat_f, t + 5);
                        }
                    }
                px = clampf((gx + 1.f) * 0.5f * (W - 1), -2.f, (float)W + 1.f);
                py = clampf((gy + 1.f) * 0.5f * (H - 1), -2.f, (float)H + 1.f);
                x0 = floorf(px); y0 = floorf(py); fx = px - x0; fy = py - y0;
                xi = (int)x0; yi = (int)y0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        int xx = xi + dx, yy = yi + dy;
                        if (xx >= 0 && xx < W && yy >= 0 && yy < H) axpy4(rgb, (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy), img4[(size_t)yy * W + xx]);
                    }
            }
            if (l == 1) {                                                         // rows 6-8 = tapped rgb, 9-11 = 0
                extras[(tile * 12 + 6) * 32 + j] = rgb.x; extras[(tile * 12 + 7) * 32 + j] = rgb.y; extras[(tile * 12 + 8) * 32 + j] = rgb.z;
#pragma unroll
                for (int r = 9; r < 12; ++r) extras[(tile * 12 + r) * 32 + j] = 0.f;
            }
        } else {
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) acc[s_][0].a = acc[s_][0].b = acc[s_][1].a = acc[s_][1].b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 6; ++r) extras[(tile * 12 + 6 * l + r) * 32 + j] = 0.f;     // padding columns of the last tile
        }
        // voxel taps: lane l makes FOUR of a level's eight corner look-ups (corners 4l .. 4l + 3 = the z-plane zi + l) and the pair exchanges the results
        if (!(dbg & 4)) {
            const float gz = ((xv[2] - vox_min[2]) / 0.005f) / (float)vox_sh.x * 2.f - 1.f;
            const float gy = ((xv[1] - vox_min[1]) / 0.005f) / (float)vox_sh.y * 2.f - 1.f;
            const float gx = ((xv[0] - vox_min[0]) / 0.005f) / (float)vox_sh.z * 2.f - 1.f;
            const int pair0 = (threadIdx.x & 63) & ~1;
#pragma unroll 1
            for (int L = 0; L < 3; ++L) {
                const sherf_vox_level& lev = lv.l[L];
                float px = clampf((gx + 1.f) * 0.5f * (lev.W - 1), -2.f, (float)lev.W + 1.f);
                float py = clampf((gy + 1.f) * 0.5f * (lev.H - 1), -2.f, (float)lev.H + 1.f);
                float pz = clampf((gz + 1.f) * 0.5f * (lev.D - 1), -2.f, (float)lev.D + 1.f);
                float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
                float fx = px - x0, fy = py - y0, fz = pz - z0;
                int xi = (int)x0, yi = (int)y0, zi = (int)z0;
                int mine[4];                               // corner 4l + u: row index, or -1 if the voxel is absent
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int xx = xi + (u & 1), yy = yi + (u >> 1), zz = zi + l;
                    const bool inb = valid && xx >= 0 && xx < lev.W && yy >= 0 && yy < lev.H && zz >= 0 && zz < lev.D;
                    const int key = inb ? (zz * lev.H + yy) * lev.W + xx : 0;
                    const uint2 rr = reinterpret_cast<const uint2*>(lev.wp)[key >> 5];
                    const uint32_t bit = 1u << (key & 31);
                    mine[u] = (inb && (rr.x & bit)) ? (int)(rr.y + __popc(rr.x & (bit - 1u))) : -1;
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int row = __shfl(mine[t & 3], pair0 | (t >> 2));
                    if (row >= 0) {
                        const float w = ((t & 1) ? fx : 1.f - fx) * (((t >> 1) & 1) ? fy : 1.f - fy) * ((t >> 2) ? fz : 1.f - fz);
                        const uint32_t r = (uint32_t)row * 12u + 2u * l;
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) {
                            axpy8(acc[s_][0], w, lev.rows, r + 4 * s_);
                            axpy8(acc[s_][1], w, lev.rows, r + 4 * s_ + 1);
                        }
                    }
                }
            }
        }
        // tokens[tile][slot][quad][j] (float4): this lane owns quads 4l .. 4l + 3 of every slot
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                tokens[((tile * 3 + s_) * 8 + 4 * l + 2 * o) * 32 + j] = acc[s_][o].a;
                tokens[((tile * 3 + s_) * 8 + 4 * l + 2 * o + 1) * 32 + j] = acc[s_][o].b;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the gather (BASELINE config 5; oracle/backward_explicit.py: folded_taps_bwd, step (i)): every tap is linear
// in its table, so d_tokens is scattered with the forward's tap weights into the gradients of the FOLDED tables
// (d_planes_f, d_feat_f, d_rows of the three voxel levels; zeroed by the caller) -- same stencils, same lane mapping
// (8 lanes x float4 per sample), loads replaced by hardware fp32 atomic adds (sums are order dependent in the last ulp,
// like the reference's grid_sample backward).  d_tok_bias[3][32] = sum over samples, reduced per workgroup first.
// The direct form (20 ms per step at 512 x 512 x 64): the training step uses the binned form below; this one stays as its check.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void scatter4(float4* dst, float w, const float4 d) {
    float* p = reinterpret_cast<float*>(dst);
    unsafeAtomicAdd(p + 0, w * d.x); unsafeAtomicAdd(p + 1, w * d.y); unsafeAtomicAdd(p + 2, w * d.z); unsafeAtomicAdd(p + 3, w * d.w);
}

struct LevelsBwd { sherf_vox_level l[3]; float* d_rows[3]; };

__global__ void __launch_bounds__(256) gather_tokens_bwd_kernel(const int32_t* __restrict__ counters, const float* __restrict__ geom,
                                                                const float4* __restrict__ d_tokens, int P, int Hf, int Wf, int H, int W,
                                                                LevelsBwd lv, const float* __restrict__ bounds,
                                                                const float* __restrict__ vox_min, int3 vox_sh, int64_t capacity,
                                                                float4* __restrict__ d_planes_f, float4* __restrict__ d_feat_f,
                                                                float* __restrict__ d_tok_bias) {
    __shared__ float s_bias[3][32];
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t n_tiles = (nv + 31) / 32;
    const int l = threadIdx.x & 7, j = threadIdx.x >> 3;
    float4 bsum[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t c = tile * 32 + j;
        if (c >= nv) continue;
        float4 d[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            d[s] = d_tokens[((tile * 3 + s) * 8 + l) * 32 + j];
            bsum[s].x += d[s].x; bsum[s].y += d[s].y; bsum[s].z += d[s].z; bsum[s].w += d[s].w;
        }
        const float* gm = geom + c * 8;
        const float xc[3] = {gm[0], gm[1], gm[2]};
        // ---- tri-plane (align_corners=False, zeros padding): slot p <- plane p ----
        float n[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) n[a] = 2.f * (xc[a] - bounds[a]) / (bounds[3 + a] - bounds[a]) - 1.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float ga = p == 2 ? n[2] : n[0];
            const float gb = p == 1 ? n[2] : n[1];
            float px = clampf(((ga + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
            float py = clampf(((gb + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
            float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
            int xi = (int)x0, yi = (int)y0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    int xx = xi + dx, yy = yi + dy;
                    if (xx >= 0 && xx < P && yy >= 0 && yy < P)
                        scatter4(d_planes_f + ((size_t)(p * P + yy) * P + xx) * 8 + l, (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy), d[p]);
                }
        }
        // ---- pixel-aligned feature map (align_corners=True): slots 0, 1 ----
        {
            float gx = 2.0f * gm[6] / (float)W - 1.0f, gy = 2.0f * gm[7] / (float)H - 1.0f;
            float px = clampf((gx + 1.f) * 0.5f * (Wf - 1), -2.f, (float)Wf + 1.f);
            float py = clampf((gy + 1.f) * 0.5f * (Hf - 1), -2.f, (float)Hf + 1.f);
            float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
            int xi = (int)x0, yi = (int)y0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    int xx = xi + dx, yy = yi + dy;
                    if (xx >= 0 && xx < Wf && yy >= 0 && yy < Hf) {
                        const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                        float4* t = d_feat_f + ((size_t)yy * Wf + xx) * 16;
                        scatter4(t + l, w, d[0]);
                        scatter4(t + 8 + l, w, d[1]);
                    }
                }
        }
        // ---- sparse voxel levels (align_corners=True): all three slots ----
        {
            float gz = ((xc[2] - vox_min[2]) / 0.005f) / (float)vox_sh.x * 2.f - 1.f;
            float gy = ((xc[1] - vox_min[1]) / 0.005f) / (float)vox_sh.y * 2.f - 1.f;
            float gx = ((xc[0] - vox_min[0]) / 0.005f) / (float)vox_sh.z * 2.f - 1.f;
#pragma unroll 1
            for (int L = 0; L < 3; ++L) {
                const sherf_vox_level& lev = lv.l[L];
                float px = clampf((gx + 1.f) * 0.5f * (lev.W - 1), -2.f, (float)lev.W + 1.f);
                float py = clampf((gy + 1.f) * 0.5f * (lev.H - 1), -2.f, (float)lev.H + 1.f);
                float pz = clampf((gz + 1.f) * 0.5f * (lev.D - 1), -2.f, (float)lev.D + 1.f);
                float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
                float fx = px - x0, fy = py - y0, fz = pz - z0;
                int xi = (int)x0, yi = (int)y0, zi = (int)z0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int xx = xi + (t & 1), yy = yi + ((t >> 1) & 1), zz = zi + (t >> 2);
                    if (!(xx >= 0 && xx < lev.W && yy >= 0 && yy < lev.H && zz >= 0 && zz < lev.D)) continue;
                    const int key = (zz * lev.H + yy) * lev.W + xx;
                    const uint2 rr = reinterpret_cast<const uint2*>(lev.wp)[key >> 5];
                    const uint32_t bit = 1u << (key & 31);
                    if (!(rr.x & bit)) continue;
                    const size_t row = rr.y + __popc(rr.x & (bit - 1u));
                    const float w = ((t & 1) ? fx : 1.f - fx) * (((t >> 1) & 1) ? fy : 1.f - fy) * ((t >> 2) ? fz : 1.f - fz);
                    float4* rp = reinterpret_cast<float4*>(lv.d_rows[L]) + row * 24;
                    scatter4(rp + l, w, d[0]);
                    scatter4(rp + 8 + l, w, d[1]);
                    scatter4(rp + 16 + l, w, d[2]);
                }
            }
        }
    }
    // d_tok_bias: sum this thread's samples over the block's 32 sample lanes, then 96 atomics per workgroup
    for (int i = threadIdx.x; i < 96; i += 256) (&s_bias[0][0])[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        atomicAdd(&s_bias[s][4 * l + 0], bsum[s].x); atomicAdd(&s_bias[s][4 * l + 1], bsum[s].y);
        atomicAdd(&s_bias[s][4 * l + 2], bsum[s].z); atomicAdd(&s_bias[s][4 * l + 3], bsum[s].w);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 96; i += 256) unsafeAtomicAdd(d_tok_bias + i, (&s_bias[0][0])[i]);
}


// ---------------------------------------------------------------------------------------------------------------------
// The same scatter with the samples BINNED by the coarsest tapped voxel cell they fall in (4 cm) and the lanes of a wave changing
// roles.  The direct form above issues ~2900 device-scope fp32 atomics per sample in 16-byte pieces of eight different rows per
// instruction, 78 % of them into the voxel rows, with chains of hundreds to thousands of same-address updates at the coarsest level:
// 20.3 ms of round 2's 87 ms of kernels per step (profiles/r02_train_step_final_rocprofv3_stats.txt).  Here, per wave and chunk of 64 of
// a bin's samples:
//   (1) lane = SAMPLE: every lane works out the stencils of its own sample once -- a target row / texel and a weight for each of
//       its 24 voxel, 12 plane and 4 feature-map corners, kept in registers;
//   (2) lane = CHANNEL: the wave walks its samples one by one, broadcasts that sample's (target, weight) pairs with v_readlane and
//       adds weight x d[channel]: 64 consecutive floats of one row per atomic instruction (whole cache lines, no two lanes on one
//       address).  All samples of a bin share their 8 corners at the coarsest level -- the contended ones -- so those sums stay in
//       REGISTERS (12 per lane) and reach memory once per wave and bin.
// Measured on the way (profiles/r03_scatter_ablation.txt, 690 K samples): LDS windows for every level with 8 samples x 8 channel quads per
// instruction 14.4 ms; the same with one sample per instruction and the stencils recomputed by all lanes 15-16 ms; with roles (1)/(2)
// but LDS windows 14.9 ms, of which 12 ms were the LDS float atomics themselves (~150-300 cycles per ds_add_f32 instruction on this
// part) -- the variant whose windows were shrunk to one voxel, i.e. whose adds went to memory instead, took 6.1 ms.  Hence: no LDS.
// Sums are order dependent in the last ulp exactly as before.  Four launches: count (+ list of non-empty bins), scan, fill, scatter.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBinNT = 256;                       // threads of the scatter workgroup: four waves, each on its own chunks of the bin

struct VoxTap { int xi, yi, zi; float fx, fy, fz; };

__device__ __forceinline__ void vox_grid_coords(const float* __restrict__ gm, const float* __restrict__ vox_min, int3 vox_sh, float& gx, float& gy, float& gz) {
    gz = ((gm[2] - vox_min[2]) / 0.005f) / (float)vox_sh.x * 2.f - 1.f;
    gy = ((gm[1] - vox_min[1]) / 0.005f) / (float)vox_sh.y * 2.f - 1.f;
    gx = ((gm[0] - vox_min[0]) / 0.005f) / (float)vox_sh.z * 2.f - 1.f;
}

__device__ __forceinline__ VoxTap vox_tap(const sherf_vox_level& lev, float gx, float gy, float gz) {     // (the forward's stencil, align_corners=True)
    const float px = clampf((gx + 1.f) * 0.5f * (lev.W - 1), -2.f, (float)lev.W + 1.f);
    const float py = clampf((gy + 1.f) * 0.5f * (lev.H - 1), -2.f, (float)lev.H + 1.f);
    const float pz = clampf((gz + 1.f) * 0.5f * (lev.D - 1), -2.f, (float)lev.D + 1.f);
    const float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
    return VoxTap{(int)x0, (int)y0, (int)z0, px - x0, py - y0, pz - z0};
}

struct PlaneTap { int xi, yi; float fx, fy; };
__device__ __forceinline__ PlaneTap plane_tap(int p, const float* n, int P) {                            // (align_corners=False)
    const float ga = p == 2 ? n[2] : n[0], gb = p == 1 ? n[2] : n[1];
    const float px = clampf(((ga + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
    const float py = clampf(((gb + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
    const float x0 = floorf(px), y0 = floorf(py);
    return PlaneTap{(int)x0, (int)y0, px - x0, py - y0};
}

struct PixTap { int xi, yi; float fx, fy; };
__device__ __forceinline__ PixTap pix_tap(const float* __restrict__ gm, int W, int H, int Wf, int Hf) {   // (align_corners=True; gm[6..7] = pixel of the observed view)
    const float gx = 2.0f * gm[6] / (float)W - 1.0f, gy = 2.0f * gm[7] / (float)H - 1.0f;
    const float px = clampf((gx + 1.f) * 0.5f * (Wf - 1), -2.f, (float)Wf + 1.f);
    const float py = clampf((gy + 1.f) * 0.5f * (Hf - 1), -2.f, (float)Hf + 1.f);
    const float x0 = floorf(px), y0 = floorf(py);
    return PixTap{(int)x0, (int)y0, px - x0, py - y0};
}

// bin of a sample: its base corner at the coarsest tapped level, each component in [-2, dim + 1] after the stencil's clamp
__device__ __forceinline__ int bin_of(const sherf_vox_level& lev, const VoxTap& t) {
    return ((t.zi + 2) * (lev.H + 4) + (t.yi + 2)) * (lev.W + 4) + (t.xi + 2);
}

// scratch words: [0] number of non-empty bins, [4 ..) counts[n_bins], cursor[n_bins], offsets[n_bins], nonempty[n_bins], bin[cap], sorted[cap]
struct BinWs { int32_t *n_nonempty, *counts, *cursor, *offsets, *nonempty, *bin, *sorted, *blocks; int n_bins; };
__host__ __device__ inline BinWs bin_ws(int32_t* scratch, int n_bins, int64_t cap) {
    BinWs w;
    w.n_nonempty = scratch; w.counts = scratch + 4; w.cursor = w.counts + n_bins; w.offsets = w.cursor + n_bins; w.nonempty = w.offsets + n_bins;
    w.bin = w.nonempty + n_bins; w.sorted = w.bin + cap; w.blocks = w.sorted + cap; w.n_bins = n_bins;          // blocks[ceil(n_bins / 1024)]: run order only
    return w;
}

__global__ void __launch_bounds__(256) bin_count_kernel(const int32_t* __restrict__ counters, const float* __restrict__ geom, sherf_vox_level lev2,
                                                        const float* __restrict__ vox_min, int3 vox_sh, int64_t capacity, BinWs w) {
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nv) return;
    float gx, gy, gz;
    vox_grid_coords(geom + c * 8, vox_min, vox_sh, gx, gy, gz);
    const int b = bin_of(lev2, vox_tap(lev2, gx, gy, gz));
    w.bin[c] = b;
    if (atomicAdd(w.counts + b, 1) == 0) w.nonempty[atomicAdd(w.n_nonempty, 1)] = b;       // first sample of the bin lists it
}

// exclusive scan of counts -> offsets by ONE workgroup (tens of thousands of bins: a chunk per thread, the chunk totals scanned in LDS)
__global__ void __launch_bounds__(1024) bin_scan_kernel(BinWs w) {
    __shared__ int s_tot[1024];
    const int per = (w.n_bins + 1023) / 1024;
    const int b0 = min((int)threadIdx.x * per, w.n_bins), b1 = min(b0 + per, w.n_bins);
    int t = 0;
    for (int b = b0; b < b1; ++b) t += w.counts[b];
    s_tot[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive scan of the chunk totals
        const int v = threadIdx.x >= (unsigned)o ? s_tot[threadIdx.x - o] : 0;
        __syncthreads();
        s_tot[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_tot[threadIdx.x] - t;
    for (int b = b0; b < b1; ++b) { w.offsets[b] = run; run += w.counts[b]; }
}

__global__ void __launch_bounds__(256) bin_fill_kernel(const int32_t* __restrict__ counters, int64_t capacity, BinWs w) {
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nv) return;
    const int b = w.bin[c];
    w.sorted[w.offsets[b] + atomicAdd(w.cursor + b, 1)] = (int32_t)c;
}

__global__ void __launch_bounds__(kBinNT) gather_tokens_bwd_binned_kernel(const float* __restrict__ geom, const float4* __restrict__ d_tokens, int P, int Hf,
                                                                          int Wf, int H, int W, LevelsBwd lv, const float* __restrict__ bounds,
                                                                          const float* __restrict__ vox_min, int3 vox_sh, BinWs w,
                                                                          float4* __restrict__ d_planes_f, float4* __restrict__ d_feat_f,
                                                                          float* __restrict__ d_tok_bias, int dbg) {
    // (timing ablations, results incomplete: sherf_set_debug bit 14 no pixel-aligned taps, bit 15 no voxel taps, bit 16 no plane taps;
    //  bit 13: the coarsest level through memory like the others instead of registers -- same results: tools/scatter_bench.py, tests)
    const bool do_pix = !(dbg & 16384), do_vox = !(dbg & 32768), do_pl = !(dbg & 65536), reg2 = !(dbg & 8192);
    __shared__ float s_bias[96];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, c31 = lane & 31;
    const int n_list = *w.n_nonempty;
    // (descriptors in registers: indexed at run time they are re-read from the kernel-argument segment at every use)
    const sherf_vox_level levs[3] = {lv.l[0], lv.l[1], lv.l[2]};
    float* const drows[3] = {lv.d_rows[0], lv.d_rows[1], lv.d_rows[2]};
    const float* dt = reinterpret_cast<const float*>(d_tokens);
    float* const dpl = reinterpret_cast<float*>(d_planes_f);
    float* const dfm = reinterpret_cast<float*>(d_feat_f);
    float bs0 = 0.f, bs1 = 0.f;                    // d_tok_bias: this lane's channel of slots 0-1, of slot 2 (lower half of the wave)
    for (int bi = blockIdx.x; bi < n_list; bi += gridDim.x) {
        const int b = w.nonempty[bi], cnt = w.counts[b], start = w.offsets[b];
        float a2[8], b2[4];                        // coarsest level: corner k x channel `lane` (slots 0-1); corner pair x channel 64 + c31 (slot 2)
#pragma unroll
        for (int k = 0; k < 8; ++k) a2[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) b2[k] = 0.f;
        int row2 = -1;                             // lanes 0-7: the row of corner `lane` of the bin's cell (the same for all its samples)
        for (int base = wv * 64; base < cnt; base += (kBinNT / 64) * 64) {
            // ---- (1) lane = sample ----
            const bool live = base + lane < cnt;
            const int cs = live ? w.sorted[start + base + lane] : 0;
            int tv[24], tp[12], tf[4];
            float wvx[24], wpl[12], wfm[4];
            {
                const float* gm = geom + (int64_t)cs * 8;
                float n[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) n[a] = 2.f * (gm[a] - bounds[a]) / (bounds[3 + a] - bounds[a]) - 1.f;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const PlaneTap t = plane_tap(p, n, P);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int xx = t.xi + (k & 1), yy = t.yi + (k >> 1);
                        tp[4 * p + k] = (live && do_pl && xx >= 0 && xx < P && yy >= 0 && yy < P) ? (p * P + yy) * P + xx : -1;
                        wpl[4 * p + k] = ((k & 1) ? t.fx : 1.f - t.fx) * ((k >> 1) ? t.fy : 1.f - t.fy);
                    }
                }
                const PixTap t = pix_tap(gm, W, H, Wf, Hf);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int xx = t.xi + (k & 1), yy = t.yi + (k >> 1);
                    tf[k] = (live && do_pix && xx >= 0 && xx < Wf && yy >= 0 && yy < Hf) ? yy * Wf + xx : -1;
                    wfm[k] = ((k & 1) ? t.fx : 1.f - t.fx) * ((k >> 1) ? t.fy : 1.f - t.fy);
                }
                float gx, gy, gz;
                vox_grid_coords(gm, vox_min, vox_sh, gx, gy, gz);
#pragma unroll
                for (int L = 0; L < 3; ++L) {
                    const sherf_vox_level lev = levs[L];
                    const VoxTap vt = vox_tap(lev, gx, gy, gz);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int xx = vt.xi + (k & 1), yy = vt.yi + ((k >> 1) & 1), zz = vt.zi + (k >> 2);
                        // (a dead lane or a switched-off level weighs nothing: the register sums of the coarsest level take every sample)
                        wvx[8 * L + k] = (live && do_vox) ? ((k & 1) ? vt.fx : 1.f - vt.fx) * (((k >> 1) & 1) ? vt.fy : 1.f - vt.fy) * ((k >> 2) ? vt.fz : 1.f - vt.fz) : 0.f;
                        int tg = -1;
                        if (live && do_vox && xx >= 0 && xx < lev.W && yy >= 0 && yy < lev.H && zz >= 0 && zz < lev.D) {
                            const int key = (zz * lev.H + yy) * lev.W + xx;
                            const uint2 rr = reinterpret_cast<const uint2*>(lev.wp)[key >> 5];
                            const uint32_t bit = 1u << (key & 31);
                            if (rr.x & bit) tg = (int)(rr.y + __popc(rr.x & (bit - 1u)));
                        }
                        tv[8 * L + k] = tg;
                    }
                }
            }
            // the rows of the bin's eight coarsest-level corners: every sample of the bin has the same ones; lane k < 8 keeps corner k's
            if (base == wv * 64) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = __builtin_amdgcn_readlane(tv[16 + k], 0);                      // (lane 0 is live whenever the wave has a chunk)
                    if (lane == k) row2 = r;
                }
            }
            // ---- (2) lane = channel ----
            const int nact = min(64, cnt - base);
            for (int sidx = 0; sidx < nact; ++sidx) {
                const int c = __builtin_amdgcn_readlane(cs, sidx);
                const int64_t tile = c >> 5;
                const int j = c & 31;
                // d_tokens[tile][slot][quad][sample j] float4: channel ch of slot s = component ch & 3 of quad ch >> 2
                const int64_t dbase = (((tile * 3) * 8 + (c31 >> 2)) * 32 + j) * 4 + (c31 & 3);
                const float dp0 = dt[dbase], dp1 = dt[dbase + 1024], d1 = dt[dbase + 2048];            // channel c31 of slots 0, 1, 2 (both halves)
                const float d0 = half ? dp1 : dp0;                                                       // channel `lane` of slots 0-1
                bs0 += d0; bs1 += half ? 0.f : d1;
#define SHERF_RL(v) __builtin_amdgcn_readlane((v), sidx)
#define SHERF_RLF(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), sidx))
                // tri-planes: plane p <- slot p (32 channels); the two corners dx = 0 / 1 share an instruction
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const float dpv = p == 0 ? dp0 : (p == 1 ? dp1 : d1);
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const int tA = SHERF_RL(tp[4 * p + 2 * dy]), tB = SHERF_RL(tp[4 * p + 2 * dy + 1]);
                        const float wA = SHERF_RLF(wpl[4 * p + 2 * dy]), wB = SHERF_RLF(wpl[4 * p + 2 * dy + 1]);
                        if (tA < 0 && tB < 0) continue;
                        const int tg = half ? tB : tA;
                        if (tg >= 0) unsafeAtomicAdd(dpl + (size_t)tg * 32 + c31, (half ? wB : wA) * dpv);
                    }
                }
                // pixel-aligned feature map: slots 0-1 = 64 channels, one corner per instruction
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int tg = SHERF_RL(tf[k]);
                    if (tg >= 0) unsafeAtomicAdd(dfm + (size_t)tg * 64 + lane, SHERF_RLF(wfm[k]) * d0);
                }
                // voxel levels 0, 1: per corner one instruction for channels 0-63, per corner PAIR one for channels 64-95
#pragma unroll
                for (int L = 0; L < 3; ++L) {
                    if (L == 2 && reg2) break;
                    float* drow = drows[L];
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const int tA = SHERF_RL(tv[8 * L + k]), tB = SHERF_RL(tv[8 * L + k + 1]);
                        const float wA = SHERF_RLF(wvx[8 * L + k]), wB = SHERF_RLF(wvx[8 * L + k + 1]);
                        if (tA >= 0) unsafeAtomicAdd(drow + (size_t)tA * 96 + lane, wA * d0);
                        if (tB >= 0) unsafeAtomicAdd(drow + (size_t)tB * 96 + lane, wB * d0);
                        if (tA < 0 && tB < 0) continue;
                        const int tg = half ? tB : tA;
                        if (tg >= 0) unsafeAtomicAdd(drow + (size_t)tg * 96 + 64 + c31, (half ? wB : wA) * d1);
                    }
                }
                // coarsest level: the bin's own eight corners, summed in registers
                if (reg2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) a2[k] += SHERF_RLF(wvx[16 + k]) * d0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float wA = SHERF_RLF(wvx[16 + 2 * k]), wB = SHERF_RLF(wvx[16 + 2 * k + 1]);
                        b2[k] += (half ? wB : wA) * d1;
                    }
                }
#undef SHERF_RL
#undef SHERF_RLF
            }
        }
        // the wave's sums of the coarsest level: one atomic per row and channel
        if (reg2 && wv * 64 < cnt) {
            float* drow = drows[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = __builtin_amdgcn_readlane(row2, k);
                if (r >= 0) unsafeAtomicAdd(drow + (size_t)r * 96 + lane, a2[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rA = __builtin_amdgcn_readlane(row2, 2 * k), rB = __builtin_amdgcn_readlane(row2, 2 * k + 1);
                const int r = half ? rB : rA;
                if (r >= 0) unsafeAtomicAdd(drow + (size_t)r * 96 + 64 + c31, b2[k]);
            }
        }
    }
    // d_tok_bias
    for (int i = tid; i < 96; i += kBinNT) s_bias[i] = 0.f;
    __syncthreads();
    atomicAdd(s_bias + lane, bs0);
    if (lane < 32) atomicAdd(s_bias + 64 + lane, bs1);
    __syncthreads();
    for (int i = tid; i < 96; i += kBinNT) unsafeAtomicAdd(d_tok_bias + i, s_bias[i]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same scatter over the samples SORTED by the finest tapped voxel cell they fall in (1 cm), every level's sums in registers.
// The kernel above keeps the coarsest level's eight corner rows in registers because all samples of a 4 cm bin share them, and sends the two
// finer levels' 16 corners, the 12 plane and the 4 feature-map corners of every sample to memory: 34 atomic instructions of 64 elements per
// sample.  With the samples in the order of their finest cell (x fastest) the corner rows of EVERY level -- and the base texels of the planes and of
// the feature map -- stay the same over runs of consecutive samples, so a wave walks a contiguous stretch of the sorted list, adds into 46
// registers per lane and sends a level's eight rows (a plane's / the feature map's four texels) to memory when that cell changes (run-length).
// Measured on the MI355X (tools/scatter_bench.py, 690 K samples): 4.1 -> 2.4-2.5 ms, of which sorting + the walk without any tap 0.9; a 1 cm cell
// holds 4.9 samples on average, so a finest-level row still receives one flush per adjacent cell -- ~260 M atomically added ELEMENTS in all (round 3's
// kernel: 1.5 G), and that count, not bytes or latency, is what the time follows (~175 G elements / s through the L2 atomic units; the gradient rows
// loaded four samples ahead, or staged through LDS, changed nothing).  The bins become the finest level's cells (1.67 M at the bench subject, 140 K
// of them occupied): their scan runs on every workgroup (local scans + block totals; the one-workgroup scan above would walk 1 600 bins per thread).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) bin_scan_blocks_kernel(BinWs w) {          // offsets[b] = exclusive scan inside the block of 1024 bins; blocks[blk] = its total
    __shared__ int s_v[1024];
    const int b = blockIdx.x * 1024 + threadIdx.x;
    const int v = b < w.n_bins ? w.counts[b] : 0;
    s_v[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int a = threadIdx.x >= (unsigned)o ? s_v[threadIdx.x - o] : 0;
        __syncthreads();
        s_v[threadIdx.x] += a;
        __syncthreads();
    }
    if (b < w.n_bins) w.offsets[b] = s_v[threadIdx.x] - v;
    if (threadIdx.x == 1023) w.blocks[blockIdx.x] = s_v[1023];
}

__global__ void __launch_bounds__(1024) bin_scan_top_kernel(BinWs w, int n_blocks) {        // blocks[] -> exclusive scan, by one workgroup
    __shared__ int s_v[1024];
    int carry = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const int v = i < n_blocks ? w.blocks[i] : 0;
        s_v[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = threadIdx.x >= (unsigned)o ? s_v[threadIdx.x - o] : 0;
            __syncthreads();
            s_v[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_blocks) w.blocks[i] = carry + s_v[threadIdx.x] - v;
        const int tot = s_v[1023];
        __syncthreads();
        carry += tot;
    }
}

__global__ void __launch_bounds__(256) bin_fill_blocks_kernel(const int32_t* __restrict__ counters, int64_t capacity, BinWs w) {
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nv) return;
    const int b = w.bin[c];
    w.sorted[w.blocks[b >> 10] + w.offsets[b] + atomicAdd(w.cursor + b, 1)] = (int32_t)c;
}

__global__ void __launch_bounds__(kBinNT) gather_tokens_bwd_runs_kernel(const int32_t* __restrict__ counters, int64_t capacity, const float* __restrict__ geom,
                                                                        const float4* __restrict__ d_tokens, int P, int Hf, int Wf, int H, int W, LevelsBwd lv,
                                                                        const float* __restrict__ bounds, const float* __restrict__ vox_min, int3 vox_sh,
                                                                        const int32_t* __restrict__ sorted, float4* __restrict__ d_planes_f,
                                                                        float4* __restrict__ d_feat_f, float* __restrict__ d_tok_bias, int dbg) {
    const bool do_pix = !(dbg & 16384), do_vox = !(dbg & 32768), do_pl = !(dbg & 65536);          // (timing ablations as above)
    __shared__ float s_bias[96];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, c31 = lane & 31;
    const sherf_vox_level levs[3] = {lv.l[0], lv.l[1], lv.l[2]};
    float* const drows[3] = {lv.d_rows[0], lv.d_rows[1], lv.d_rows[2]};
    const float* dt = reinterpret_cast<const float*>(d_tokens);
    float* const dpl = reinterpret_cast<float*>(d_planes_f);
    float* const dfm = reinterpret_cast<float*>(d_feat_f);
    float bs0 = 0.f, bs1 = 0.f;
    // this wave's contiguous stretch of the sorted list, in chunks of 64 samples
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t n_chunks = (nv + 63) >> 6, n_waves = (int64_t)gridDim.x * (kBinNT / 64);
    const int64_t per = (n_chunks + n_waves - 1) / n_waves, wid = (int64_t)blockIdx.x * (kBinNT / 64) + wv;
    const int64_t ch0 = min(wid * per, n_chunks), ch1 = min(ch0 + per, n_chunks);
    float acc[3][8], accb[3][4];                   // level L: corner k x channel `lane` (slots 0-1); corner pair x channel 64 + c31 (slot 2)
    int cur[3] = {-1, -1, -1};                     // the cell of the running sums (wave-uniform); -1: none
    int rows[3] = {-1, -1, -1};                    // lanes 0-7: the row of corner `lane` of that cell
#pragma unroll
    for (int L = 0; L < 3; ++L) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[L][k] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) accb[L][k] = 0.f;
    }
    // the same for the tri-planes (plane p: corner (dx = half, dy) x channel c31) and the pixel-aligned feature map (corner k x channel `lane`): samples of one
    // 1 cm cell fall on a handful of texels
    float accp[3][2], accf[4];
    int curp[3] = {-1, -1, -1}, curf = -1, tgp[3] = {-1, -1, -1}, tgf = -1;       // tgp[p] / tgf: lanes 0-3 hold the texel of corner `lane`
#pragma unroll
    for (int p = 0; p < 3; ++p) accp[p][0] = accp[p][1] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) accf[k] = 0.f;
    auto flush_plane = [&](int p) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int tA = __builtin_amdgcn_readlane(tgp[p], 2 * dy), tB = __builtin_amdgcn_readlane(tgp[p], 2 * dy + 1);
            const int tg = half ? tB : tA;
            if (tg >= 0) unsafeAtomicAdd(dpl + (size_t)tg * 32 + c31, accp[p][dy]);
            accp[p][dy] = 0.f;
        }
    };
    auto flush_feat = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tg = __builtin_amdgcn_readlane(tgf, k);
            if (tg >= 0) unsafeAtomicAdd(dfm + (size_t)tg * 64 + lane, accf[k]);
            accf[k] = 0.f;
        }
    };
    auto flush = [&](int L) {                      // a level's running sums to memory: one atomic per row and channel
        float* drow = drows[L];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = __builtin_amdgcn_readlane(rows[L], k);
            if (r >= 0) unsafeAtomicAdd(drow + (size_t)r * 96 + lane, acc[L][k]);
            acc[L][k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rA = __builtin_amdgcn_readlane(rows[L], 2 * k), rB = __builtin_amdgcn_readlane(rows[L], 2 * k + 1);
            const int r = half ? rB : rA;
            if (r >= 0) unsafeAtomicAdd(drow + (size_t)r * 96 + 64 + c31, accb[L][k]);
            accb[L][k] = 0.f;
        }
    };
    for (int64_t ch = ch0; ch < ch1; ++ch) {
        // ---- (1) lane = sample ----
        const int64_t base = ch << 6;
        const bool live = base + lane < nv;
        const int cs = live ? sorted[base + lane] : 0;
        int tv[24], tp[12], tf[4], ck[3], kp[3], kf;
        float wvx[24], wpl[12], wfm[4];
        {
            const float* gm = geom + (int64_t)cs * 8;
            float n[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) n[a] = 2.f * (gm[a] - bounds[a]) / (bounds[3 + a] - bounds[a]) - 1.f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const PlaneTap t = plane_tap(p, n, P);
                kp[p] = (t.yi + 2) * (P + 4) + (t.xi + 2);                 // the stencil's base texel (components in [-2, P + 1] after the clamp)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int xx = t.xi + (k & 1), yy = t.yi + (k >> 1);
                    tp[4 * p + k] = (live && do_pl && xx >= 0 && xx < P && yy >= 0 && yy < P) ? (p * P + yy) * P + xx : -1;
                    wpl[4 * p + k] = (live && do_pl) ? ((k & 1) ? t.fx : 1.f - t.fx) * ((k >> 1) ? t.fy : 1.f - t.fy) : 0.f;
                }
            }
            const PixTap t = pix_tap(gm, W, H, Wf, Hf);
            kf = (t.yi + 2) * (Wf + 4) + (t.xi + 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = t.xi + (k & 1), yy = t.yi + (k >> 1);
                tf[k] = (live && do_pix && xx >= 0 && xx < Wf && yy >= 0 && yy < Hf) ? yy * Wf + xx : -1;
                wfm[k] = (live && do_pix) ? ((k & 1) ? t.fx : 1.f - t.fx) * ((k >> 1) ? t.fy : 1.f - t.fy) : 0.f;
            }
            float gx, gy, gz;
            vox_grid_coords(gm, vox_min, vox_sh, gx, gy, gz);
#pragma unroll
            for (int L = 0; L < 3; ++L) {
                const sherf_vox_level lev = levs[L];
                const VoxTap vt = vox_tap(lev, gx, gy, gz);
                ck[L] = bin_of(lev, vt);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int xx = vt.xi + (k & 1), yy = vt.yi + ((k >> 1) & 1), zz = vt.zi + (k >> 2);
                    wvx[8 * L + k] = (live && do_vox) ? ((k & 1) ? vt.fx : 1.f - vt.fx) * (((k >> 1) & 1) ? vt.fy : 1.f - vt.fy) * ((k >> 2) ? vt.fz : 1.f - vt.fz) : 0.f;
                    const bool in = live && do_vox && xx >= 0 && xx < lev.W && yy >= 0 && yy < lev.H && zz >= 0 && zz < lev.D;
                    // (the occupancy record is loaded unconditionally from a clamped key and the result selected: a load inside the branch is waited for on the spot)
                    const int key = in ? (zz * lev.H + yy) * lev.W + xx : 0;
                    const uint2 rr = reinterpret_cast<const uint2*>(lev.wp)[key >> 5];
                    const uint32_t bit = 1u << (key & 31);
                    tv[8 * L + k] = (in && (rr.x & bit)) ? (int)(rr.y + __popc(rr.x & (bit - 1u))) : -1;
                }
            }
        }
        // ---- (2) lane = channel ----
        const int nact = (int)min((int64_t)64, nv - base);
        // (the gradient rows are loaded where they are used: four samples ahead, or staged through LDS 32 samples at a time, measured the same 2.2 ms --
        //  the walk does not wait for them; what is left is the atomic units' element rate, see the header)
        {
          for (int sidx = 0; sidx < nact; ++sidx) {
            const int c = __builtin_amdgcn_readlane(cs, sidx);
            const int64_t dbase = ((((int64_t)(c >> 5) * 3) * 8 + (c31 >> 2)) * 32 + (c & 31)) * 4 + (c31 & 3);       // d_tokens[tile][slot][quad][sample j] float4
            const float dp0 = dt[dbase], dp1 = dt[dbase + 1024], d1 = dt[dbase + 2048];            // channel c31 of slots 0, 1, 2 (both halves)
            const float d0 = half ? dp1 : dp0;                                                       // channel `lane` of slots 0-1
            bs0 += d0; bs1 += half ? 0.f : d1;
#define SHERF_RL(v) __builtin_amdgcn_readlane((v), sidx)
#define SHERF_RLF(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), sidx))
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float dpv = p == 0 ? dp0 : (p == 1 ? dp1 : d1);
                const int key = SHERF_RL(kp[p]);
                if (key != curp[p]) {
                    if (curp[p] >= 0) flush_plane(p);
                    curp[p] = key;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = SHERF_RL(tp[4 * p + k]);
                        if (lane == k) tgp[p] = r;
                    }
                }
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const float wA = SHERF_RLF(wpl[4 * p + 2 * dy]), wB = SHERF_RLF(wpl[4 * p + 2 * dy + 1]);
                    accp[p][dy] += (half ? wB : wA) * dpv;
                }
            }
            {
                const int key = SHERF_RL(kf);
                if (key != curf) {
                    if (curf >= 0) flush_feat();
                    curf = key;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = SHERF_RL(tf[k]);
                        if (lane == k) tgf = r;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) accf[k] += SHERF_RLF(wfm[k]) * d0;
            }
#pragma unroll
            for (int L = 0; L < 3; ++L) {
                const int cell = SHERF_RL(ck[L]);
                if (cell != cur[L]) {                                      // (wave-uniform) the level's cell changes: its sums leave, its rows are re-read
                    if (cur[L] >= 0) flush(L);
                    cur[L] = cell;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = SHERF_RL(tv[8 * L + k]);
                        if (lane == k) rows[L] = r;
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[L][k] += SHERF_RLF(wvx[8 * L + k]) * d0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float wA = SHERF_RLF(wvx[8 * L + 2 * k]), wB = SHERF_RLF(wvx[8 * L + 2 * k + 1]);
                    accb[L][k] += (half ? wB : wA) * d1;
                }
            }
#undef SHERF_RL
#undef SHERF_RLF
          }
        }
    }
#pragma unroll
    for (int L = 0; L < 3; ++L)
        if (cur[L] >= 0) flush(L);
#pragma unroll
    for (int p = 0; p < 3; ++p)
        if (curp[p] >= 0) flush_plane(p);
    if (curf >= 0) flush_feat();
    // d_tok_bias
    for (int i = tid; i < 96; i += kBinNT) s_bias[i] = 0.f;
    __syncthreads();
    atomicAdd(s_bias + lane, bs0);
    if (lane < 32) atomicAdd(s_bias + 64 + lane, bs1);
    __syncthreads();
    for (int i = tid; i < 96; i += kBinNT) unsafeAtomicAdd(d_tok_bias + i, s_bias[i]);
}

}  // namespace

static int gather_tokens_impl(const int32_t* counters, const float* geom, const float* planes_f, int P,
                              const float* feat_f, int Hf, int Wf, const float* img4, int H, int W,
                              const sherf_vox_level* levels_host, const float* tok_bias, const float* bounds,
                              const float* vox_min, const int32_t* vox_sh_host, int mode, int64_t capacity, float* tokens,
                              float* extras, void* pefrag, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && geom && planes_f && feat_f && img4 && tok_bias && bounds && vox_min && vox_sh_host && tokens && extras);
    const bool branchless = (mode & 4) != 0 || SHERF_GATHER_BRANCHLESS;
    const bool squeezed = branchless && (mode & 8) != 0;
    const bool half_tables = (mode & 16) != 0;
    const int part = (mode >> 8) & 255, nparts = (mode >> 16) & 255;       // a contiguous part of the tiles (common.h: sherf_part_range)
    SHERF_CHECK_ARG(nparts == 0 || part < nparts);
    mode &= 3;
    SHERF_CHECK_ARG(mode >= 0 && mode <= 2 && (mode == 1 || levels_host));
    SHERF_CHECK_ARG(P > 0 && Hf > 0 && Wf > 0 && H > 0 && W > 0 && capacity > 0);
    Levels lv = {};
    for (int i = 0; i < 3 && mode != 1; ++i) {
        lv.l[i] = levels_host[i];
        SHERF_CHECK_ARG(lv.l[i].wp && lv.l[i].rows && lv.l[i].D > 0 && lv.l[i].H > 0 && lv.l[i].W > 0);
    }
    int3 sh = make_int3(vox_sh_host[0], vox_sh_host[1], vox_sh_host[2]);
    const int64_t tiles = nparts > 1 ? ((capacity + 255) / 256 + nparts - 1) / nparts * 8 + 8 : (capacity + 31) / 32;    // most a part can hold
    const int kmode = mode | (part << 8) | (nparts << 16);
#define SHERF_GATHER(BL, MW, HT)                                                                                               \
    hipLaunchKernelGGL((gather_tokens_kernel<BL, MW, HT>), dim3((unsigned)(tiles < 16384 ? (tiles + 7) / 8 * 8 : 16384)), dim3(256), 0, as_stream(stream), \
                       counters, geom, static_cast<const void*>(planes_f), P, static_cast<const void*>(feat_f), Hf, Wf,         \
                       reinterpret_cast<const float4*>(img4), H, W, lv, reinterpret_cast<const float4*>(tok_bias), bounds,       \
                       vox_min, sh, capacity, reinterpret_cast<float4*>(tokens), extras, g_sherf_debug, kmode)
    // the eight-channel kernel indexes its fp16 tables with 32-bit offsets of 16-byte units
    SHERF_CHECK_ARG(!half_tables || ((int64_t)3 * P * P * 64 < ((int64_t)1 << 32) && (int64_t)Hf * Wf * 128 < ((int64_t)1 << 32)));
    // the encodings as fragments (sherf_gather_tokens_pe): the eight-channel kernel's pass that visits every sample once (mode 0 or 1)
    SHERF_CHECK_ARG(!pefrag || (half_tables && !branchless && mode != 2));
    if (pefrag) {
        const int64_t pairs = (tiles + 1) / 2;
        hipLaunchKernelGGL((gather_tokens_h8_kernel<false, true>), dim3((unsigned)(pairs < 16384 ? (pairs + 7) / 8 * 8 : 16384)), dim3(256), 0, as_stream(stream),
                           counters, geom, static_cast<const void*>(planes_f), P, static_cast<const void*>(feat_f), Hf, Wf,
                           reinterpret_cast<const float4*>(img4), H, W, lv, reinterpret_cast<const float4*>(tok_bias), bounds, vox_min, sh,
                           capacity, reinterpret_cast<float4*>(tokens), extras, g_sherf_debug, kmode, reinterpret_cast<uint4*>(pefrag));
    } else
    if (half_tables && !branchless && !(g_sherf_debug & 2048)) {      // eight channels per lane (debug bit 11: the four-per-lane kernel on fp16 tables)
        const int64_t pairs = (tiles + 1) / 2;
        // debug bit 29: the records of the two coarser levels staged in LDS (gather_tokens_h8_kernel<true>), a workgroup walks ~`ppw` pairs (4; debug bit
        // 30: 16); only when they fit 40 KiB (four workgroups per CU keep their LDS) and every level is tapped (mode 0 / 2)
        const size_t stage_bytes = mode == 1 ? 0 : 8 * (((size_t)lv.l[1].D * lv.l[1].H * lv.l[1].W + 31) / 32 + ((size_t)lv.l[2].D * lv.l[2].H * lv.l[2].W + 31) / 32);
        if ((g_sherf_debug & (1 << 29)) && mode != 1 && stage_bytes <= 40 * 1024) {
            const int ppw = (g_sherf_debug & (1 << 30)) ? 16 : 4;
            const int64_t wgs = std::max<int64_t>(8, std::min<int64_t>(16384, ((pairs + ppw - 1) / ppw + 7) / 8 * 8));
            hipLaunchKernelGGL(gather_tokens_h8_kernel<true>, dim3((unsigned)wgs), dim3(256), stage_bytes, as_stream(stream),
                               counters, geom, static_cast<const void*>(planes_f), P, static_cast<const void*>(feat_f), Hf, Wf,
                               reinterpret_cast<const float4*>(img4), H, W, lv, reinterpret_cast<const float4*>(tok_bias), bounds, vox_min, sh,
                               capacity, reinterpret_cast<float4*>(tokens), extras, g_sherf_debug, kmode, static_cast<uint4*>(nullptr));
        } else
        if (mode == 0 && nparts <= 1 && !(sherf_experiment() & 1024))      // two lanes per sample (round 6; SHERF_EXPERIMENT bit 10: the eight-channel kernel -- A/B runs)
        hipLaunchKernelGGL(gather_tokens_h16_kernel, dim3((unsigned)((tiles + 3) / 4 < 16384 ? ((tiles + 3) / 4 + 7) / 8 * 8 : 16384)), dim3(256), 0, as_stream(stream),
                           counters, geom, static_cast<const void*>(planes_f), P, static_cast<const void*>(feat_f), Hf, Wf,
                           reinterpret_cast<const float4*>(img4), H, W, lv, reinterpret_cast<const float4*>(tok_bias), bounds, vox_min, sh,
                           capacity, reinterpret_cast<float4*>(tokens), extras, g_sherf_debug);
        else if (!(sherf_experiment() & 512))        // the voxel rows of the next corner requested ahead (PF; measured -2.8 % of this kernel, -1 % of the frame, bit-identical: profiles/r06_call_k_*).  SHERF_EXPERIMENT bit 9: round 5's one-corner-at-a-time loop (A/B runs)
        hipLaunchKernelGGL((gather_tokens_h8_kernel<false, false, true>), dim3((unsigned)(pairs < 16384 ? (pairs + 7) / 8 * 8 : 16384)), dim3(256), 0, as_stream(stream),
                           counters, geom, static_cast<const void*>(planes_f), P, static_cast<const void*>(feat_f), Hf, Wf,
                           reinterpret_cast<const float4*>(img4), H, W, lv, reinterpret_cast<const float4*>(tok_bias), bounds, vox_min, sh,
                           capacity, reinterpret_cast<float4*>(tokens), extras, g_sherf_debug, kmode, static_cast<uint4*>(nullptr));
        else
        hipLaunchKernelGGL(gather_tokens_h8_kernel<false>, dim3((unsigned)(pairs < 16384 ? (pairs + 7) / 8 * 8 : 16384)), dim3(256), 0, as_stream(stream),
                           counters, geom, static_cast<const void*>(planes_f), P, static_cast<const void*>(feat_f), Hf, Wf,
                           reinterpret_cast<const float4*>(img4), H, W, lv, reinterpret_cast<const float4*>(tok_bias), bounds, vox_min, sh,
                           capacity, reinterpret_cast<float4*>(tokens), extras, g_sherf_debug, kmode, static_cast<uint4*>(nullptr));
    } else if (half_tables) { if (squeezed) SHERF_GATHER(true, 4, true); else if (branchless) SHERF_GATHER(true, 1, true); else SHERF_GATHER(false, 1, true); }
    else { if (squeezed) SHERF_GATHER(true, 4, false); else if (branchless) SHERF_GATHER(true, 1, false); else SHERF_GATHER(false, 1, false); }
#undef SHERF_GATHER
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_gather_tokens(const int32_t* counters, const float* geom, const float* planes_f, int P,
                                   const float* feat_f, int Hf, int Wf, const float* img4, int H, int W,
                                   const sherf_vox_level* levels_host, const float* tok_bias, const float* bounds,
                                   const float* vox_min, const int32_t* vox_sh_host, int mode, int64_t capacity, float* tokens,
                                   float* extras, sherf_stream_t stream) {
    return gather_tokens_impl(counters, geom, planes_f, P, feat_f, Hf, Wf, img4, H, W, levels_host, tok_bias, bounds, vox_min, vox_sh_host, mode,
                              capacity, tokens, extras, nullptr, stream);
}

// sherf_gather_tokens on fp16 tables (`mode | 16`, mode 0 or 1) that ALSO writes the positional encodings of every sample as fp16 MFMA operand
// fragments for sherf_nerf_mlp3_pe (gather_tokens_h8_kernel<false, true>): pefrag = ((capacity + 31) / 32) tiles x 7 KiB.
extern "C" int sherf_gather_tokens_pe(const int32_t* counters, const float* geom, const float* planes_f, int P,
                                      const float* feat_f, int Hf, int Wf, const float* img4, int H, int W,
                                      const sherf_vox_level* levels_host, const float* tok_bias, const float* bounds,
                                      const float* vox_min, const int32_t* vox_sh_host, int mode, int64_t capacity, float* tokens,
                                      float* extras, void* pefrag, sherf_stream_t stream) {
    SHERF_CHECK_ARG(pefrag);
    return gather_tokens_impl(counters, geom, planes_f, P, feat_f, Hf, Wf, img4, H, W, levels_host, tok_bias, bounds, vox_min, vox_sh_host, mode,
                              capacity, tokens, extras, pefrag, stream);
}

extern "C" int sherf_gather_tokens_bwd(const int32_t* counters, const float* geom, const float* d_tokens, int P, int Hf, int Wf,
                                       int H, int W, const sherf_vox_level* levels_host, const float* bounds, const float* vox_min,
                                       const int32_t* vox_sh_host, int64_t capacity, float* d_planes_f, float* d_feat_f,
                                       float* d_rows0, float* d_rows1, float* d_rows2, float* d_tok_bias, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && geom && d_tokens && levels_host && bounds && vox_min && vox_sh_host && d_planes_f && d_feat_f &&
                    d_rows0 && d_rows1 && d_rows2 && d_tok_bias);
    SHERF_CHECK_ARG(P > 0 && Hf > 0 && Wf > 0 && H > 0 && W > 0 && capacity > 0);
    LevelsBwd lv = {};
    for (int i = 0; i < 3; ++i) {
        lv.l[i] = levels_host[i];
        SHERF_CHECK_ARG(lv.l[i].wp && lv.l[i].D > 0 && lv.l[i].H > 0 && lv.l[i].W > 0);
    }
    lv.d_rows[0] = d_rows0; lv.d_rows[1] = d_rows1; lv.d_rows[2] = d_rows2;
    int3 sh = make_int3(vox_sh_host[0], vox_sh_host[1], vox_sh_host[2]);
    const int64_t tiles = (capacity + 31) / 32;
    hipLaunchKernelGGL(gather_tokens_bwd_kernel, dim3((unsigned)(tiles < 16384 ? tiles : 16384)), dim3(256), 0, as_stream(stream), counters,
                       geom, reinterpret_cast<const float4*>(d_tokens), P, Hf, Wf, H, W, lv, bounds, vox_min, sh, capacity,
                       reinterpret_cast<float4*>(d_planes_f), reinterpret_cast<float4*>(d_feat_f), d_tok_bias);
    SHERF_LAUNCH_CHECK();
}

static int bwd_bins(const sherf_vox_level& l2) { return (l2.D + 4) * (l2.H + 4) * (l2.W + 4); }

static int64_t bwd_scratch_words(const sherf_vox_level* levels_host, int64_t capacity) {      // (bins of the finest tapped level: round 5's run order; the coarsest level's need less)
    const int64_t nb = bwd_bins(levels_host[0]) > bwd_bins(levels_host[2]) ? bwd_bins(levels_host[0]) : bwd_bins(levels_host[2]);
    return 4 + 4 * nb + 2 * capacity + (nb + 1023) / 1024 + 4;
}

extern "C" int sherf_gather_bwd_scratch_words(const sherf_vox_level* levels_host, int64_t capacity, int64_t* words_host) {
    SHERF_CHECK_ARG(levels_host && words_host && capacity > 0 && levels_host[2].D > 0 && levels_host[2].H > 0 && levels_host[2].W > 0);
    *words_host = bwd_scratch_words(levels_host, capacity);
    return SHERF_OK;
}

extern "C" int sherf_gather_tokens_bwd_binned(const int32_t* counters, const float* geom, const float* d_tokens, int P, int Hf, int Wf,
                                              int H, int W, const sherf_vox_level* levels_host, const float* bounds, const float* vox_min,
                                              const int32_t* vox_sh_host, int64_t capacity, float* d_planes_f, float* d_feat_f,
                                              float* d_rows0, float* d_rows1, float* d_rows2, float* d_tok_bias, int32_t* scratch,
                                              int64_t scratch_words, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && geom && d_tokens && levels_host && bounds && vox_min && vox_sh_host && d_planes_f && d_feat_f &&
                    d_rows0 && d_rows1 && d_rows2 && d_tok_bias && scratch);
    SHERF_CHECK_ARG(P > 0 && Hf > 0 && Wf > 0 && H > 0 && W > 0 && capacity > 0);
    LevelsBwd lv = {};
    for (int i = 0; i < 3; ++i) {
        lv.l[i] = levels_host[i];
        SHERF_CHECK_ARG(lv.l[i].wp && lv.l[i].D > 0 && lv.l[i].H > 0 && lv.l[i].W > 0);
    }
    SHERF_CHECK_ARG(scratch_words >= bwd_scratch_words(levels_host, capacity));
    lv.d_rows[0] = d_rows0; lv.d_rows[1] = d_rows1; lv.d_rows[2] = d_rows2;
    const int3 sh = make_int3(vox_sh_host[0], vox_sh_host[1], vox_sh_host[2]);
    hipStream_t st = as_stream(stream);
    if (!(sherf_experiment() & 256)) {               // round 5: samples in the order of their finest cell, every level's sums in registers (SHERF_EXPERIMENT bit 8: round 3's kernel)
        SHERF_CHECK_ARG((int64_t)(lv.l[0].D + 4) * (lv.l[0].H + 4) * (lv.l[0].W + 4) < (int64_t)1 << 30);
        const int nb = bwd_bins(lv.l[0]), n_blocks = (nb + 1023) / 1024;
        const BinWs w0 = bin_ws(scratch, nb, capacity);
        SHERF_HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)(4 + 2 * (int64_t)nb) * sizeof(int32_t), st));
        const unsigned sb0 = (unsigned)((capacity + 255) / 256);
        hipLaunchKernelGGL(bin_count_kernel, dim3(sb0), dim3(256), 0, st, counters, geom, lv.l[0], vox_min, sh, capacity, w0);
        hipLaunchKernelGGL(bin_scan_blocks_kernel, dim3(n_blocks), dim3(1024), 0, st, w0);
        hipLaunchKernelGGL(bin_scan_top_kernel, dim3(1), dim3(1024), 0, st, w0, n_blocks);
        hipLaunchKernelGGL(bin_fill_blocks_kernel, dim3(sb0), dim3(256), 0, st, counters, capacity, w0);
        const int64_t chunks = (capacity + 63) / 64;
        const int grid = (int)(chunks / 4 + 1 < 4 * n_cus() ? chunks / 4 + 1 : 4 * n_cus());
        hipLaunchKernelGGL(gather_tokens_bwd_runs_kernel, dim3(grid), dim3(kBinNT), 0, st, counters, capacity, geom, reinterpret_cast<const float4*>(d_tokens),
                           P, Hf, Wf, H, W, lv, bounds, vox_min, sh, w0.sorted, reinterpret_cast<float4*>(d_planes_f), reinterpret_cast<float4*>(d_feat_f),
                           d_tok_bias, g_sherf_debug);
        SHERF_LAUNCH_CHECK();
    }
    const int n_bins = bwd_bins(lv.l[2]);
    const BinWs w = bin_ws(scratch, n_bins, capacity);
    SHERF_HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)(4 + 2 * (int64_t)n_bins) * sizeof(int32_t), st));      // list length, counts, cursors
    const unsigned sb = (unsigned)((capacity + 255) / 256);
    hipLaunchKernelGGL(bin_count_kernel, dim3(sb), dim3(256), 0, st, counters, geom, lv.l[2], vox_min, sh, capacity, w);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, st, w);
    hipLaunchKernelGGL(bin_fill_kernel, dim3(sb), dim3(256), 0, st, counters, capacity, w);
    const int64_t max_list = capacity < n_bins ? capacity : n_bins;
    hipLaunchKernelGGL(gather_tokens_bwd_binned_kernel, dim3((unsigned)(max_list < 2048 ? max_list : 2048)), dim3(kBinNT), 0, st, geom,
                       reinterpret_cast<const float4*>(d_tokens), P, Hf, Wf, H, W, lv, bounds, vox_min, sh, w,
                       reinterpret_cast<float4*>(d_planes_f), reinterpret_cast<float4*>(d_feat_f), d_tok_bias, g_sherf_debug);
    SHERF_LAUNCH_CHECK();
}
