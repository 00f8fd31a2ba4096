// Geometry stage (gfx950): cell list build, fused depth sampling + SMPL-frame transform + exact nearest
// vertex + 5 cm shell mask + deterministic stream compaction, and the per-sample warp.
// Rows a4, a5, a6, a8, a9, a10(geometry) of SURVEY.md section 8.
//
// The reference runs a brute-force K-NN of all R*S samples against 6890 vertices three times
// (renderer.py:315,564,627: 1.16e11 pair tests at 512x512x64).  Here a uniform cell list with 5 cm cells
// bounds the exact search to a 3x3x3 neighbourhood; the second K-NN is redundant (same id) and the third
// is an exact ball query seeded by the same-index T-pose vertex.
#include "common.h"

namespace {

constexpr float kThresh2 = (float)(0.05 * 0.05);   // renderer.py:318 (python float -> fp32 on comparison)

// (v - Th) @ R, evaluated without contraction in the order ((a0*R0c + a1*R1c) + a2*R2c)
__device__ __forceinline__ void to_smpl_frame(float x, float y, float z, const float* __restrict__ R,
                                              const float* __restrict__ Th, float& ox, float& oy, float& oz) {
    float a0 = __fsub_rn(x, Th[0]), a1 = __fsub_rn(y, Th[1]), a2 = __fsub_rn(z, Th[2]);
    ox = __fadd_rn(__fadd_rn(__fmul_rn(a0, R[0]), __fmul_rn(a1, R[3])), __fmul_rn(a2, R[6]));
    oy = __fadd_rn(__fadd_rn(__fmul_rn(a0, R[1]), __fmul_rn(a1, R[4])), __fmul_rn(a2, R[7]));
    oz = __fadd_rn(__fadd_rn(__fmul_rn(a0, R[2]), __fmul_rn(a1, R[5])), __fmul_rn(a2, R[8]));
}
__device__ __forceinline__ void rot_only(float x, float y, float z, const float* __restrict__ R, float& ox,
                                         float& oy, float& oz) {
    ox = __fadd_rn(__fadd_rn(__fmul_rn(x, R[0]), __fmul_rn(y, R[3])), __fmul_rn(z, R[6]));
    oy = __fadd_rn(__fadd_rn(__fmul_rn(x, R[1]), __fmul_rn(y, R[4])), __fmul_rn(z, R[7]));
    oz = __fadd_rn(__fadd_rn(__fmul_rn(x, R[2]), __fmul_rn(y, R[5])), __fmul_rn(z, R[8]));
}

// ---------------------------------------------------------------------------------------------
// cell list: one block of 1024 threads (n <= 6890 points)
// ---------------------------------------------------------------------------------------------
constexpr int kLdsCells = 24576;      // cell counters that fit the LDS fast path of build_cells_body (96 KiB)

__device__ __forceinline__ void build_cells_body(const float* __restrict__ verts, int n,
                                                           const float* __restrict__ R, const float* __restrict__ Th,
                                                           float cell_size, float* __restrict__ hdr,
                                                           int32_t* __restrict__ cell_start, float4* __restrict__ cell_pts,
                                                           int32_t* __restrict__ scratch, uint32_t* __restrict__ near_mask) {
    __shared__ float red[6][1024 / 64];
    __shared__ float s_hdr[kGridHdr];
    __shared__ int s_part[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* pos = reinterpret_cast<float*>(scratch) ;            // [n][3] transformed positions
    int32_t* cid = scratch + 3 * n;                             // [n]
    int32_t* rank = scratch + 4 * n;                            // [n]
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < n; i += 1024) {
        float x = verts[i * 3], y = verts[i * 3 + 1], z = verts[i * 3 + 2];
        if (R) to_smpl_frame(x, y, z, R, Th, x, y, z);
        pos[i * 3] = x; pos[i * 3 + 1] = y; pos[i * 3 + 2] = z;
        mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
    }
    for (int c = 0; c < 3; ++c) {
        float a = mn[c], b = mx[c];
        for (int off = 32; off > 0; off >>= 1) { a = fminf(a, __shfl_xor(a, off)); b = fmaxf(b, __shfl_xor(b, off)); }
        if (lane == 0) { red[c][wave] = a; red[3 + c][wave] = b; }
    }
    __syncthreads();
    if (tid == 0) {
        float lo[3], hi[3];
        for (int c = 0; c < 3; ++c) {
            lo[c] = red[c][0]; hi[c] = red[3 + c][0];
            for (int w = 1; w < 16; ++w) { lo[c] = fminf(lo[c], red[c][w]); hi[c] = fmaxf(hi[c], red[3 + c][w]); }
        }
        float cell = cell_size;
        for (int c = 0; c < 3; ++c) cell = fmaxf(cell, (hi[c] - lo[c]) / 60.0f);   // keep every axis <= 64 cells
        float inv = 1.0f / cell;
        int dims[3];
        for (int c = 0; c < 3; ++c) dims[c] = min(64, (int)floorf((hi[c] - lo[c]) * inv) + 3);
        s_hdr[0] = lo[0] - cell; s_hdr[1] = lo[1] - cell; s_hdr[2] = lo[2] - cell; s_hdr[3] = cell; s_hdr[4] = inv;
        s_hdr[5] = __int_as_float(dims[0]); s_hdr[6] = __int_as_float(dims[1]); s_hdr[7] = __int_as_float(dims[2]);
        const int sub = (8 * dims[0] * dims[1] * dims[2] <= 16384 * 32) ? 2 : 1;   // refined mask must fit 64 KiB of LDS
        s_hdr[8] = __int_as_float(sub); s_hdr[9] = s_hdr[10] = s_hdr[11] = 0.f;
        for (int i = 0; i < kGridHdr; ++i) hdr[i] = s_hdr[i];
    }
    __syncthreads();
    const float ox = s_hdr[0], oy = s_hdr[1], oz = s_hdr[2], inv = s_hdr[4];
    const int nx = __float_as_int(s_hdr[5]), ny = __float_as_int(s_hdr[6]), nz = __float_as_int(s_hdr[7]);
    const int ncell = nx * ny * nz;
    if (ncell + 1 <= kLdsCells && n <= 8 * 1024) {
        // Usual case (a body is ~10-16 K cells of 5 cm): counts, ranks and the scan stay in LDS and the points in registers
        // -- the global-memory version below is a chain of ~10 dependent round trips on a single workgroup.
        __shared__ int s_cnt[kLdsCells];
        for (int i = tid; i <= ncell; i += 1024) s_cnt[i] = 0;
        __syncthreads();
        constexpr int kPer = 8;                                  // points per thread (n <= 8192)
        float px[kPer], py[kPer], pz[kPer];
        int pc[kPer], pr[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int i = tid + k * 1024;
            pc[k] = -1;
            if (i < n) {
                px[k] = pos[i * 3]; py[k] = pos[i * 3 + 1]; pz[k] = pos[i * 3 + 2];       // written by this same thread above
                const int cx = min(nx - 1, max(0, (int)floorf((px[k] - ox) * inv)));
                const int cy = min(ny - 1, max(0, (int)floorf((py[k] - oy) * inv)));
                const int cz = min(nz - 1, max(0, (int)floorf((pz[k] - oz) * inv)));
                pc[k] = (cz * ny + cy) * nx + cx;
                pr[k] = atomicAdd(&s_cnt[pc[k]], 1);
            }
        }
        __syncthreads();
        const int seg = (ncell + 1 + 1023) / 1024;
        const int s0 = tid * seg, s1 = min(ncell + 1, s0 + seg);
        int sum = 0;
        for (int i = s0; i < s1; ++i) sum += s_cnt[i];
        s_part[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int v = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        int run = s_part[tid] - sum;
        for (int i = s0; i < s1; ++i) { const int c = s_cnt[i]; s_cnt[i] = run; run += c; }
        __syncthreads();
        for (int i = tid; i <= ncell; i += 1024) cell_start[i] = s_cnt[i];
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (pc[k] >= 0) cell_pts[s_cnt[pc[k]] + pr[k]] = make_float4(px[k], py[k], pz[k], __int_as_float(tid + k * 1024));
        return;
    }
    for (int i = tid; i <= ncell; i += 1024) cell_start[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        int cx = min(nx - 1, max(0, (int)floorf((pos[i * 3] - ox) * inv)));
        int cy = min(ny - 1, max(0, (int)floorf((pos[i * 3 + 1] - oy) * inv)));
        int cz = min(nz - 1, max(0, (int)floorf((pos[i * 3 + 2] - oz) * inv)));
        int c = (cz * ny + cy) * nx + cx;
        cid[i] = c;
        rank[i] = atomicAdd(&cell_start[c], 1);
    }
    __syncthreads();
    // exclusive scan of cell_start[0..ncell] in place: per-thread segments + block scan of the partials
    const int seg = (ncell + 1 + 1023) / 1024;
    const int s0 = tid * seg, s1 = min(ncell + 1, s0 + seg);
    int sum = 0;
    // counts were produced by global atomics (performed at L2): read them L1-bypassing
    for (int i = s0; i < s1; ++i) sum += __hip_atomic_load(&cell_start[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int i = s0; i < s1; ++i) {
        int c = __hip_atomic_load(&cell_start[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cell_start[i] = run; run += c;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024)
        cell_pts[cell_start[cid[i]] + rank[i]] = make_float4(pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2], __int_as_float(i));
}

__global__ void __launch_bounds__(1024) build_cells_kernel(const float* __restrict__ verts, int n, const float* __restrict__ R,
                                                           const float* __restrict__ Th, float cell_size, float* __restrict__ hdr,
                                                           int32_t* __restrict__ cell_start, float4* __restrict__ cell_pts,
                                                           int32_t* __restrict__ scratch, uint32_t* __restrict__ near_mask) {
    build_cells_body(verts, n, R, Th, cell_size, hdr, cell_start, cell_pts, scratch, near_mask);
}

// both per-frame cell lists in one launch: block 0 = posed vertices in the SMPL frame (+ near mask), block 1 = T-pose vertices
__global__ void __launch_bounds__(1024) build_cells2_kernel(const float* __restrict__ verts_a, const float* __restrict__ R_a,
                                                            const float* __restrict__ Th_a, const float* __restrict__ verts_b, int n,
                                                            float cell_size, float* __restrict__ hdr, int32_t* __restrict__ cell_start,
                                                            float4* __restrict__ cell_pts, int32_t* __restrict__ scratch,
                                                            uint32_t* __restrict__ near_mask) {
    if (blockIdx.x == 0)
        build_cells_body(verts_a, n, R_a, Th_a, cell_size, hdr, cell_start, cell_pts, scratch, near_mask);
    else
        build_cells_body(verts_b, n, nullptr, nullptr, cell_size, hdr + kGridHdr, cell_start + (SHERF_MAX_CELLS + 1), cell_pts + n,
                         scratch + 5 * n, nullptr);
}

// near mask: 1 bit per sub-cell (edge cell/sub) whose box comes within the query radius of some vertex; sub = 2 when the
// refined grid fits 2^19 bits (64 KiB of LDS), else 1 (<= 2^18 bits).  Each block marks its 256 vertices in an LDS copy of the mask (fast,
// heavily contended atomics stay on-chip) and ORs only its non-zero words into the global mask.
__global__ void __launch_bounds__(256) near_mask_kernel(const float* __restrict__ pos, int n, const float* __restrict__ hdr,
                                                        float radius, uint32_t* __restrict__ near_mask) {
    extern __shared__ uint32_t s_mask[];
    const CellGrid g = load_grid(hdr);
    const int sub = g.sub, snx = g.nx * sub, sny = g.ny * sub, snz = g.nz * sub;
    const int words = (snx * sny * snz + 31) / 32;
    for (int w = threadIdx.x; w < words; w += 256) s_mask[w] = 0u;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float px = pos[i * 3], py = pos[i * 3 + 1], pz = pos[i * 3 + 2];
        const float cs = g.cell / (float)sub, rad = radius + 1e-3f * cs, rad2 = rad * rad, fs = g.inv_cell * (float)sub;
        const int sx = (int)floorf((px - g.ox) * fs), sy = (int)floorf((py - g.oy) * fs), sz = (int)floorf((pz - g.oz) * fs);
        for (int dz = -sub; dz <= sub; ++dz)
            for (int dy = -sub; dy <= sub; ++dy)
                for (int dx = -sub; dx <= sub; ++dx) {
                    const int qx = sx + dx, qy = sy + dy, qz = sz + dz;
                    if (qx < 0 || qx >= snx || qy < 0 || qy >= sny || qz < 0 || qz >= snz) continue;
                    const float bx = g.ox + qx * cs, by = g.oy + qy * cs, bz = g.oz + qz * cs;
                    const float ex = fmaxf(fmaxf(bx - px, px - (bx + cs)), 0.f), ey = fmaxf(fmaxf(by - py, py - (by + cs)), 0.f),
                                ez = fmaxf(fmaxf(bz - pz, pz - (bz + cs)), 0.f);
                    if (ex * ex + ey * ey + ez * ez < rad2) {
                        const int q = (qz * sny + qy) * snx + qx;
                        atomicOr(&s_mask[q >> 5], 1u << (q & 31));
                    }
                }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < words; w += 256) {
        const uint32_t m = s_mask[w];
        if (m) atomicOr(&near_mask[w], m);
    }
}

// ---------------------------------------------------------------------------------------------
// Near lists (round 4): for every sub-cell of the near mask, the EXACT set of vertices that can be within the query radius of a point of
// that sub-cell (box distance < radius + margin: the near mask's own criterion, so "bit set" == "list not empty").  A candidate sample
// then tests ONE contiguous list (40 vertices on average on a body) instead of walking the nine trimmed x-rows of its 3 x 3 x 3 cell
// neighbourhood (75 on average), and finding point t of the walk no longer takes a nine-way select per point: the candidate search was
// bound by exactly those VALU instructions (142 per four points; profiles/r04_*near_lists*).  Entries are u16 indices into cell_pts
// (n <= 65535), every list starts on a multiple of four entries (one 8-byte load = four entries); the order inside a list is whatever
// the atomics made it -- the search takes the lexicographic minimum of (d^2, vertex id), which no order changes.
//   near_hdr[q] = (start, count) per sub-cell q (zeroed by the caller), near_list[list_cap] u16, cursor = one word (zeroed).
// Three launches: count (vertex-centric, like near_mask_kernel), allocate (+ the near mask from the counts), fill.
// ---------------------------------------------------------------------------------------------
template <bool FILL>
__global__ void __launch_bounds__(256) near_lists_pairs_kernel(const float4* __restrict__ pts, int n, const float* __restrict__ hdr,
                                                               float radius, int2* __restrict__ near_hdr,
                                                               uint16_t* __restrict__ near_list, int64_t list_cap, int32_t* __restrict__ cursor) {
    const CellGrid g = load_grid(hdr);
    const int sub = g.sub, snx = g.nx * sub, sny = g.ny * sub, snz = g.nz * sub;
    // (the counting pass also clears the allocation cursor behind the last header -- two words the caller's memset leaves out so that it is whole 16-byte
    //  blocks: one fill kernel instead of two at the head of the ray side's chain; the cursor is first read by near_lists_alloc_kernel, a launch later)
    if (!FILL && cursor && blockIdx.x == 0 && threadIdx.x == 0) { cursor[0] = 0; cursor[1] = 0; }
    // one vertex per 8 lanes x ... : a vertex's (2 sub + 1)^3 <= 125 sub-cells are spread over the 32 threads of its group
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 5, t = threadIdx.x & 31;
    if (i >= n) return;
    const float4 p = pts[i];
    const float cs = g.cell / (float)sub, rad = radius + 1e-3f * cs, rad2 = rad * rad, fs = g.inv_cell * (float)sub;
    const int sx = (int)floorf((p.x - g.ox) * fs), sy = (int)floorf((p.y - g.oy) * fs), sz = (int)floorf((p.z - g.oz) * fs);
    const int w = 2 * sub + 1, w3 = w * w * w;
    for (int e = t; e < w3; e += 32) {
        const int dz = e / (w * w) - sub, dy = (e / w) % w - sub, dx = e % w - sub;
        const int qx = sx + dx, qy = sy + dy, qz = sz + dz;
        if (qx < 0 || qx >= snx || qy < 0 || qy >= sny || qz < 0 || qz >= snz) continue;
        const float bx = g.ox + qx * cs, by = g.oy + qy * cs, bz = g.oz + qz * cs;
        const float ex = fmaxf(fmaxf(bx - p.x, p.x - (bx + cs)), 0.f), ey = fmaxf(fmaxf(by - p.y, p.y - (by + cs)), 0.f),
                    ez = fmaxf(fmaxf(bz - p.z, p.z - (bz + cs)), 0.f);
        if (ex * ex + ey * ey + ez * ez < rad2) {
            const int q = (qz * sny + qy) * snx + qx;
            if (!FILL) atomicAdd(&near_hdr[q].y, 1);
            else {
                const int start = near_hdr[q].x;                      // (requested before the atomic's round trip, not behind it)
                const int64_t at = (int64_t)start + atomicAdd(&near_hdr[q].y, 1);
                if (at < list_cap) near_list[at] = (uint16_t)i;
            }
        }
    }
}

// thread per sub-cell: its list's start from ONE atomic per wave (lists padded to whole quads), the count reset for the fill pass to
// count up again, and the near mask's word straight from the ballot
__global__ void __launch_bounds__(256) near_lists_alloc_kernel(const float* __restrict__ hdr, int2* __restrict__ near_hdr, int32_t* __restrict__ cursor,
                                                               int64_t list_cap, uint32_t* __restrict__ near_mask) {
    const CellGrid g = load_grid(hdr);
    const int nsub = g.nx * g.ny * g.nz * g.sub * g.sub * g.sub;
    if ((int)blockIdx.x * 256 >= ((nsub + 63) & ~63)) return;         // (whole workgroups beyond the grid; the launch covers SHERF_NEAR_SUBCELLS)
    const int q = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const int c = q < nsub ? near_hdr[q].y : 0;
    const int pc = (c + 3) & ~3;
    int incl = pc;
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    const int total = __shfl(incl, 63);
    int base = 0;
    if (lane == 0 && total) base = atomicAdd(cursor, total);
    base = __shfl(base, 0);
    const unsigned long long any = __ballot(c > 0);
    if (near_mask && (lane & 31) == 0) near_mask[q >> 5] = (uint32_t)(any >> (lane & 32));
    if (q < nsub) {
        const int64_t start = (int64_t)base + (incl - pc);
        near_hdr[q] = start + pc <= list_cap ? make_int2((int)start, 0) : make_int2(0, 0);      // (cannot happen with the documented list_cap)
    }
}

// ---------------------------------------------------------------------------------------------
// pass 1: one wave per ray: depths, positions, exact NN within 5 cm, validity mask
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float depth_at(float near, float range, int k, int S) {
    // math_utils.py:101-118: steps = arange(S)/(S-1);  t = near + steps * (far - near)
    float step = __fdiv_rn((float)k, (float)(S - 1));
    return __fadd_rn(near, __fmul_rn(step, range));
}

// One wave per ray.  Candidate samples (near-mask hit: ~12 % of the samples, ~30 on a ray that crosses the body) are searched by
// EIGHT-LANE GROUPS, eight candidates at a time: each candidate lane first fetches the nine x-contiguous point segments of its
// 3x3x3 cell neighbourhood, trimmed to the cells its 5 cm ball reaches (18 independent loads, all candidates in parallel), and parks
// them with its position in an LDS record;
// then group g of a round takes candidate 8 r + g, its 8 lanes walk the concatenated segments 8 points per step with every step's
// loads in flight before the first distance is evaluated, and the lexicographic minimum of (d^2, vertex id) is taken with a 64-bit
// LDS atomic min per group (d^2 >= 0, so the IEEE bit pattern orders like the value; only points inside the 5 cm threshold ever
// reach the atomic).  Round 1 searched ONE candidate per step with all 64 lanes: one dependent L2 round trip per candidate,
// ~30 per wave, was what the kernel's 450 us consisted of; a round of eight costs about the same trip.
// The record is 14 dwords (segment starts and cumulative counts as u16: at most 65535 points, enforced where the cells are built):
// four waves' records take 14 KiB of LDS, eight waves per SIMD are resident (24-dword records: six).  More resident rays is what
// this latency-bound kernel wants -- round 2 also tried the opposite trade, persistent workgroups holding the whole vertex table
// in LDS (no L2 trips in the rounds, but 8-16 waves per CU): 1.0-1.5 ms (profiles/r02_kernel_trace_v5_lds_sampler_rejected.txt).
struct Cand { float x, y, z; uint16_t s[9]; uint16_t cum[9]; unsigned long long key; };
static_assert(sizeof(Cand) == 56, "candidate record layout");
constexpr int kMaxPtsUnroll = 4;                   // point loads in flight per lane and step group (8 lanes x 4 = 32 points)

template <int NCH>
__global__ void __launch_bounds__(256) sample_nn_kernel(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                        const float* __restrict__ near, const float* __restrict__ far,
                                                        int R, int S, const float* __restrict__ Rg,
                                                        const float* __restrict__ Th, const float* __restrict__ hdr,
                                                        const int32_t* __restrict__ cell_start,
                                                        const float4* __restrict__ cell_pts,
                                                        const uint32_t* __restrict__ near_mask,
                                                        int32_t* __restrict__ ray_cnt, uint64_t* __restrict__ ray_mask,
                                                        int32_t* __restrict__ dense_vid, int dbg) {
    __shared__ Cand s_cand[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= R) return;
    const CellGrid g = load_grid(hdr);
    const float o0 = ray_o[ray * 3], o1 = ray_o[ray * 3 + 1], o2 = ray_o[ray * 3 + 2];
    const float d0 = ray_d[ray * 3], d1 = ray_d[ray * 3 + 1], d2 = ray_d[ray * 3 + 2];
    const float nr = near[ray], range = __fsub_rn(far[ray], nr);
    const unsigned long long kInit = ((unsigned long long)__float_as_uint(kThresh2) << 32) | 0x7FFFFFFFull;
    Cand* const rec = s_cand[wave];
    const int grp = lane >> 3, sub = lane & 7;
    int total = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int k = ch * 64 + lane;
        bool cand = false;
        float xs = 0.f, ys = 0.f, zs = 0.f;
        int cx = 0, cy = 0, cz = 0;
        if (k < S) {
            float t = depth_at(nr, range, k, S);
            float x = __fadd_rn(o0, __fmul_rn(t, d0)), y = __fadd_rn(o1, __fmul_rn(t, d1)), z = __fadd_rn(o2, __fmul_rn(t, d2));
            to_smpl_frame(x, y, z, Rg, Th, xs, ys, zs);
            // quick reject: an unset near-mask bit proves that no vertex lies within 5 cm of this sample
            const float fs = g.inv_cell * (float)g.sub;
            const int sx = (int)floorf((xs - g.ox) * fs), sy = (int)floorf((ys - g.oy) * fs), sz = (int)floorf((zs - g.oz) * fs);
            cx = g.sub == 2 ? sx >> 1 : sx; cy = g.sub == 2 ? sy >> 1 : sy; cz = g.sub == 2 ? sz >> 1 : sz;
            if (sx >= 0 && cx < g.nx && sy >= 0 && cy < g.ny && sz >= 0 && cz < g.nz) {
                const int q = (sz * (g.ny * g.sub) + sy) * (g.nx * g.sub) + sx;
                cand = (((dbg & 2) || ((near_mask[q >> 5] >> (q & 31)) & 1u)) && !(dbg & 1));
            }
        }
        const unsigned long long cmask = __ballot(cand);
        const int ncand = __popcll(cmask);
        const int rank = __popcll(cmask & ((1ull << lane) - 1ull));
        if (cand) {                                  // this candidate's record: position, the 9 segments (start, cumulative count)
            Cand& c = rec[rank];
            c.x = xs; c.y = ys; c.z = zs; c.key = kInit;
            // only the cells the 5 cm ball can reach: a (y, z) row of cells farther than r in the yz plane is skipped, the others are
            // trimmed in x to [x - sqrt(r^2 - d_yz^2), x + ...] (+ a safety margin) -- the 27-cell cube holds ~80 vertices of the body
            // surface, the ball ~ a fifth of them; every vertex within r is still examined, so the minimum is unchanged
            const float r = 0.05f + 1e-4f * g.cell;
            int st[9], en[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int qz = cz + i / 3 - 1, qy = cy + i % 3 - 1;
                const float y0 = g.oy + qy * g.cell, z0 = g.oz + qz * g.cell;
                const float ey = fmaxf(fmaxf(y0 - ys, ys - (y0 + g.cell)), 0.f), ez = fmaxf(fmaxf(z0 - zs, zs - (z0 + g.cell)), 0.f);
                const float rem = r * r - (ey * ey + ez * ez);
                const bool ok = qz >= 0 && qz < g.nz && qy >= 0 && qy < g.ny && rem > 0.f;
                const float rx = sqrtf(fmaxf(rem, 0.f)) + 1e-4f * g.cell;
                const int x0 = max(max((int)floorf((xs - rx - g.ox) * g.inv_cell), cx - 1), 0);
                const int x1 = min(min((int)floorf((xs + rx - g.ox) * g.inv_cell), cx + 1), g.nx - 1);
                const int row = ok ? (qz * g.ny + qy) * g.nx : 0;
                st[i] = cell_start[row + (ok ? x0 : 0)];
                en[i] = ok && x1 >= x0 ? cell_start[row + x1 + 1] : st[i];
            }
            int cum = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) { c.s[i] = (uint16_t)st[i]; cum += en[i] - st[i]; c.cum[i] = (uint16_t)cum; }
        }
        __builtin_amdgcn_wave_barrier();             // (the wave runs in lockstep; the LDS records are ordered by s_waitcnt lgkmcnt)
        for (int c0 = 0; c0 < ncand; c0 += 8) {
            const int ci = c0 + grp;
            if (ci < ncand) {
                Cand& c = rec[ci];
                const float qx = c.x, qy = c.y, qz = c.z;
                int bs[9], cum[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) { bs[i] = c.s[i]; cum[i] = c.cum[i]; }
                const int npts = cum[8];
                for (int base = 0; base < npts; base += 8 * kMaxPtsUnroll) {
                    float4 v[kMaxPtsUnroll];
#pragma unroll
                    for (int u = 0; u < kMaxPtsUnroll; ++u) {
                        const int t = base + u * 8 + sub;
                        int p = bs[0] + t;
#pragma unroll
                        for (int i = 1; i < 9; ++i) p = (t >= cum[i - 1]) ? bs[i] + (t - cum[i - 1]) : p;
                        v[u] = cell_pts[t < npts ? p : 0];
                    }
#pragma unroll
                    for (int u = 0; u < kMaxPtsUnroll; ++u) {
                        const int t = base + u * 8 + sub;
                        const float dd = dist2_exact(qx, qy, qz, v[u].x, v[u].y, v[u].z);
                        if (t < npts && dd < kThresh2)
                            atomicMin(&c.key, ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(v[u].w));
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long my_key = cand ? __hip_atomic_load(&rec[rank].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : kInit;
        __builtin_amdgcn_wave_barrier();             // every result is read before the next chunk's records overwrite them
        const bool valid = (unsigned)(my_key >> 32) < __float_as_uint(kThresh2);
        const uint64_t m = __ballot(valid);
        if (valid) dense_vid[(size_t)ray * S + k] = (int)(my_key & 0x7FFFFFFFull);
        if (lane == 0) ray_mask[(size_t)ray * NCH + ch] = m;
        total += __popcll(m);
    }
    if (lane == 0) ray_cnt[ray] = total;
}

// ---------------------------------------------------------------------------------------------
// Round 3: the same search as TWO passes over a dense candidate list.  sample_nn_kernel above gives every
// ray one wave for the whole job, so a ray through the body (~30 candidates, four dependent search rounds) holds its wave ~10x
// longer than a ray beside it (none): 250 of its 330 us were candidate search at a quarter of the chip's occupancy.  Here
//   pass 1 (cand_mark): positions + near-mask reject for every sample, four rays in flight per wave; a workgroup's candidates
//           (dense index ray * S + k) are staged in LDS and appended to a global list with ONE atomic reservation per workgroup
//           (order is irrelevant: results land in per-ray bit masks);
//   pass 2 (cand_search): the list is walked eight lanes per candidate, 32 candidates per workgroup step, every lane group busy --
//           the same trimmed 3x3x3 cell walk and the same lexicographic (d^2, vertex id) minimum as above, bit for bit;
// then ray counts come from the masks' popcounts inside the first scan kernel.  The list lives in cs_xs (written only by the
// compaction afterwards): `capacity` 16-byte records (x_s, dense index), so the path is taken when that provably holds every sample
// (capacity >= R S, the default workspace).
// ---------------------------------------------------------------------------------------------
constexpr int kMarkRays = 4;          // rays in flight per wave of pass 1

template <int NCH>
__global__ void __launch_bounds__(256) cand_mark_kernel(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                        const float* __restrict__ near, const float* __restrict__ far, int R, int S,
                                                        const float* __restrict__ Rg, const float* __restrict__ Th,
                                                        const float* __restrict__ hdr, const uint32_t* __restrict__ near_mask,
                                                        float4* __restrict__ cand_list, int64_t list_cap, int32_t* __restrict__ cand_count,
                                                        uint64_t* __restrict__ ray_mask, int dbg) {
    constexpr int RPW = 16 / NCH;                    // rays per wave: a workgroup stages at most 4 * RPW * 64 * NCH = 4096 candidates
    __shared__ int s_list[4096];
    __shared__ int s_n, s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const CellGrid g = load_grid(hdr);
    const float fs = g.inv_cell * (float)g.sub;
    const int ray0 = (blockIdx.x * 4 + wave) * RPW;
    for (int rb = 0; rb < RPW; rb += kMarkRays) {
        float o[kMarkRays][3], d[kMarkRays][3], nr[kMarkRays], rg[kMarkRays];
#pragma unroll
        for (int u = 0; u < kMarkRays; ++u) {
            const int ray = min(ray0 + rb + u, R - 1);
            o[u][0] = ray_o[ray * 3]; o[u][1] = ray_o[ray * 3 + 1]; o[u][2] = ray_o[ray * 3 + 2];
            d[u][0] = ray_d[ray * 3]; d[u][1] = ray_d[ray * 3 + 1]; d[u][2] = ray_d[ray * 3 + 2];
            nr[u] = near[ray]; rg[u] = __fsub_rn(far[ray], nr[u]);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int k = ch * 64 + lane;
            uint32_t word[kMarkRays];
            int bit[kMarkRays];
#pragma unroll
            for (int u = 0; u < kMarkRays; ++u) {      // every ray's mask word is requested before the first one is looked at
                bit[u] = -1; word[u] = 0u;
                if (k < S && rb + u < RPW && ray0 + rb + u < R) {
                    const float t = depth_at(nr[u], rg[u], k, S);
                    const float x = __fadd_rn(o[u][0], __fmul_rn(t, d[u][0])), y = __fadd_rn(o[u][1], __fmul_rn(t, d[u][1])),
                                z = __fadd_rn(o[u][2], __fmul_rn(t, d[u][2]));
                    float xs, ys, zs;
                    to_smpl_frame(x, y, z, Rg, Th, xs, ys, zs);
                    const int sx = (int)floorf((xs - g.ox) * fs), sy = (int)floorf((ys - g.oy) * fs), sz = (int)floorf((zs - g.oz) * fs);
                    const int cx = g.sub == 2 ? sx >> 1 : sx, cy = g.sub == 2 ? sy >> 1 : sy, cz = g.sub == 2 ? sz >> 1 : sz;
                    if (sx >= 0 && cx < g.nx && sy >= 0 && cy < g.ny && sz >= 0 && cz < g.nz) {
                        const int q = (sz * (g.ny * g.sub) + sy) * (g.nx * g.sub) + sx;
                        bit[u] = q & 31;
                        word[u] = near_mask[q >> 5];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kMarkRays; ++u) {
                const int ray = ray0 + rb + u;
                const bool cand = bit[u] >= 0 && (((dbg & 2) || ((word[u] >> bit[u]) & 1u)) && !(dbg & 1));
                const unsigned long long cm = __ballot(cand);
                if (rb + u < RPW && ray < R && lane == 0) ray_mask[(size_t)ray * NCH + ch] = 0ull;       // pass 2 ORs the valid bits in
                if (cm) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_n, __popcll(cm));
                    base = __shfl(base, 0);
                    if (cand) s_list[base + __popcll(cm & ((1ull << lane) - 1ull))] = ray * S + k;
                }
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (threadIdx.x == 0) s_base = n ? atomicAdd(cand_count, n) : 0;
    __syncthreads();
    // a candidate's record = (x_s, its dense index): the position is evaluated HERE, one thread per candidate, with the same operations the
    // search used to repeat in all eight lanes of a group behind two more dependent loads (its list entry, then its ray) -- round 4
    const int64_t gb = s_base;
    for (int i = threadIdx.x; i < n; i += 256)
        if (gb + i < list_cap) {
            const int idx = s_list[i];
            const int ray = idx / S, k = idx - ray * S;
            const float nr1 = near[ray], t = depth_at(nr1, __fsub_rn(far[ray], nr1), k, S);
            const float x = __fadd_rn(ray_o[ray * 3], __fmul_rn(t, ray_d[ray * 3])), y = __fadd_rn(ray_o[ray * 3 + 1], __fmul_rn(t, ray_d[ray * 3 + 1])),
                        z = __fadd_rn(ray_o[ray * 3 + 2], __fmul_rn(t, ray_d[ray * 3 + 2]));
            float xs, ys, zs;
            to_smpl_frame(x, y, z, Rg, Th, xs, ys, zs);
            cand_list[gb + i] = make_float4(xs, ys, zs, __int_as_float(idx));
        }
}

// candidates in flight per eight-lane group.  Two (every stage's loads issued for both before either is consumed) was measured in round 3:
// 232 us against 209 us with one (profiles/r03_bench_f_sconv8_search2.txt) -- the extra registers cost more occupancy than the second
// chain hides.
constexpr int kSearchU = 1;

template <int NCH>
__global__ void __launch_bounds__(256) cand_search_kernel(const float4* __restrict__ cand_list, int64_t list_cap,
                                                          const int32_t* __restrict__ cand_count, int S, const float* __restrict__ hdr,
                                                          const int32_t* __restrict__ cell_start, const float4* __restrict__ cell_pts,
                                                          unsigned long long* __restrict__ ray_mask, int32_t* __restrict__ dense_vid) {
    constexpr int U = kSearchU;
    __shared__ int s_seg[U][32][20];                 // per candidate of the step: 9 segment starts, 9 counts
    const int grp = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const CellGrid g = load_grid(hdr);
    const int64_t n = min((int64_t)*cand_count, list_cap);
    const float r = 0.05f + 1e-4f * g.cell;
    const unsigned long long kInit = ((unsigned long long)__float_as_uint(kThresh2) << 32) | 0x7FFFFFFFull;
    // the NEXT step's records are requested before this step's dependent chain (cell rows -> points) starts: one round trip off the chain
    float4 rec_next[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t ci = (int64_t)blockIdx.x * 32 * U + u * 32 + grp;
        rec_next[u] = cand_list[ci < n ? ci : 0];
    }
    for (int64_t c0 = (int64_t)blockIdx.x * 32 * U; c0 < n; c0 += (int64_t)gridDim.x * 32 * U) {
        bool live[U];
        int idx[U], ray[U], k[U];
        float xs[U], ys[U], zs[U];
        // every load of a stage is issued for all U candidates before any of them is consumed: U independent chains per lane group
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t ci = c0 + u * 32 + grp;
            live[u] = ci < n;
            const float4 rec = rec_next[u];                            // (x_s, dense index): written by cand_mark
            const int64_t cn = ci + (int64_t)gridDim.x * 32 * U;
            rec_next[u] = cand_list[cn < n ? cn : 0];
            xs[u] = rec.x; ys[u] = rec.y; zs[u] = rec.z;
            idx[u] = live[u] ? __float_as_int(rec.w) : 0;
            ray[u] = idx[u] / S; k[u] = idx[u] - ray[u] * S;
        }
        // the nine x-contiguous point segments of the 3x3x3 neighbourhood, trimmed to what the 5 cm ball reaches (as in sample_nn_kernel):
        // lane `sub` of the group prepares row `sub`, lane 0 also row 8
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cx = (int)floorf((xs[u] - g.ox) * g.inv_cell), cy = (int)floorf((ys[u] - g.oy) * g.inv_cell), cz = (int)floorf((zs[u] - g.oz) * g.inv_cell);
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int i = pass ? 8 : sub;
                if (pass && sub != 0) break;
                const int qz = cz + i / 3 - 1, qy = cy + i % 3 - 1;
                const float y0 = g.oy + qy * g.cell, z0 = g.oz + qz * g.cell;
                const float ey = fmaxf(fmaxf(y0 - ys[u], ys[u] - (y0 + g.cell)), 0.f), ez = fmaxf(fmaxf(z0 - zs[u], zs[u] - (z0 + g.cell)), 0.f);
                const float rem = r * r - (ey * ey + ez * ez);
                const bool ok = live[u] && qz >= 0 && qz < g.nz && qy >= 0 && qy < g.ny && rem > 0.f;
                const float rx = sqrtf(fmaxf(rem, 0.f)) + 1e-4f * g.cell;
                const int x0 = max(max((int)floorf((xs[u] - rx - g.ox) * g.inv_cell), cx - 1), 0);
                const int x1 = min(min((int)floorf((xs[u] + rx - g.ox) * g.inv_cell), cx + 1), g.nx - 1);
                const int row = ok ? (qz * g.ny + qy) * g.nx : 0;
                const int st = cell_start[row + (ok ? x0 : 0)];
                const int en = ok && x1 >= x0 ? cell_start[row + x1 + 1] : st;
                s_seg[u][grp][i] = st; s_seg[u][grp][9 + i] = en - st;
            }
        }
        __builtin_amdgcn_wave_barrier();             // (a group's eight lanes sit in one wave, which runs in lockstep; LDS ordered by lgkmcnt)
        int bs[U][9], cum[U][9], npts[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int run = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) { bs[u][i] = s_seg[u][grp][i]; run += s_seg[u][grp][9 + i]; cum[u][i] = run; }
            npts[u] = run;
        }
        __builtin_amdgcn_wave_barrier();             // every lane has read the step's segments before the next step rewrites them
        unsigned long long key[U];
        int most = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) { key[u] = kInit; most = max(most, npts[u]); }
        for (int base = 0; base < most; base += 8 * kMaxPtsUnroll) {
            float4 v[U][kMaxPtsUnroll];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int w = 0; w < kMaxPtsUnroll; ++w) {
                    const int tt = base + w * 8 + sub;
                    int p = bs[u][0] + tt;
#pragma unroll
                    for (int i = 1; i < 9; ++i) p = (tt >= cum[u][i - 1]) ? bs[u][i] + (tt - cum[u][i - 1]) : p;
                    v[u][w] = cell_pts[tt < npts[u] ? p : 0];
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int w = 0; w < kMaxPtsUnroll; ++w) {
                    const int tt = base + w * 8 + sub;
                    const float dd = dist2_exact(xs[u], ys[u], zs[u], v[u][w].x, v[u][w].y, v[u][w].z);
                    if (tt < npts[u] && dd < kThresh2) {
                        const unsigned long long cand = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(v[u][w].w);
                        key[u] = cand < key[u] ? cand : key[u];
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {  // lexicographic (d^2, vertex id) minimum over the group's eight lanes
                const unsigned lo = __shfl_xor((unsigned)key[u], off), hi = __shfl_xor((unsigned)(key[u] >> 32), off);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key[u] = other < key[u] ? other : key[u];
            }
            if (live[u] && sub == 0 && (unsigned)(key[u] >> 32) < __float_as_uint(kThresh2)) {
                dense_vid[idx[u]] = (int)(key[u] & 0x7FFFFFFFull);
                atomicOr(&ray_mask[(size_t)ray[u] * NCH + (k[u] >> 6)], 1ull << (k[u] & 63));
            }
        }
    }
}

// first scan kernel of the two-pass path: a ray's count is the popcount of its masks (also stored, the compositing reads ray_cnt)
// The candidate search over the near lists (see near_lists_pairs_kernel): eight lanes per candidate as in cand_search_kernel, a lane
// takes four list entries per step (one 8-byte load) and their four points; the next step's record and list header are in flight while
// this step's points are compared.  Same comparisons on the same points' bits, fewer of them: results identical.
template <int NCH>
__global__ void __launch_bounds__(256) cand_search_lists_kernel(const float4* __restrict__ cand_list, int64_t list_cap,
                                                                const int32_t* __restrict__ cand_count, int S, const float* __restrict__ hdr,
                                                                const int2* __restrict__ near_hdr, const uint16_t* __restrict__ near_list,
                                                                const float4* __restrict__ cell_pts,
                                                                unsigned long long* __restrict__ ray_mask, int32_t* __restrict__ dense_vid) {
    const int grp = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const CellGrid g = load_grid(hdr);
    const int snx = g.nx * g.sub, sny = g.ny * g.sub, snz = g.nz * g.sub;
    const float fs = g.inv_cell * (float)g.sub;
    const int64_t n = min((int64_t)*cand_count, list_cap);
    const unsigned long long kInit = ((unsigned long long)__float_as_uint(kThresh2) << 32) | 0x7FFFFFFFull;
    const int64_t stride = (int64_t)gridDim.x * 32;
    auto sub_cell = [&](const float4& rec) -> int {                   // cand_mark's own arithmetic on the same floats -> the same sub-cell
        const int sx = (int)floorf((rec.x - g.ox) * fs), sy = (int)floorf((rec.y - g.oy) * fs), sz = (int)floorf((rec.z - g.oz) * fs);
        const bool in = sx >= 0 && sx < snx && sy >= 0 && sy < sny && sz >= 0 && sz < snz;
        return in ? (sz * sny + sy) * snx + sx : -1;
    };
    int64_t ci = (int64_t)blockIdx.x * 32 + grp;
    float4 rec = cand_list[ci < n ? ci : 0];
    float4 rec_next = cand_list[ci + stride < n ? ci + stride : 0];
    int q0 = sub_cell(rec);
    int2 h = near_hdr[ci < n && q0 >= 0 ? q0 : 0];
    for (; ci - grp < n; ci += stride) {                              // (the whole workgroup leaves together: c0 = ci - grp is uniform)
        const bool live = ci < n && sub_cell(rec) >= 0;
        const float xs = rec.x, ys = rec.y, zs = rec.z;
        const int idx = live ? __float_as_int(rec.w) : 0;
        const int st = h.x, cn = live ? h.y : 0;
        // next step: its header can be requested now (its record arrived a step ago), and the record after it
        const int64_t cn1 = ci + stride, cn2 = ci + 2 * stride;
        rec = rec_next;
        const int qn = sub_cell(rec);
        h = near_hdr[cn1 < n && qn >= 0 ? qn : 0];
        rec_next = cand_list[cn2 < n ? cn2 : 0];
        unsigned long long key = kInit;
        for (int base = 0; base < cn; base += 32) {
            const int e = base + sub * 4;
            uint2 w = make_uint2(0u, 0u);
            if (e < cn) w = *reinterpret_cast<const uint2*>(near_list + st + e);
            const int id[4] = {(int)(w.x & 0xFFFFu), (int)(w.x >> 16), (int)(w.y & 0xFFFFu), (int)(w.y >> 16)};
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = cell_pts[e + j < cn ? id[j] : 0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dd = dist2_exact(xs, ys, zs, v[j].x, v[j].y, v[j].z);
                if (e + j < cn && dd < kThresh2) {
                    const unsigned long long cand = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(v[j].w);
                    key = cand < key ? cand : key;
                }
            }
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {      // lexicographic (d^2, vertex id) minimum over the group's eight lanes
            const unsigned lo = __shfl_xor((unsigned)key, off), hi = __shfl_xor((unsigned)(key >> 32), off);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            key = other < key ? other : key;
        }
        if (live && sub == 0 && (unsigned)(key >> 32) < __float_as_uint(kThresh2)) {
            const int ray = idx / S, k = idx - ray * S;
            dense_vid[idx] = (int)(key & 0x7FFFFFFFull);
            atomicOr(&ray_mask[(size_t)ray * NCH + (k >> 6)], 1ull << (k & 63));
        }
    }
}

// The same search one pipeline stage deeper (SHERF_EXPERIMENT bit 14; round 6): cand_search_lists_kernel waits twice per candidate -- for its list entries, then for the points
// they name.  Here the first 64 entries of the NEXT candidate's list are requested while this candidate's points are compared (its header arrived a step ago, the header after it
// and the record after that are requested now), and a lane takes eight entries per step: one dependent wait per candidate for every list of up to 64 vertices (the average is 40).
// Same comparisons on the same points' bits: identical results.
template <int NCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) cand_search_lists2_kernel(const float4* __restrict__ cand_list, int64_t list_cap,
                                                                 const int32_t* __restrict__ cand_count, int S, const float* __restrict__ hdr,
                                                                 const int2* __restrict__ near_hdr, const uint16_t* __restrict__ near_list,
                                                                 const float4* __restrict__ cell_pts,
                                                                 unsigned long long* __restrict__ ray_mask, int32_t* __restrict__ dense_vid) {
    const int grp = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const CellGrid g = load_grid(hdr);
    const int snx = g.nx * g.sub, sny = g.ny * g.sub, snz = g.nz * g.sub;
    const float fs = g.inv_cell * (float)g.sub;
    const int64_t n = min((int64_t)*cand_count, list_cap);
    const unsigned long long kInit = ((unsigned long long)__float_as_uint(kThresh2) << 32) | 0x7FFFFFFFull;
    const int64_t stride = (int64_t)gridDim.x * 32;
    auto sub_cell = [&](const float4& rec) -> int {                   // cand_mark's own arithmetic on the same floats -> the same sub-cell
        const int sx = (int)floorf((rec.x - g.ox) * fs), sy = (int)floorf((rec.y - g.oy) * fs), sz = (int)floorf((rec.z - g.oz) * fs);
        const bool in = sx >= 0 && sx < snx && sy >= 0 && sy < sny && sz >= 0 && sz < snz;
        return in ? (sz * sny + sy) * snx + sx : -1;
    };
    auto entries = [&](int st, int cn, int e, uint2& a, uint2& b) {  // entries e .. e + 7 of a list (lists are padded to four entries)
        a = make_uint2(0u, 0u); b = make_uint2(0u, 0u);
        if (e < cn) a = *reinterpret_cast<const uint2*>(near_list + st + e);
        if (e + 4 < cn) b = *reinterpret_cast<const uint2*>(near_list + st + e + 4);
    };
    int64_t ci = (int64_t)blockIdx.x * 32 + grp;
    float4 r0 = cand_list[ci < n ? ci : 0];
    float4 r1 = cand_list[ci + stride < n ? ci + stride : 0];
    float4 r2 = cand_list[ci + 2 * stride < n ? ci + 2 * stride : 0];
    const int q0 = sub_cell(r0), q1 = sub_cell(r1);
    bool l0 = ci < n && q0 >= 0, l1 = ci + stride < n && q1 >= 0;
    int2 h0 = near_hdr[l0 ? q0 : 0], h1 = near_hdr[l1 ? q1 : 0];
    uint2 wa, wb;
    entries(h0.x, l0 ? h0.y : 0, sub * 8, wa, wb);
    for (; ci - grp < n; ci += stride) {                              // (the whole workgroup leaves together: ci - grp is uniform)
        const bool live = l0;
        const float xs = r0.x, ys = r0.y, zs = r0.z;
        const int idx = live ? __float_as_int(r0.w) : 0;
        const int st = h0.x, cn = live ? h0.y : 0;
        uint2 ca = wa, cb = wb;
        // the stages behind this candidate: entries of the next one, header of the one after it, record of the third
        const int cn1 = l1 ? h1.y : 0;
        entries(h1.x, cn1, sub * 8, wa, wb);
        const int q2 = sub_cell(r2);
        const bool l2 = ci + 2 * stride < n && q2 >= 0;
        const int2 h2 = near_hdr[l2 ? q2 : 0];
        const float4 r3 = cand_list[ci + 3 * stride < n ? ci + 3 * stride : 0];
        unsigned long long key = kInit;
        for (int base = 0; base < cn; base += 64) {
            const int e = base + sub * 8;
            if (base > 0) entries(st, cn, e, ca, cb);
            const int id[8] = {(int)(ca.x & 0xFFFFu), (int)(ca.x >> 16), (int)(ca.y & 0xFFFFu), (int)(ca.y >> 16),
                               (int)(cb.x & 0xFFFFu), (int)(cb.x >> 16), (int)(cb.y & 0xFFFFu), (int)(cb.y >> 16)};
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = cell_pts[e + j < cn ? id[j] : 0];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dd = dist2_exact(xs, ys, zs, v[j].x, v[j].y, v[j].z);
                if (e + j < cn && dd < kThresh2) {
                    const unsigned long long cand = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(v[j].w);
                    key = cand < key ? cand : key;
                }
            }
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {      // lexicographic (d^2, vertex id) minimum over the group's eight lanes
            const unsigned lo = __shfl_xor((unsigned)key, off), hi = __shfl_xor((unsigned)(key >> 32), off);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            key = other < key ? other : key;
        }
        if (live && sub == 0 && (unsigned)(key >> 32) < __float_as_uint(kThresh2)) {
            const int ray = idx / S, k = idx - ray * S;
            dense_vid[idx] = (int)(key & 0x7FFFFFFFull);
            atomicOr(&ray_mask[(size_t)ray * NCH + (k >> 6)], 1ull << (k & 63));
        }
        r0 = r1; r1 = r2; r2 = r3; l0 = l1; l1 = l2; h0 = h1; h1 = h2;
    }
}

template <int NCH>
__global__ void __launch_bounds__(1024) scan_chunk_mask_kernel(const uint64_t* __restrict__ ray_mask, int R, int32_t* __restrict__ cnt,
                                                               int32_t* __restrict__ base, int32_t* __restrict__ chunk_sum) {
    __shared__ int s[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int v = 0;
    if (i < R) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) v += __popcll(ray_mask[(size_t)i * NCH + ch]);
        cnt[i] = v;
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int a = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += a;
        __syncthreads();
    }
    if (i < R) base[i] = s[threadIdx.x] - v;
    if (threadIdx.x == 1023) chunk_sum[blockIdx.x] = s[1023];
}

// ---------------------------------------------------------------------------------------------
// exclusive scan of ray_cnt -> ray_base (chunks of 1024 rays), total -> counters[0]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) scan_chunk_kernel(const int32_t* __restrict__ cnt, int R,
                                                          int32_t* __restrict__ base, int32_t* __restrict__ chunk_sum) {
    __shared__ int s[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int v = i < R ? cnt[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int a = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += a;
        __syncthreads();
    }
    if (i < R) base[i] = s[threadIdx.x] - v;
    if (threadIdx.x == 1023) chunk_sum[blockIdx.x] = s[1023];
}

__global__ void __launch_bounds__(1024) scan_top_kernel(int32_t* __restrict__ chunk_sum, int n_chunks,
                                                        int32_t* __restrict__ counters) {
    __shared__ int s[1024];
    int carry = 0;
    for (int b0 = 0; b0 < n_chunks; b0 += 1024) {
        int i = b0 + threadIdx.x;
        int v = i < n_chunks ? chunk_sum[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int a = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_chunks) chunk_sum[i] = carry + s[threadIdx.x] - v;
        int tot = s[1023];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) counters[0] = carry;
}

// global min / max of all depths (ray_marcher.py:57 clamps with torch.min/max over the whole tensor)
__global__ void __launch_bounds__(256) depth_minmax_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                                           int R, int S, int32_t* __restrict__ counters) {
    __shared__ float smn[4], smx[4];
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
        float nr = near[r], range = __fsub_rn(far[r], nr);
        float t0 = depth_at(nr, range, 0, S), t1 = depth_at(nr, range, S - 1, S);
        mn = fminf(mn, fminf(t0, t1)); mx = fmaxf(mx, fmaxf(t0, t1));
    }
    for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); }
        atomicMin(&counters[1], f2ord(fminf(mn, smn[0])));
        atomicMax(&counters[2], f2ord(fmaxf(mx, smx[0])));
    }
}

// `sticky` (optional, the frame driver's): what the frame that used these counters before left in them is folded in before they are reset -- the largest valid-sample
// count, the OR of the flag words, the number of frames folded -- so that a caller may look at its frames' flags every few frames instead of after every one.
__global__ void init_counters_kernel(int32_t* counters, int32_t* cand_count, int32_t* sticky) {
    if (sticky) { sticky[0] = max(sticky[0], counters[0]); sticky[1] |= counters[3]; sticky[2] += 1; }
    counters[0] = 0; counters[1] = 0x7FFFFFFF; counters[2] = (int32_t)0x80000000; counters[3] = 0;
    *cand_count = 0;
}

// ---------------------------------------------------------------------------------------------
// pass 2: write the compact records in ray-major / ascending-k order (== the reference's boolean-mask order)
// ---------------------------------------------------------------------------------------------
template <int NCH>
__global__ void __launch_bounds__(256) compact_kernel(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                      const float* __restrict__ near, const float* __restrict__ far,
                                                      int R, int S, const float* __restrict__ Rg, const float* __restrict__ Th,
                                                      const int32_t* __restrict__ ray_base_local,
                                                      const int32_t* __restrict__ chunk_off, const uint64_t* __restrict__ ray_mask,
                                                      const int32_t* __restrict__ dense_vid, int64_t capacity,
                                                      int32_t* __restrict__ ray_base, int32_t* __restrict__ cs_idx,
                                                      int32_t* __restrict__ cs_vid, float4* __restrict__ cs_xs,
                                                      const float4* __restrict__ cell_pts_unused) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    int base = ray_base_local[ray] + chunk_off[ray >> 10];
    if (lane == 0) ray_base[ray] = base;
    uint64_t any = 0;                                  // most rays miss the body: they leave before the eight ray loads
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) any |= ray_mask[(size_t)ray * NCH + ch];
    if (any == 0) return;
    const float o0 = ray_o[ray * 3], o1 = ray_o[ray * 3 + 1], o2 = ray_o[ray * 3 + 2];
    const float d0 = ray_d[ray * 3], d1 = ray_d[ray * 3 + 1], d2 = ray_d[ray * 3 + 2];
    const float nr = near[ray], range = __fsub_rn(far[ray], nr);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const uint64_t m = ray_mask[(size_t)ray * NCH + ch];
        const int k = ch * 64 + lane;
        if ((m >> lane) & 1ull) {
            int rank = __popcll(m & ((1ull << lane) - 1ull));
            int64_t c = (int64_t)base + rank;
            if (c < capacity) {
                float t = depth_at(nr, range, k, S);
                float x = __fadd_rn(o0, __fmul_rn(t, d0)), y = __fadd_rn(o1, __fmul_rn(t, d1)), z = __fadd_rn(o2, __fmul_rn(t, d2));
                float xs, ys, zs;
                to_smpl_frame(x, y, z, Rg, Th, xs, ys, zs);
                int vid = dense_vid[(size_t)ray * S + k];
                cs_idx[c] = ray * S + k;
                cs_vid[c] = vid;
                cs_xs[c] = make_float4(xs, ys, zs, 0.f);
            }
        }
        base += __popcll(m);
    }
}

// The same compaction with ONE LANE per ray for the part every ray takes (its base offset, its mask words) and a whole wave only for the rays that hold valid
// samples (round 6 experiment, SHERF_EXPERIMENT bit 11).  The idea: compact_kernel above gives every ray a wave -- 262 144 waves at 512 x 512 of which three
// quarters read eight bytes and leave.  Here a wave takes 64 consecutive rays and walks its hit rays one after the other with all 64 lanes.  Same records at the same
// positions (bit-identical cs_idx / cs_vid / cs_xs / ray_base) -- and SLOWER on the MI355X (96 vs 77 us): the 262 144 short waves were never the cost, and ~16 hit rays
// per wave in sequence put ~16 dependent round trips where the wave-per-ray kernel has one.  Kept as the measured counter-example.
template <int NCH>
__global__ void __launch_bounds__(256) compact_rays_kernel(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                           const float* __restrict__ near, const float* __restrict__ far,
                                                           int R, int S, const float* __restrict__ Rg, const float* __restrict__ Th,
                                                           const int32_t* __restrict__ ray_base_local,
                                                           const int32_t* __restrict__ chunk_off, const uint64_t* __restrict__ ray_mask,
                                                           const int32_t* __restrict__ dense_vid, int64_t capacity,
                                                           int32_t* __restrict__ ray_base, int32_t* __restrict__ cs_idx,
                                                           int32_t* __restrict__ cs_vid, float4* __restrict__ cs_xs) {
    const int lane = threadIdx.x & 63;
    const int ray0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;              // this wave's 64 rays
    if (ray0 >= R) return;
    const int my = ray0 + lane;
    int my_base = 0;
    uint64_t my_any = 0;
    if (my < R) {
        my_base = ray_base_local[my] + chunk_off[my >> 10];
        ray_base[my] = my_base;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) my_any |= ray_mask[(size_t)my * NCH + ch];
    }
    uint64_t hits = __ballot(my_any != 0);
    while (hits) {
        const int rl = __ffsll((unsigned long long)hits) - 1;
        hits &= hits - 1;
        const int ray = ray0 + rl;
        int base = __shfl(my_base, rl);
        const float o0 = ray_o[ray * 3], o1 = ray_o[ray * 3 + 1], o2 = ray_o[ray * 3 + 2];
        const float d0 = ray_d[ray * 3], d1 = ray_d[ray * 3 + 1], d2 = ray_d[ray * 3 + 2];
        const float nr = near[ray], range = __fsub_rn(far[ray], nr);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const uint64_t m = ray_mask[(size_t)ray * NCH + ch];
            const int k = ch * 64 + lane;
            if ((m >> lane) & 1ull) {
                const int rank = __popcll(m & ((1ull << lane) - 1ull));
                const int64_t c = (int64_t)base + rank;
                if (c < capacity) {
                    const float t = depth_at(nr, range, k, S);
                    const float x = __fadd_rn(o0, __fmul_rn(t, d0)), y = __fadd_rn(o1, __fmul_rn(t, d1)), z = __fadd_rn(o2, __fmul_rn(t, d2));
                    float xs, ys, zs;
                    to_smpl_frame(x, y, z, Rg, Th, xs, ys, zs);
                    cs_idx[c] = ray * S + k;
                    cs_vid[c] = dense_vid[(size_t)ray * S + k];
                    cs_xs[c] = make_float4(xs, ys, zs, 0.f);
                }
            }
            base += __popcll(m);
        }
    }
}

// The same compaction with SIXTEEN LANES per ray: a wave takes four consecutive rays, a lane the 4 NCH consecutive samples of its ray whose mask bits sit in one word
// (NCH = 1, 2, 4: 64 is a multiple of 4 NCH).  A quarter of the waves of compact_kernel, the same three dependent round trips (offsets + masks, ray, vertex ids), and a lane's
// vertex-id loads are issued together.  Same records at the same positions (SHERF_EXPERIMENT bit 13; round 6).
template <int NCH>
__global__ void __launch_bounds__(256) compact_quad_kernel(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                           const float* __restrict__ near, const float* __restrict__ far,
                                                           int R, int S, const float* __restrict__ Rg, const float* __restrict__ Th,
                                                           const int32_t* __restrict__ ray_base_local,
                                                           const int32_t* __restrict__ chunk_off, const uint64_t* __restrict__ ray_mask,
                                                           const int32_t* __restrict__ dense_vid, int64_t capacity,
                                                           int32_t* __restrict__ ray_base, int32_t* __restrict__ cs_idx,
                                                           int32_t* __restrict__ cs_vid, float4* __restrict__ cs_xs) {
    static_assert(NCH == 1 || NCH == 2 || NCH == 4, "a lane's samples must not straddle a mask word");
    constexpr int SPL = 4 * NCH;                                               // samples per lane
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int ray = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    if (ray >= R) return;
    const int base = ray_base_local[ray] + chunk_off[ray >> 10];
    if (sub == 0) ray_base[ray] = base;
    const int k0 = sub * SPL, word = k0 >> 6, shift = k0 & 63;
    int below = 0;                                                             // set bits of the ray below sample k0
    uint64_t mine = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const uint64_t m = ray_mask[(size_t)ray * NCH + ch];
        below += ch < word ? __popcll(m) : (ch == word ? __popcll(m & ((1ull << shift) - 1ull)) : 0);
        mine = ch == word ? m : mine;
    }
    const unsigned bits = (unsigned)(mine >> shift) & ((1u << SPL) - 1u);
    if (bits == 0) return;
    const float o0 = ray_o[ray * 3], o1 = ray_o[ray * 3 + 1], o2 = ray_o[ray * 3 + 2];
    const float d0 = ray_d[ray * 3], d1 = ray_d[ray * 3 + 1], d2 = ray_d[ray * 3 + 2];
    const float nr = near[ray], range = __fsub_rn(far[ray], nr);
    int vid[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) vid[j] = ((bits >> j) & 1u) ? dense_vid[(size_t)ray * S + k0 + j] : 0;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        if (!((bits >> j) & 1u)) continue;
        const int k = k0 + j;
        const int64_t c = (int64_t)base + below + __popc(bits & ((1u << j) - 1u));
        if (c < capacity) {
            const float t = depth_at(nr, range, k, S);
            const float x = __fadd_rn(o0, __fmul_rn(t, d0)), y = __fadd_rn(o1, __fmul_rn(t, d1)), z = __fadd_rn(o2, __fmul_rn(t, d2));
            float xs, ys, zs;
            to_smpl_frame(x, y, z, Rg, Th, xs, ys, zs);
            cs_idx[c] = ray * S + k;
            cs_vid[c] = vid[j];
            cs_xs[c] = make_float4(xs, ys, zs, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-sample warp: canonical point/direction, exact nearest T-pose vertex, pixel in the observation view
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) warp_geom_kernel(const int32_t* __restrict__ counters, const int32_t* __restrict__ cs_idx,
                                                        const int32_t* __restrict__ cs_vid, const float4* __restrict__ cs_xs,
                                                        const float* __restrict__ ray_d, int S, const float* __restrict__ Rg,
                                                        const float* __restrict__ T2C, const float* __restrict__ C2S,
                                                        const float* __restrict__ t_verts, const float* __restrict__ thdr,
                                                        const int32_t* __restrict__ tcell_start,
                                                        const float4* __restrict__ tcell_pts, int64_t capacity,
                                                        float* __restrict__ geom, int32_t* __restrict__ cs_tvid) {
    const int64_t nv = min((int64_t)counters[0], capacity);
    const CellGrid g = load_grid(thdr);
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nv; c += (int64_t)gridDim.x * 256) {
    const int ray = cs_idx[c] / S;
    const int vid = cs_vid[c];
    const float4 xs = cs_xs[c];
    float vx, vy, vz;
    rot_only(ray_d[ray * 3], ray_d[ray * 3 + 1], ray_d[ray * 3 + 2], Rg, vx, vy, vz);   // renderer.py:310
    // (the 3 x 4 affine as three 16-byte loads: rows of 48 bytes are 16-byte aligned; twelve scalar loads were a third of this kernel's
    //  load instructions, and it is bound by those like the gather -- DESIGN section 5.2)
    const float4* P4 = reinterpret_cast<const float4*>(T2C) + (size_t)vid * 3;
    const float4 pa = P4[0], pb = P4[1], pc = P4[2];
    const float P[12] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w, pc.x, pc.y, pc.z, pc.w};
    float xc = P[0] * xs.x + P[1] * xs.y + P[2] * xs.z + P[9];
    float yc = P[3] * xs.x + P[4] * xs.y + P[5] * xs.z + P[10];
    float zc = P[6] * xs.x + P[7] * xs.y + P[8] * xs.z + P[11];
    float ux = P[0] * vx + P[1] * vy + P[2] * vz;
    float uy = P[3] * vx + P[4] * vy + P[5] * vz;
    float uz = P[6] * vx + P[7] * vy + P[8] * vz;
    // exact nearest T-pose vertex: the same-index vertex bounds the search ball (renderer.py:627)
    float best = dist2_exact(xc, yc, zc, t_verts[vid * 3], t_verts[vid * 3 + 1], t_verts[vid * 3 + 2]);
    int bid = vid;
    float r = sqrtf(best) * 1.00001f + 1e-6f;
    nn_search_batched(g, tcell_start, tcell_pts, xc, yc, zc, r, best, bid);
    const float4* L4 = reinterpret_cast<const float4*>(C2S) + (size_t)bid * 3;
    const float4 la = L4[0], lb = L4[1], lc = L4[2];
    const float L[12] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w, lc.x, lc.y, lc.z, lc.w};
    float hx = L[0] * xc + L[1] * yc + L[2] * zc + L[9];
    float hy = L[3] * xc + L[4] * yc + L[5] * zc + L[10];
    float hz = L[6] * xc + L[7] * yc + L[8] * zc + L[11];
    float iz = hz + 1e-5f;                                 // renderer.py:699
    float4* o = reinterpret_cast<float4*>(geom + c * 8);
    o[0] = make_float4(xc, yc, zc, ux); o[1] = make_float4(uy, uz, hx / iz, hy / iz);
    cs_tvid[c] = bid;
    }
}

}  // namespace

int sherf_sample_mask_nn_impl(const float* ray_o, const float* ray_d, const float* near, const float* far, int R, int S, const float* Rg, const float* Th,
                              const float* grid_hdr, const int32_t* cell_start, const float* cell_pts, const uint32_t* near_mask, int64_t capacity,
                              int32_t* counters, int32_t* ray_base, int32_t* ray_cnt, int32_t* cs_idx, int32_t* cs_vid, float* cs_xs, int32_t* dense_vid,
                              uint64_t* ray_mask, int32_t* scan_ws, const int32_t* near_hdr, const uint16_t* near_list, sherf_stream_t stream, int32_t* sticky);

extern "C" int sherf_build_cells(const float* verts, int n, const float* R, const float* Th, float cell_size,
                                 float* grid_hdr, int32_t* cell_start, float* cell_pts, int32_t* scratch,
                                 uint32_t* near_mask, sherf_stream_t stream) {
    SHERF_CHECK_ARG(verts && grid_hdr && cell_start && cell_pts && scratch);
    SHERF_CHECK_ARG(n > 0 && n <= 65535 && cell_size > 0.f);        // (u16 point indices in the sampler's candidate records)
    SHERF_CHECK_ARG((R == nullptr) == (Th == nullptr));
    hipLaunchKernelGGL(build_cells_kernel, dim3(1), dim3(1024), 0, as_stream(stream), verts, n, R, Th, cell_size,
                       grid_hdr, cell_start, reinterpret_cast<float4*>(cell_pts), scratch, near_mask);
    if (near_mask) {
        (void)hipMemsetAsync(near_mask, 0, 32768 * sizeof(uint32_t), as_stream(stream));
        hipLaunchKernelGGL(near_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 16384 * sizeof(uint32_t), as_stream(stream),
                           reinterpret_cast<const float*>(scratch), n, grid_hdr, cell_size, near_mask);
    }
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_build_cells2(const float* verts_a, const float* R_a, const float* Th_a, const float* verts_b, int n,
                                  float cell_size, float* grid_hdr, int32_t* cell_start, float* cell_pts, int32_t* scratch,
                                  uint32_t* near_mask, sherf_stream_t stream) {
    SHERF_CHECK_ARG(verts_a && R_a && Th_a && verts_b && grid_hdr && cell_start && cell_pts && scratch);
    SHERF_CHECK_ARG(n > 0 && n <= 65535 && cell_size > 0.f);        // (u16 point indices in the sampler's candidate records)
    hipLaunchKernelGGL(build_cells2_kernel, dim3(2), dim3(1024), 0, as_stream(stream), verts_a, R_a, Th_a, verts_b, n, cell_size,
                       grid_hdr, cell_start, reinterpret_cast<float4*>(cell_pts), scratch, near_mask);
    if (near_mask) {                                                 // (NULL: sherf_build_near_lists writes the mask from its counts)
        (void)hipMemsetAsync(near_mask, 0, 32768 * sizeof(uint32_t), as_stream(stream));
        hipLaunchKernelGGL(near_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 16384 * sizeof(uint32_t), as_stream(stream),
                           reinterpret_cast<const float*>(scratch), n, grid_hdr, cell_size, near_mask);
    }
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_build_near_lists(const float* grid_hdr, const float* cell_pts, int n, float radius, int32_t* near_hdr,
                                      uint16_t* near_list, int64_t list_cap, uint32_t* near_mask, sherf_stream_t stream) {
    SHERF_CHECK_ARG(grid_hdr && cell_pts && near_hdr && near_list);
    SHERF_CHECK_ARG(n > 0 && n <= 65535 && radius > 0.f && list_cap >= (int64_t)125 * n + 3 * SHERF_NEAR_SUBCELLS);
    hipStream_t st = as_stream(stream);
    int2* nh = reinterpret_cast<int2*>(near_hdr);
    int32_t* cursor = near_hdr + 2 * SHERF_NEAR_SUBCELLS;                 // the word behind the last header
    (void)hipMemsetAsync(near_hdr, 0, 2 * (size_t)SHERF_NEAR_SUBCELLS * sizeof(int32_t), st);       // (the headers; the two cursor words: near_lists_pairs_kernel<false>)
    const float4* pts = reinterpret_cast<const float4*>(cell_pts);
    hipLaunchKernelGGL(near_lists_pairs_kernel<false>, dim3(cdiv(n * 32, 256)), dim3(256), 0, st, pts, n, grid_hdr, radius, nh, near_list, list_cap, cursor);
    hipLaunchKernelGGL(near_lists_alloc_kernel, dim3(SHERF_NEAR_SUBCELLS / 256), dim3(256), 0, st, grid_hdr, nh, cursor, list_cap, near_mask);
    hipLaunchKernelGGL(near_lists_pairs_kernel<true>, dim3(cdiv(n * 32, 256)), dim3(256), 0, st, pts, n, grid_hdr, radius, nh, near_list, list_cap, static_cast<int32_t*>(nullptr));
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_sample_mask_nn(const float* ray_o, const float* ray_d, const float* near, const float* far,
                                       int R, int S, const float* Rg, const float* Th, const float* grid_hdr,
                                       const int32_t* cell_start, const float* cell_pts, const uint32_t* near_mask, int64_t capacity,
                                    int32_t* counters, int32_t* ray_base, int32_t* ray_cnt, int32_t* cs_idx,
                                       int32_t* cs_vid, float* cs_xs, int32_t* dense_vid, uint64_t* ray_mask,
                                       int32_t* scan_ws, const int32_t* near_hdr, const uint16_t* near_list, sherf_stream_t stream) {
    return sherf_sample_mask_nn_impl(ray_o, ray_d, near, far, R, S, Rg, Th, grid_hdr, cell_start, cell_pts, near_mask, capacity, counters, ray_base, ray_cnt,
                                     cs_idx, cs_vid, cs_xs, dense_vid, ray_mask, scan_ws, near_hdr, near_list, stream, nullptr);
}

// (the frame driver's entry: `sticky`, see init_counters_kernel)
int sherf_sample_mask_nn_impl(const float* ray_o, const float* ray_d, const float* near, const float* far,
                              int R, int S, const float* Rg, const float* Th, const float* grid_hdr,
                              const int32_t* cell_start, const float* cell_pts, const uint32_t* near_mask, int64_t capacity,
                              int32_t* counters, int32_t* ray_base, int32_t* ray_cnt, int32_t* cs_idx,
                              int32_t* cs_vid, float* cs_xs, int32_t* dense_vid, uint64_t* ray_mask,
                              int32_t* scan_ws, const int32_t* near_hdr, const uint16_t* near_list, sherf_stream_t stream, int32_t* sticky) {
    SHERF_CHECK_ARG(ray_o && ray_d && near && far && Rg && Th && grid_hdr && cell_start && cell_pts && near_mask);
    SHERF_CHECK_ARG(counters && ray_base && ray_cnt && cs_idx && cs_vid && cs_xs && dense_vid && ray_mask && scan_ws);
    SHERF_CHECK_ARG(R > 0 && S >= 2 && S <= 256 && (int64_t)R * S < 2147483647LL && capacity > 0 && (near_hdr == nullptr) == (near_list == nullptr));
    hipStream_t st = as_stream(stream);
    const int nch = (S + 63) / 64;
    const int n_chunks = cdiv(R, 1024);
    int32_t* base_local = scan_ws;              // [R]
    int32_t* chunk_sum = scan_ws + R;           // [n_chunks]
    const float4* cp = reinterpret_cast<const float4*>(cell_pts);
    int32_t* cand_count = scan_ws + R + R / 1024 + 1;      // last word of scan_ws ([R] local bases, [<= R/1024 + 1] chunk sums, this)
    hipLaunchKernelGGL(init_counters_kernel, dim3(1), dim3(1), 0, st, counters, cand_count, sticky);
    hipLaunchKernelGGL(depth_minmax_kernel, dim3(min(256, cdiv(R, 256))), dim3(256), 0, st, near, far, R, S, counters);
    // two passes over a dense candidate list (see cand_mark_kernel) whenever the list -- `capacity` records in cs_xs, which the
    // compaction below only writes afterwards -- provably holds every sample; sherf_set_debug bit 9 forces the one-wave-per-ray kernel
    const bool two_pass = !(g_sherf_debug & 512) && capacity >= (int64_t)R * S;
    if (two_pass) {
        // the search is a PERSISTENT grid (the candidate count is only known on the device), eight workgroups per CU.  Capping it at six
        // (22 KiB of LDS padding, as sample_nn_kernel below is) to keep wave slots free for the encoder's launches on the other stream
        // was measured and LOST: 1.346 -> 1.381 ms per frame (profiles/r03_bench_d_*.txt) -- the search simply ran longer.
        float4* cand_list = reinterpret_cast<float4*>(cs_xs);            // one 16-byte record per candidate: the list holds `capacity` of them
        const int64_t list_cap = capacity;
        unsigned long long* rm = reinterpret_cast<unsigned long long*>(ray_mask);
        const bool lists = near_hdr && near_list && !(g_sherf_debug & 16384);      // debug bit 14: the cell walk, for A/B runs
        // persistent workgroups per CU of the list search: SIX of the eight that fit -- the other two wave slots per SIMD go to the
        // encoder's small dependent launches on the other stream, which the frame waits for just as long (MI355X, dense framing: 8 ->
        // 1.751 ms, 6 -> 1.726, 4 -> 1.774, 3 -> 1.826 per frame; profiles/r04_call_m_*).  Debug bits 20-23 override for A/B runs.
        const int search_wgs = ((g_sherf_debug >> 20) & 15) ? ((g_sherf_debug >> 20) & 15) : 6;
        const bool tight = (sherf_experiment() & 32768) != 0;           // bit 15: ... held to 80 registers (six workgroups per CU; three spilled) instead of 87 (five)
        const bool deep = (sherf_experiment() & 16384) != 0;            // SHERF_EXPERIMENT bit 14: the list search one pipeline stage deeper (round 6)
#define SHERF_TWO_PASS(N)                                                                                                          \
        hipLaunchKernelGGL(cand_mark_kernel<N>, dim3(cdiv(R, 4 * (16 / N))), dim3(256), 0, st, ray_o, ray_d, near, far, R, S, Rg, Th,    \
                           grid_hdr, near_mask, cand_list, list_cap, cand_count, ray_mask, g_sherf_debug);                         \
        if (lists && deep && tight)                                                                                                \
            hipLaunchKernelGGL((cand_search_lists2_kernel<N, 6>), dim3(search_wgs * n_cus()), dim3(256), 0, st, cand_list, list_cap, cand_count, S, \
                               grid_hdr, reinterpret_cast<const int2*>(near_hdr), near_list, cp, rm, dense_vid);                   \
        else if (lists && deep)                                                                                                    \
            hipLaunchKernelGGL((cand_search_lists2_kernel<N, 5>), dim3(min(search_wgs, 5) * n_cus()), dim3(256), 0, st, cand_list, list_cap, cand_count, S, \
                               grid_hdr, reinterpret_cast<const int2*>(near_hdr), near_list, cp, rm, dense_vid);                   \
        else if (lists)                                                                                                            \
            hipLaunchKernelGGL(cand_search_lists_kernel<N>, dim3(search_wgs * n_cus()), dim3(256), 0, st, cand_list, list_cap, cand_count, S, \
                               grid_hdr, reinterpret_cast<const int2*>(near_hdr), near_list, cp, rm, dense_vid);                   \
        else                                                                                                                       \
            hipLaunchKernelGGL(cand_search_kernel<N>, dim3(8 * n_cus()), dim3(256), 0, st, cand_list, list_cap, cand_count, S,          \
                               grid_hdr, cell_start, cp, rm, dense_vid);                                                           \
        hipLaunchKernelGGL(scan_chunk_mask_kernel<N>, dim3(n_chunks), dim3(1024), 0, st, ray_mask, R, ray_cnt, base_local, chunk_sum)
        if (nch == 1) { SHERF_TWO_PASS(1); } else if (nch == 2) { SHERF_TWO_PASS(2); } else if (nch == 3) { SHERF_TWO_PASS(3); } else { SHERF_TWO_PASS(4); }
#undef SHERF_TWO_PASS
    } else {
    // 10 KiB of unused dynamic LDS per workgroup: caps the sampler at 6 workgroups per CU (6 of the 8 wave slots per SIMD).  The
    // encoder's small dependent launches on the other stream are the frame's critical path until the gather; with every wave slot
    // taken by the sampler their workgroups queue behind it (measured: frame 1.97 ms uncapped, 1.91 at 6, 1.92 at 5, 1.94 at 4;
    // profiles/r02_sconv_sampler_balance.txt).
    constexpr int nn_pad = 10 * 1024;
#define SHERF_SAMPLE_LAUNCH(N)                                                                                        \
    hipLaunchKernelGGL(sample_nn_kernel<N>, dim3(cdiv(R, 4)), dim3(256), nn_pad, st, ray_o, ray_d, near, far, R, S, Rg, Th, \
                       grid_hdr, cell_start, cp, near_mask, ray_cnt, ray_mask, dense_vid, g_sherf_debug)
    if (nch == 1) SHERF_SAMPLE_LAUNCH(1); else if (nch == 2) SHERF_SAMPLE_LAUNCH(2);
    else if (nch == 3) SHERF_SAMPLE_LAUNCH(3); else SHERF_SAMPLE_LAUNCH(4);
    hipLaunchKernelGGL(scan_chunk_kernel, dim3(n_chunks), dim3(1024), 0, st, ray_cnt, R, base_local, chunk_sum);
    }
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, st, chunk_sum, n_chunks, counters);
#define SHERF_COMPACT_LAUNCH(N)                                                                                        \
    hipLaunchKernelGGL(compact_kernel<N>, dim3(cdiv(R, 4)), dim3(256), 0, st, ray_o, ray_d, near, far, R, S, Rg, Th,   \
                       base_local, chunk_sum, ray_mask, dense_vid, capacity, ray_base, cs_idx, cs_vid,                 \
                       reinterpret_cast<float4*>(cs_xs), cp)
#define SHERF_COMPACT_RAYS_LAUNCH(N)                                                                                   \
    hipLaunchKernelGGL(compact_rays_kernel<N>, dim3(cdiv(R, 256)), dim3(256), 0, st, ray_o, ray_d, near, far, R, S, Rg, Th, \
                       base_local, chunk_sum, ray_mask, dense_vid, capacity, ray_base, cs_idx, cs_vid,                 \
                       reinterpret_cast<float4*>(cs_xs))
#define SHERF_COMPACT_QUAD_LAUNCH(N)                                                                                   \
    hipLaunchKernelGGL(compact_quad_kernel<N>, dim3(cdiv(R, 16)), dim3(256), 0, st, ray_o, ray_d, near, far, R, S, Rg, Th, \
                       base_local, chunk_sum, ray_mask, dense_vid, capacity, ray_base, cs_idx, cs_vid,                 \
                       reinterpret_cast<float4*>(cs_xs))
    if ((sherf_experiment() & 8192) && nch != 3) {      // SHERF_EXPERIMENT bit 13: sixteen lanes per ray (round 6)
        if (nch == 1) SHERF_COMPACT_QUAD_LAUNCH(1); else if (nch == 2) SHERF_COMPACT_QUAD_LAUNCH(2); else SHERF_COMPACT_QUAD_LAUNCH(4);
    } else
    if (sherf_experiment() & 2048) {            // SHERF_EXPERIMENT bit 11: one lane per ray, whole waves for the hit rays only (round 6; MEASURED SLOWER: 96 vs 77 us, frame 1.615 vs 1.597 ms -- the hit rays of a wave are walked one after the other, each behind its own round trips; profiles/r06_call_m_*)
        if (nch == 1) SHERF_COMPACT_RAYS_LAUNCH(1); else if (nch == 2) SHERF_COMPACT_RAYS_LAUNCH(2);
        else if (nch == 3) SHERF_COMPACT_RAYS_LAUNCH(3); else SHERF_COMPACT_RAYS_LAUNCH(4);
    } else {
    if (nch == 1) SHERF_COMPACT_LAUNCH(1); else if (nch == 2) SHERF_COMPACT_LAUNCH(2);
    else if (nch == 3) SHERF_COMPACT_LAUNCH(3); else SHERF_COMPACT_LAUNCH(4);
    }
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_warp_geom(const int32_t* counters, const int32_t* cs_idx, const int32_t* cs_vid,
                               const float* cs_xs, const float* ray_d, int S, const float* Rg, const float* T2C,
                               const float* C2S, const float* t_verts, const float* tgrid_hdr,
                               const int32_t* tcell_start, const float* tcell_pts, int64_t capacity, float* geom,
                               int32_t* cs_tvid, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && cs_idx && cs_vid && cs_xs && ray_d && Rg && T2C && C2S && t_verts && tgrid_hdr &&
                    tcell_start && tcell_pts && geom && cs_tvid);
    SHERF_CHECK_ARG(S >= 2 && capacity > 0);
    // (SHERF_EXPERIMENT bits 16-19, round 6: w persistent workgroups per CU instead of one per 256 samples -- room for the encoder's big-register waves beside it)
    const int warp_wgs = (sherf_experiment() >> 16) & 15;
    const int warp_grid = warp_wgs ? min(warp_wgs * n_cus(), cdiv(capacity, 256)) : min(8192, cdiv(capacity, 256));
    hipLaunchKernelGGL(warp_geom_kernel, dim3(warp_grid), dim3(256), 0, as_stream(stream), counters, cs_idx,
                       cs_vid, reinterpret_cast<const float4*>(cs_xs), ray_d, S, Rg, T2C, C2S, t_verts, tgrid_hdr,
                       tcell_start, reinterpret_cast<const float4*>(tcell_pts), capacity, geom, cs_tvid);
    SHERF_LAUNCH_CHECK();
}
