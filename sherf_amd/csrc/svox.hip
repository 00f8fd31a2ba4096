// Sparse voxel encoder (gfx950): row a11 of SURVEY.md section 8 -- SparseConvNet (renderer.py:708-871) over the
// <= 6890 voxels that hold an SMPL vertex, without spconv and without ever materialising `.dense()` volumes.
//
// A level is {bitmap (1 bit / voxel), popcount prefix per 32-bit word, keys[row]}: row id == rank of the voxel's
// bit, so rows are sorted by linear voxel index, neighbour lookup is one bit test + popcount, and the level
// order is deterministic (no hash insertion order).  Semantics follow the published spconv-2.x indice-pair
// algorithm as restated in oracle/sherf_oracle.py (rows sharing a voxel sum at that voxel; `mult` carries the
// multiplicity so BatchNorm statistics are taken over the reference's ROW set).
#include "common.h"

#include <mutex>

#include <functional>

namespace {

__global__ void __launch_bounds__(256) mark_rows_kernel(const int32_t* __restrict__ coord, int n, int D, int H, int W,
                                                        uint32_t* __restrict__ bitmap) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int z = coord[i * 4 + 1], y = coord[i * 4 + 2], x = coord[i * 4 + 3];
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return;     // outside the T-pose box: dropped
    int key = (z * H + y) * W + x;
    atomicOr(&bitmap[key >> 5], 1u << (key & 31));
}

// SparseConv3d(k=3,s=2,p=1) output sites: o = (q + 1 - k)/2 for every tap k with q+1-k even and o in range
__global__ void __launch_bounds__(256) mark_down_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ n_rows,
                                                        int D, int H, int W, uint32_t* __restrict__ bitmap_out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *n_rows) return;
    const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int key = keys[i];
    int z = key / (H * W), y = (key / W) % H, x = key % W;
    for (int kz = 0; kz < 3; ++kz) {
        int nz = z + 1 - kz;
        if (nz < 0 || (nz & 1) || (nz >> 1) >= Do) continue;
        for (int ky = 0; ky < 3; ++ky) {
            int ny = y + 1 - ky;
            if (ny < 0 || (ny & 1) || (ny >> 1) >= Ho) continue;
            for (int kx = 0; kx < 3; ++kx) {
                int nx = x + 1 - kx;
                if (nx < 0 || (nx & 1) || (nx >> 1) >= Wo) continue;
                int ok = ((nz >> 1) * Ho + (ny >> 1)) * Wo + (nx >> 1);
                // an output voxel is hit by up to 27 rows and a word by hundreds: same-address atomics serialise, so look
                // first (a stale 0 only costs a redundant, idempotent atomicOr)
                const uint32_t bit = 1u << (ok & 31);
                if (!(__hip_atomic_load(&bitmap_out[ok >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit))
                    atomicOr(&bitmap_out[ok >> 5], bit);
            }
        }
    }
}

// exclusive popcount scan, 3 launches: per-1024-word chunk scan, scan of chunk sums, add chunk offsets
__global__ void __launch_bounds__(256) scan_local_kernel(const uint32_t* __restrict__ bitmap, int n_words,
                                                        int32_t* __restrict__ prefix, int32_t* __restrict__ chunk_sum) {
    __shared__ int s[256];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * 1024 + tid * 4;
    int c[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[i] = (w0 + i < n_words) ? __popc(bitmap[w0 + i]) : 0; sum += c[i]; }
    s[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int v = tid >= off ? s[tid - off] : 0;
        __syncthreads();
        s[tid] += v;
        __syncthreads();
    }
    int run = s[tid] - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (w0 + i < n_words) prefix[w0 + i] = run; run += c[i]; }
    if (tid == 255) chunk_sum[blockIdx.x] = s[255];
}

// one SMALL workgroup: a 1024-thread block needs 16 free wave slots on one CU at once and starved for up to 240 us next to
// the chip-filling sampler (rocprofv3 timeline); 256 threads slot in anywhere
__global__ void __launch_bounds__(256) scan_chunks_kernel(int32_t* __restrict__ chunk_sum, int n_chunks, int32_t* __restrict__ n_rows) {
    __shared__ int s[256];
    int carry = 0;
    for (int b0 = 0; b0 < n_chunks; b0 += 256) {
        int i = b0 + threadIdx.x;
        int v = i < n_chunks ? chunk_sum[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int a = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_chunks) chunk_sum[i] = carry + s[threadIdx.x] - v;
        int tot = s[255];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) *n_rows = carry;
}

__global__ void __launch_bounds__(256) scan_add_kernel(const uint32_t* __restrict__ bitmap, int32_t* __restrict__ prefix, int n_words,
                                                       const int32_t* __restrict__ chunk_off, uint2* __restrict__ wp,
                                                       int32_t* __restrict__ keys) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w < n_words) {
        const int pf = prefix[w] + chunk_off[w >> 10];
        uint32_t bits = bitmap[w];
        prefix[w] = pf;
        wp[w] = make_uint2(bits, (uint32_t)pf);               // one 8-byte record per word: a lookup is a single load
        if (keys) {                                           // row id -> linear voxel index (fused keys_kernel)
            int r = pf;
            while (bits) {
                keys[r++] = w * 32 + __ffs(bits) - 1;
                bits &= bits - 1;
            }
        }
    }
}

__global__ void __launch_bounds__(256) keys_kernel(const uint32_t* __restrict__ bitmap, const int32_t* __restrict__ prefix,
                                                   int n_words, int32_t* __restrict__ keys) {
    int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    uint32_t bits = bitmap[w];
    int r = prefix[w];
    while (bits) {
        int b = __ffs(bits) - 1;
        keys[r++] = w * 32 + b;
        bits &= bits - 1;
    }
}

// Rows that share a voxel SUM there (spconv indice pairs).  Floating-point atomics would make the sum depend on the
// arrival order when three or more rows collide, so the sum is taken in 2^-30 fixed point (integer adds commute:
// bitwise reproducible), then converted once.  |feature| must stay below 2^33.
constexpr double kFix = 1073741824.0;   // 2^30
__global__ void __launch_bounds__(256) scatter_rows_kernel(const int32_t* __restrict__ coord, const float* __restrict__ feat,
                                                           int n, int C, int D, int H, int W, const uint32_t* __restrict__ bitmap,
                                                           const int32_t* __restrict__ prefix, unsigned long long* __restrict__ acc,
                                                           int32_t* __restrict__ mult) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    int i = idx / C, c = idx % C;
    if (i >= n) return;
    int z = coord[i * 4 + 1], y = coord[i * 4 + 2], x = coord[i * 4 + 3];
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return;
    int key = (z * H + y) * W + x;
    uint32_t word = bitmap[key >> 5], bit = 1u << (key & 31);
    int row = prefix[key >> 5] + __popc(word & (bit - 1u));
    long long q = __double2ll_rn((double)feat[(size_t)i * C + c] * kFix);
    atomicAdd(&acc[(size_t)row * C + c], (unsigned long long)q);
    if (c == 0) atomicAdd(&mult[row], 1);
}

__global__ void __launch_bounds__(256) fix_to_float_kernel(const long long* __restrict__ acc, const int32_t* __restrict__ n_rows, int C,
                                                           float* __restrict__ g) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < *n_rows * C) g[idx] = (float)((double)acc[idx] * (1.0 / kFix));
}

// ---------------------------------------------------------------------------------------------
// Sparse 3x3x3 convolution on MFMA (v_mfma_f32_32x32x16_f16, operands split hi + lo in fp16 -> fp32-grade "f16x3": 22 significant
// bits per operand, three products; round 1 split in bf16 (16 bits) and its ~3e-5 error on the voxel rows was what the tokens --
// and through the decoder's gain the per-sample sigma -- inherited.  Inputs are BatchNorm+ReLU outputs / O(1) features: far inside fp16's range).
//   out[row][co] = sum_tap sum_ci act(in[nbr(row,tap)][ci]) * W[tap][ci][co]
// A workgroup (4 waves) owns 32 output rows; the taps present in the tile are dealt round-robin to the waves, each
// wave accumulates its taps into 32 x Cout fp32 tiles, and the four partial tiles are summed in a fixed order
// through LDS (deterministic).  The A operand (32 rows x 16 input channels) is gathered straight from global memory
// -- lane (row, half) reads 8 consecutive channels of its neighbour row -- with the producer's BatchNorm+ReLU applied
// on the fly; the B operand comes pre-packed in fragment order (sherf_amd/voxel.py: pack_conv_weights).
// mode 0 submanifold, 1 stride-2 (k3 p1), 2 pointwise (1 tap, row -> same row: folds the 1x1 projections).
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pk2(float a, float b) {
    f16x2_t v; v[0] = (_Float16)a; v[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float rt(float a) { return (float)((_Float16)a); }

// BatchNorm of a conv's INPUT.  Training: the producer conv left fixed-point (2^-24) sums of its output and of its
// squares in `acc` (integer atomics: order independent, bitwise reproducible; 8 interleaved sub-accumulators spread the
// same-address traffic); every consumer workgroup turns them into scale/shift in its prologue -- no finalize launch and
// no cross-workgroup hand-shake inside a kernel -- and workgroup 0 also publishes stats / bnparam.  Eval: bnparam holds
// the constants derived from the running statistics.  bnparam == nullptr: raw input (first layer).
// |sum of squares| must stay below 2^39 (activations of rms < ~5000 over 20 K rows).
constexpr double kAccFix = 16777216.0;    // 2^24
constexpr int kAccSub = 8;
struct BnIn {
    const long long* acc;       // [kAccSub][2][Cin] or nullptr
    const int32_t* n_total;     // rows of the reference's row set
    const float* gamma;
    const float* beta;
    float* stats;               // [2][Cin] batch mean / biased variance (written by workgroup 0 when acc != nullptr)
    float* bnparam;             // [3][Cin] scale, shift, relu(shift)
};

// SHERF_SCONV_TRACE (profiling builds only, tools/sconv_trace.py): wave 0 of every workgroup stamps s_memtime at its phase boundaries
// into a global record [launch id, block, rows, NCOT << 8 | NKB, 12 stamps].
#ifndef SHERF_SCONV_TRACE
#define SHERF_SCONV_TRACE 0
#endif
#if SHERF_SCONV_TRACE
__device__ uint32_t* g_sconv_trace = nullptr;
__device__ unsigned g_sconv_trace_cap = 0, g_sconv_trace_n = 0;
#define SCONV_STAMP(k) do { if (threadIdx.x == 0) stamps[k] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SCONV_STAMP(k) do { } while (0)
#endif

// FOLD (mode 2, one tap: the waves split the output-channel tiles) and IN_BN (BatchNorm of the input) are compile-time: as run-time
// switches they put a branch around every weight load and every MFMA group of the tap loop, and the compiler scheduled nothing
// across them (round 2's ISA: one global load per basic block).
// SP: single fp16 product per term (operands rounded to nearest even, the `hi` weight fragments only) instead of the three of the
// f16x3 split: a third of the MFMA issue and half the weight loads of a tap.  Follows the frame's table precision (the encoder's output
// is rounded to fp16 rows there anyway, and mlp_precision='auto' calibrates the whole configuration against f16x3 on the frame).
// NW: waves per workgroup the 27 taps are split over.  A workgroup's life is a serial chain of dependent gathers, ~7 taps per wave at
// NW = 4 (profiles/r02_sconv_trace_v3.txt: 28 K of its ~50 K cycles); at NW = 8 a wave walks 3-4 taps and the partial tiles of eight
// waves are summed in LDS (up to 98 KiB: these workgroups are alone or nearly alone on their CU anyway).
// Round 6: in the three-product instances (SP = false: the fp32-grade "f16x3" class -- the reference configuration of
// `auto`, every training forward, and the input-gradient convolutions of the backward) the `lo` halves of BOTH operands are carried at 2^11 times
// their value and their two products go to a SECOND accumulator that joins the first with a factor 2^-11 at the end.  Why: lo = x - fp16(x) is at
// most 2^-11 |x|, and fp16 resolves nothing below 2^-24 -- for an operand at 1e-2 of the tile's largest (weights of 0.01, the typical entries of
// a gradient row) the unscaled lo kept 6 bits instead of 11: 2e-6 of the specification per layer where fp32 has 6e-8, and every BatchNorm
// backward behind such a layer amplifies rounding 10^3-10^4 times.  Scaled, lo keeps its 11 bits down to 2^-13 of fp16's
// range: 22 bits per operand, the same three MFMAs, 16 more registers per output tile.
// CS (round 6): COLUMN SPLIT -- a workgroup computes ONE of the layer's NCOT output tiles (blockIdx.y) instead of all of them.  The 64 / 96 -> 96 instances hold three
// accumulator tiles and three tiles' weight fragments per K-block: 208-248 VGPRs a wave, and such a wave needs that many free registers on all four SIMDs of a CU at
// once -- beside the ray side's kernels (51-59 registers, up to eight waves per SIMD) it waits: those two layers run 18 us alone and 90-150 us in the frame
// (profiles/r06_kernel_trace_evidence_run_1_*).  One tile per workgroup: a third of the registers, three times the workgroups, the gathered rows read three times
// (2.7 K / 10 K rows: nothing).  Same products in the same order per output element: bit-identical rows and statistics.
template <int NCOT, int NKB, bool FOLD, int IN_BN, bool SP, int NW, bool CS = false>   // Cout = 32 * NCOT, Cin = 16 * NKB; IN_BN 0 raw input, 1 BatchNorm, 2 BatchNorm + row multiplicity
__global__ void __launch_bounds__(64 * NW, (NW == 8 ? 2 : ((NCOT * NKB >= 12 && !CS) ? 1 : 2)))   // (level 2 launches 318 workgroups: two per CU must fit)
sconv3_kernel(const int32_t* __restrict__ keys_out, const int32_t* __restrict__ n_rows_out,
                                                     int Do, int Ho, int Wo, const uint2* __restrict__ wp_in, int Di, int Hi, int Wi,
                                                     const float* __restrict__ in_raw, BnIn bin,
                                                     const int32_t* __restrict__ in_mult, const uint4* __restrict__ wpk, int mode,
                                                     float* __restrict__ out_raw, long long* __restrict__ out_acc, int trace_id,
                                                     const uint32_t* __restrict__ in_amax) {
    constexpr int COUT = 32 * NCOT, Cin = 16 * NKB;
    static_assert(!CS || !FOLD, "column split: the 27-tap instances");
    constexpr int NC = CS ? 1 : NCOT, COUTL = 32 * NC;      // output tiles / columns THIS workgroup computes
    const int c0 = CS ? (int)blockIdx.y : 0;                 // its first tile
#if SHERF_SCONV_TRACE
    uint32_t stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    SCONV_STAMP(0);
#endif
    // the encoder is a short serial chain of small launches running next to the ray side's chip-filling kernels: its
    // waves take issue priority over co-resident waves (measured -20..30 us per frame; sherf_set_debug bit 7 turns it off)
    if (!(mode & 256)) __builtin_amdgcn_s_setprio(3);
    const bool out_half = FOLD && (mode & 512);          // folded rows as fp16 (the gather's half-table mode, csrc/fold.hip)
    mode &= 255;
    // in_amax (the input-gradient use, sherf_svox_conv3_dgrad): bits of max |in_raw|.  The fp16 operand split holds 22 bits of an O(1) value
    // but gradients are O(1e-7): the rows are scaled by the power of two that brings their maximum into [0.5, 1) before the split (exact)
    // and the result is scaled back (exact); elements below 6e-5 of the maximum keep an absolute error of 3e-8 of it -- fp32's own.
    float in_scale = 1.f, out_scale = 1.f;
    if (IN_BN == 0 && in_amax) {
        const int be = (int)((*in_amax >> 23) & 0xffu);                 // biased exponent of the maximum
        if (be >= 1 && be <= 252) { in_scale = __uint_as_float((uint32_t)(253 - be) << 23); out_scale = __uint_as_float((uint32_t)(be + 1) << 23); }
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_nb = reinterpret_cast<int*>(smem);                                   // [27][32]
    float* s_bn = reinterpret_cast<float*>(smem + 27 * 32 * 4);                 // [3][Cin]
    float* s_red = s_bn + 3 * Cin;                                              // [NW][32][COUT]
    constexpr int NT = 64 * NW;
    const int n_rows = *n_rows_out;
    const int row0 = blockIdx.x * 32;
    if (row0 >= n_rows) return;
    // `wave` must be PROVABLY wave-uniform: the FOLD instances choose their output tile by it (c == csel), and with a merely divergent-looking
    // condition hipcc predicates that block by EXEC and -- when the block is one MFMA, as in the single-product instances -- drops the
    // s_cbranch_execz around it (short-skip removal).  MFMAs ignore EXEC on gfx950: the waves that do NOT own the tile then multiply with
    // whatever sits in the weight registers.  That was round 3's "single-product convolutions: right on the host build, wrong on the
    // MI355X" (tools/sconv_fold_diag.py: tile 2 of every fold garbage; the three-product instances kept their branches only because their
    // blocks are longer).  readfirstlane makes the condition scalar: real branches.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntaps = FOLD ? 1 : 27;
    {   // neighbour table: thread -> row tid & 31, taps (tid >> 5) + 8 j.  The row's key is read once and the (up to) four bitmap
        // records are fetched together: two dependent L2 trips per workgroup instead of seven (profiles/r02_sconv_trace_v1.txt:
        // this table was 6-8 K of a workgroup's 17-70 K cycles)
        const int rr = tid & 31, row = row0 + rr;
        const bool live = row < n_rows;
        const int key = (live && !FOLD) ? keys_out[row] : 0;
        const int z = key / (Ho * Wo), y = (key / Wo) % Ho, x = key % Wo;
        constexpr int TG = 2 * NW, NJ = 32 / TG;           // tap groups of 32 threads, taps per thread
        int qk[NJ];
        uint2 rec[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int tap = (tid >> 5) + TG * j;
            const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
            // mode 0: same level, offset k - 1; mode 1: fine voxel 2 o + k - 1 under coarse output o; mode 3 (transpose of mode 1, the input
            // gradient of a stride-2 layer): coarse voxel (q + k - 1) / 2 above fine output q where that is a whole, non-negative voxel
            int qz = mode == 1 ? 2 * z + kz - 1 : z + kz - 1, qy = mode == 1 ? 2 * y + ky - 1 : y + ky - 1, qx = mode == 1 ? 2 * x + kx - 1 : x + kx - 1;
            bool ok = live && !FOLD && tap < 27 && qz >= 0 && qy >= 0 && qx >= 0;
            if (mode == 3) { ok = ok && !((qz | qy | qx) & 1); qz >>= 1; qy >>= 1; qx >>= 1; }
            ok = ok && qz < Di && qy < Hi && qx < Wi;
            qk[j] = ok ? (qz * Hi + qy) * Wi + qx : -1;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) rec[j] = qk[j] >= 0 ? wp_in[qk[j] >> 5] : make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int tap = (tid >> 5) + TG * j;
            if (tap >= ntaps) continue;
            int nb = -1;
            if (FOLD) nb = live ? row : -1;
            else if (qk[j] >= 0) {
                const uint32_t bit = 1u << (qk[j] & 31);
                if (rec[j].x & bit) nb = (int)rec[j].y + __popc(rec[j].x & (bit - 1u));
            }
            s_nb[tap * 32 + rr] = nb;
        }
    }
    SCONV_STAMP(1);                                  // neighbour table done (this thread's share)
    constexpr bool in_bn = IN_BN != 0, has_mult = IN_BN == 2, SL = !SP;       // SL: scaled lo halves, second accumulator (see the kernel's header)
    if (in_bn) {
        if (bin.acc) {
            if (tid < Cin) {
                long long a1 = 0, a2 = 0;
#pragma unroll
                for (int q = 0; q < kAccSub; ++q) { a1 += bin.acc[(q * 2 + 0) * Cin + tid]; a2 += bin.acc[(q * 2 + 1) * Cin + tid]; }
                const double n = (double)(*bin.n_total);
                const double mean = (double)a1 * (1.0 / kAccFix) / n, var = fmax((double)a2 * (1.0 / kAccFix) / n - mean * mean, 0.0);
                const float meanf = (float)mean, varf = (float)var;
                const float scale = bin.gamma[tid] / sqrtf(varf + 1e-3f);
                const float shift = bin.beta[tid] - meanf * scale;
                s_bn[tid] = scale; s_bn[Cin + tid] = shift; s_bn[2 * Cin + tid] = fmaxf(shift, 0.f);
                if (blockIdx.x == 0) {
                    bin.stats[tid] = meanf; bin.stats[Cin + tid] = varf;
                    bin.bnparam[tid] = scale; bin.bnparam[Cin + tid] = shift; bin.bnparam[2 * Cin + tid] = fmaxf(shift, 0.f);
                }
            }
        } else {
            for (int i = tid; i < 3 * Cin; i += NT) s_bn[i] = bin.bnparam[i];
        }
    }
    SCONV_STAMP(2);                                  // BatchNorm prologue done
    __syncthreads();
    SCONV_STAMP(3);

    f32x16_t acc[NC];
    f32x16_t acc2[SL ? NC : 1];                        // SL: the two lo products, at 2^11 times their value
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[c][q] = 0.f; if (SL) acc2[c][q] = 0.f; }
    const int r = lane & 31, h = lane >> 5;
    // taps of this wave (tap = wave, wave+4, ...) that have at least one neighbour in the tile
    uint32_t tapmask = 0;
    for (int tap = wave; tap < ntaps; tap += NW)
        if (__ballot(s_nb[tap * 32 + r] >= 0) != 0ull) tapmask |= 1u << tap;
    // pointwise fold (one tap): the waves split the output-channel tiles instead of the taps
    const int csel = FOLD ? wave : -1;
    if (FOLD) tapmask = wave < NCOT ? 1u : 0u;
    tapmask = __builtin_amdgcn_readfirstlane(tapmask);

    // The tile is a chain of dependent gathers (neighbour row, weight fragments -> MFMA) and a workgroup is alone or nearly alone on
    // its CU (85-600 workgroups per launch), so nothing hides a load that is not issued well ahead.  Round 2's in-kernel timeline
    // (profiles/r02_sconv_trace_v1.txt) showed one exposed L2 round trip PER K-BLOCK (weights fetched one K-block ahead: 3-8 K cycles
    // per tap, 60-75 % of a workgroup's life).  Now every buffer is refilled IN PLACE right after its last use with the data of the
    // tap PF taps ahead -- rows and weight fragments, K-block by K-block -- so every load has PF whole taps of work in front of it.
    // PF is chosen by the register cost of a tap (rows 2 NKB + weights 2 NKB NCOT uint4): 3 / 2 / 1 taps.  Lanes without that
    // neighbour read row 0 and are zeroed after the BatchNorm transform (no divergent control flow in the pipeline).  The order of
    // the accumulation is unchanged (taps ascending per wave, K-blocks ascending): results are bit-identical to the simple loop.
    constexpr int PF = NKB * NC <= 2 ? 3 : (NKB * NC <= 4 ? 2 : 1);   // (32 -> 32 at PF 4 costs 198 registers: 2 workgroups per CU, and level 1 launches 602)
    struct Row { float4 v[2 * NKB]; int nb; float mlt; };
    constexpr int WPC = SP ? 1 : 2;                   // weight fragments per (K-block, output tile): hi [, lo]
    struct Wts { uint4 w[NKB][WPC * NC]; };
    auto row_src = [&](int nb) { return reinterpret_cast<const float4*>(in_raw + (size_t)(nb >= 0 ? nb : 0) * Cin) + 2 * h; };
    auto row_mult = [&](int nb) {                   // (the load is unconditional per lane: one uniform branch, no divergent one)
        const float m = has_mult ? (float)(in_mult[nb >= 0 ? nb : 0] - 1) : 0.f;
        return nb >= 0 ? m : 0.f;
    };
    auto load_w = [&](int tap, int kb, uint4 (&w)[WPC * NC]) {
        const uint4* wsrc = wpk + ((size_t)(tap * NKB + kb) * NCOT + c0) * 2 * 64 + lane;     // memory: [cot][hi, lo][64 lanes]
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < WPC; ++q)                                                  // (SP: the `lo` fragments are never fetched)
                if (!FOLD || c == csel) w[WPC * c + q] = wsrc[(2 * c + q) * 64];
    };
    auto load_all = [&](int tap, Row& R, Wts& W) {
        const int nb = s_nb[tap * 32 + r];
        R.nb = nb; R.mlt = row_mult(nb);
        const float4* src = row_src(nb);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { R.v[2 * kb] = src[4 * kb]; R.v[2 * kb + 1] = src[4 * kb + 1]; load_w(tap, kb, W.w[kb]); }
    };
    // one tap out of (R, W); each K-block's registers are refilled with tap `nxt`'s data (nxt < 0: nothing left to fetch) as soon as used
    auto compute = [&](Row& R, Wts& W, int nxt) {
        const int nb_tab = s_nb[max(nxt, 0) * 32 + r];
        const int nb_next = nxt >= 0 ? nb_tab : -1;
        // (the multiplicity is fetched here and converted after the K-blocks: consumed at once it would drain every load in flight)
        const int mult_raw = has_mult ? in_mult[nb_next >= 0 ? nb_next : 0] : 1;
        const float4* src_next = row_src(nb_next);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const float4 a = R.v[2 * kb], b = R.v[2 * kb + 1];
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (in_bn) {
                const float* sc = s_bn + kb * 16 + 8 * h;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * sc[e] + sc[Cin + e], 0.f) + R.mlt * sc[2 * Cin + e];
            }
            if constexpr (IN_BN == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= in_scale;
            }
            if (R.nb < 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            const uint4 ahi = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
            if constexpr (SP) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (FOLD && c != csel) continue;
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ahi), __builtin_bit_cast(f16x8_t, W.w[kb][WPC * c]), acc[c], 0, 0, 0);
                }
            } else {
            constexpr float LS = 2048.0f;                      // (x - rt(x)) * 2^11 is exact in fp32
            const uint4 alo = make_uint4(pk2((v[0] - rt(v[0])) * LS, (v[1] - rt(v[1])) * LS), pk2((v[2] - rt(v[2])) * LS, (v[3] - rt(v[3])) * LS),
                                         pk2((v[4] - rt(v[4])) * LS, (v[5] - rt(v[5])) * LS), pk2((v[6] - rt(v[6])) * LS, (v[7] - rt(v[7])) * LS));
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (FOLD && c != csel) continue;
                const uint4 bhi = W.w[kb][WPC * c], blo = W.w[kb][WPC * c + WPC - 1];
                // blo = fp16((W - hi) * 2^11): sherf_amd/voxel.py pack_conv_weights
                acc2[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, alo), __builtin_bit_cast(f16x8_t, bhi), acc2[c], 0, 0, 0);
                acc2[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ahi), __builtin_bit_cast(f16x8_t, blo), acc2[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ahi), __builtin_bit_cast(f16x8_t, bhi), acc[c], 0, 0, 0);
            }
            }
            if constexpr (PF == 1) {
                // one buffer: the refill goes out HERE, right behind the K-block that freed its registers (left alone the compiler
                // sinks all of a tap's refills behind its last MFMA and drains them at the top of the next tap); unconditional --
                // past the last tap it fetches tap 0 / row 0 again, a few cached loads instead of a branch per K-block
                R.v[2 * kb] = src_next[4 * kb]; R.v[2 * kb + 1] = src_next[4 * kb + 1];
                load_w(max(nxt, 0), kb, W.w[kb]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (nxt >= 0) {                   // PF buffers: two or three taps of slack, placement does not matter (uniform branch)
                R.v[2 * kb] = src_next[4 * kb]; R.v[2 * kb + 1] = src_next[4 * kb + 1];
                load_w(nxt, kb, W.w[kb]);
            }
        }
        R.nb = nb_next; R.mlt = nb_next >= 0 ? (float)(mult_raw - 1) : 0.f;
    };
    auto next_tap = [&](int tap) -> int {           // next set bit above `tap`, -1 if none (uniform)
        const uint32_t rest = tap >= 31 ? 0u : (tapmask & ~((2u << tap) - 1u));
        return rest ? __ffs(rest) - 1 : -1;
    };
    Row rows[PF];
    Wts wts[PF];
    int taps[PF];
    int t = tapmask ? __ffs(tapmask) - 1 : -1;      // the next tap nobody has fetched yet
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        taps[p] = t;
        if (t >= 0) { load_all(t, rows[p], wts[p]); t = next_tap(t); }
    }
    SCONV_STAMP(4);                                  // tap mask + the first PF taps issued
#if SHERF_SCONV_TRACE
    int n_done = 0;
#endif
    for (bool more = taps[0] >= 0; more;) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (taps[p] < 0) { more = false; break; }
            compute(rows[p], wts[p], t);
#if SHERF_SCONV_TRACE
            if (n_done == 0) SCONV_STAMP(5);         // first tap of wave 0 done
            ++n_done;
#endif
            taps[p] = t;
            if (t >= 0) t = next_tap(t);
        }
    }
    SCONV_STAMP(6);                                  // all taps of wave 0 done
    if constexpr (SL) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[c][q] = __builtin_fmaf(acc2[c][q], 1.0f / 2048.0f, acc[c][q]);
    }
    // D layout: lane = (col j = lane&31, half h), reg q <-> tile row (q&3) + 8*(q>>2) + 4*h
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) s_red[((wave * 32) + (q & 3) + 8 * (q >> 2) + 4 * h) * COUTL + c * 32 + r] = acc[c][q];
    SCONV_STAMP(7);
    __syncthreads();
    SCONV_STAMP(8);                                  // every wave's taps done
    constexpr int G = NT / COUT;                        // row groups: 8 / 4 / 2 (NW = 4), 16 / 8 / 5 (NW = 8).  (CS: the SAME grouping over this workgroup's 32 columns,
    const int g = tid / COUTL, co = tid % COUTL;        //  so that the statistics' partial sums -- and their bits -- are those of the whole-row instance)
    float s1 = 0.f, s2 = 0.f;
    if (g < G)
        for (int rr = g; rr < 32; rr += G) {
            float val = ((s_red[(0 * 32 + rr) * COUTL + co] + s_red[(1 * 32 + rr) * COUTL + co]) + s_red[(2 * 32 + rr) * COUTL + co]) +
                        s_red[(3 * 32 + rr) * COUTL + co];
            if constexpr (NW == 8)
                val += ((s_red[(4 * 32 + rr) * COUTL + co] + s_red[(5 * 32 + rr) * COUTL + co]) + s_red[(6 * 32 + rr) * COUTL + co]) +
                       s_red[(7 * 32 + rr) * COUTL + co];
            val *= out_scale;
            if (row0 + rr < n_rows) {
                if (out_half) reinterpret_cast<_Float16*>(out_raw)[(size_t)(row0 + rr) * COUT + c0 * 32 + co] = (_Float16)val;
                else out_raw[(size_t)(row0 + rr) * COUT + c0 * 32 + co] = val;
                s1 += val; s2 += val * val;
            }
        }
    if (out_acc) {
        __syncthreads();
        if (g < G) { s_red[g * COUTL + co] = s1; s_red[(G + g) * COUTL + co] = s2; }
        __syncthreads();
        if (tid < 2 * COUTL) {
            const int which = tid / COUTL, c2 = tid % COUTL;
            double t = 0.0;
            for (int q = 0; q < G; ++q) t += (double)s_red[(which * G + q) * COUTL + c2];
            atomicAdd(reinterpret_cast<unsigned long long*>(out_acc) + ((blockIdx.x % kAccSub) * 2 + which) * COUT + c0 * 32 + c2,
                      (unsigned long long)__double2ll_rn(t * kAccFix));
        }
    }
#if SHERF_SCONV_TRACE
    SCONV_STAMP(9);
    if (threadIdx.x == 0 && g_sconv_trace) {
        const unsigned slot = atomicAdd(&g_sconv_trace_n, 1u);
        if (slot < g_sconv_trace_cap) {
            uint32_t* d = g_sconv_trace + (size_t)slot * 16;
            d[0] = (uint32_t)trace_id; d[1] = blockIdx.x; d[2] = (uint32_t)n_rows; d[3] = (NCOT << 8) | NKB | ((unsigned)mode << 16) | ((unsigned)n_done << 24);
            for (int k = 0; k < 12; ++k) d[4 + k] = stamps[k];
        }
    }
#endif
}

#if SHERF_SCONV_TRACE
extern "C" int sherf_sconv_set_trace(void* buf, unsigned cap) {
    const unsigned zero = 0;
    SHERF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sconv_trace), &buf, sizeof(buf)));
    SHERF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sconv_trace_cap), &cap, sizeof(cap)));
    SHERF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sconv_trace_n), &zero, sizeof(zero)));
    return SHERF_OK;
}
#endif

// standalone form of the consumer prologue: acc (training) or running stats (eval) -> stats[2][C], bnparam[3][C]
__global__ void __launch_bounds__(128) bn_finalize_kernel(const long long* __restrict__ acc, const int32_t* __restrict__ n_total_p, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ stats, int training, float* __restrict__ bnparam) {
    const int tid = threadIdx.x;
    if (tid >= C) return;
    if (training) {
        long long a1 = 0, a2 = 0;
        for (int q = 0; q < kAccSub; ++q) { a1 += acc[(q * 2 + 0) * C + tid]; a2 += acc[(q * 2 + 1) * C + tid]; }
        const double n = (double)(*n_total_p);
        const double mean = (double)a1 * (1.0 / kAccFix) / n, var = fmax((double)a2 * (1.0 / kAccFix) / n - mean * mean, 0.0);
        stats[tid] = (float)mean; stats[C + tid] = (float)var;
    }
    const float mean = stats[tid], var = stats[C + tid];
    const float scale = gamma[tid] / sqrtf(var + 1e-3f);
    const float shift = beta[tid] - mean * scale;
    bnparam[tid] = scale; bnparam[C + tid] = shift; bnparam[2 * C + tid] = fmaxf(shift, 0.f);
}

}  // namespace

extern "C" int sherf_svox_mark_rows(const int32_t* coord, int n, int D, int H, int W, uint32_t* bitmap, sherf_stream_t stream) {
    SHERF_CHECK_ARG(coord && bitmap && n > 0 && D > 0 && H > 0 && W > 0 && (int64_t)D * H * W < 2147483647LL);
    hipLaunchKernelGGL(mark_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), coord, n, D, H, W, bitmap);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_mark_down(const int32_t* keys, const int32_t* n_rows, int D, int H, int W, uint32_t* bitmap_out,
                                    int max_rows, sherf_stream_t stream) {
    SHERF_CHECK_ARG(keys && n_rows && bitmap_out && D > 0 && H > 0 && W > 0 && max_rows > 0);
    hipLaunchKernelGGL(mark_down_kernel, dim3(cdiv(max_rows, 256)), dim3(256), 0, as_stream(stream), keys, n_rows, D, H, W,
                       bitmap_out);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_scan(const uint32_t* bitmap, int n_words, int32_t* prefix, int32_t* n_rows, int32_t* chunk_ws,
                               uint32_t* wp, sherf_stream_t stream) {
    SHERF_CHECK_ARG(bitmap && prefix && n_rows && chunk_ws && wp && n_words > 0);
    const int n_chunks = cdiv(n_words, 1024);
    hipLaunchKernelGGL(scan_local_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), bitmap, n_words, prefix, chunk_ws);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(1), dim3(256), 0, as_stream(stream), chunk_ws, n_chunks, n_rows);
    hipLaunchKernelGGL(scan_add_kernel, dim3(cdiv(n_words, 256)), dim3(256), 0, as_stream(stream), bitmap, prefix, n_words, chunk_ws,
                       reinterpret_cast<uint2*>(wp), (int32_t*)nullptr);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_keys(const uint32_t* bitmap, const int32_t* prefix, int n_words, int32_t* keys, sherf_stream_t stream) {
    SHERF_CHECK_ARG(bitmap && prefix && keys && n_words > 0);
    hipLaunchKernelGGL(keys_kernel, dim3(cdiv(n_words, 256)), dim3(256), 0, as_stream(stream), bitmap, prefix, n_words, keys);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_scatter_rows(const int32_t* coord, const float* feat, int n, int C, int D, int H, int W,
                                       const uint32_t* bitmap, const int32_t* prefix, const int32_t* n_rows, int64_t* acc_fix,
                                       float* g, int32_t* mult, sherf_stream_t stream) {
    SHERF_CHECK_ARG(coord && feat && bitmap && prefix && n_rows && acc_fix && g && mult && n > 0 && C > 0);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((int64_t)n * C, 256)), dim3(256), 0, as_stream(stream), coord, feat, n, C,
                       D, H, W, bitmap, prefix, reinterpret_cast<unsigned long long*>(acc_fix), mult);
    hipLaunchKernelGGL(fix_to_float_kernel, dim3(cdiv((int64_t)n * C, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const long long*>(acc_fix), n_rows, C, g);
    SHERF_LAUNCH_CHECK();
}





extern "C" int sherf_svox_bn_finalize(const int64_t* acc, const int32_t* n_total_rows, int C, const float* gamma, const float* beta,
                                      float* stats, int training, float* bnparam, sherf_stream_t stream) {
    SHERF_CHECK_ARG((!training || (acc && n_total_rows)) && gamma && beta && stats && bnparam && C > 0 && C <= 96);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(128), 0, as_stream(stream), reinterpret_cast<const long long*>(acc),
                       n_total_rows, C, gamma, beta, stats, training, bnparam);
    SHERF_LAUNCH_CHECK();
}

// nn.BatchNorm1d's train-mode side effect for every layer of the encoder in ONE launch (one workgroup per layer):
//   running = (1 - momentum) running + momentum batch;  the variance unbiased (n / (n - 1));  num_batches_tracked += 1
struct BnRunning {
    int n_layers;
    const float* stats[SHERF_SVOX_MAX_LAYERS];      // [2][C] batch mean / biased variance
    float* rmean[SHERF_SVOX_MAX_LAYERS];
    float* rvar[SHERF_SVOX_MAX_LAYERS];
    long long* nbt[SHERF_SVOX_MAX_LAYERS];
    const int32_t* n[SHERF_SVOX_MAX_LAYERS];        // rows the statistics were taken over
    int C[SHERF_SVOX_MAX_LAYERS];
    float momentum[SHERF_SVOX_MAX_LAYERS];
};
__global__ void __launch_bounds__(128) bn_running_kernel(BnRunning u) {
    const int l = blockIdx.x;
    const float m = u.momentum[l], n = (float)u.n[l][0];
    // a level with <= 1 row has no unbiased variance (n / (n - 1)): nn.BatchNorm1d raises there ("Expected more than 1 value per
    // channel"); a device kernel cannot, so the running statistics of such a layer are left untouched instead of turning into inf / NaN
    if (n < 2.0f) return;
    for (int c = threadIdx.x; c < u.C[l]; c += 128) {
        u.rmean[l][c] = u.rmean[l][c] * (1.0f - m) + m * u.stats[l][c];
        u.rvar[l][c] = u.rvar[l][c] * (1.0f - m) + m * u.stats[l][u.C[l] + c] * n / (n - 1.0f);
    }
    if (threadIdx.x == 0 && u.nbt[l]) u.nbt[l][0] += 1;
}

extern "C" int sherf_svox_bn_running_update(int n_layers, const float* const* stats, float* const* running_mean, float* const* running_var,
                                            int64_t* const* num_batches_tracked, const int32_t* const* n_rows, const int32_t* channels,
                                            const float* momentum, sherf_stream_t stream) {
    SHERF_CHECK_ARG(n_layers > 0 && n_layers <= SHERF_SVOX_MAX_LAYERS && stats && running_mean && running_var && num_batches_tracked &&
                    n_rows && channels && momentum);
    BnRunning u;
    u.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        SHERF_CHECK_ARG(stats[l] && running_mean[l] && running_var[l] && n_rows[l] && channels[l] > 0);
        u.stats[l] = stats[l]; u.rmean[l] = running_mean[l]; u.rvar[l] = running_var[l];
        u.nbt[l] = reinterpret_cast<long long*>(num_batches_tracked[l]); u.n[l] = n_rows[l]; u.C[l] = channels[l]; u.momentum[l] = momentum[l];
    }
    hipLaunchKernelGGL(bn_running_kernel, dim3(n_layers), dim3(128), 0, as_stream(stream), u);
    SHERF_LAUNCH_CHECK();
}

static int launch_conv3(const int32_t* keys_out, const int32_t* n_rows_out, int Do, int Ho, int Wo, const uint32_t* wp_in, int Di,
                        int Hi, int Wi, const float* in_raw, int Cin, BnIn bin, const int32_t* in_mult,
                        const void* w_packed, int Cout, int mode, int max_rows, float* out_raw, int64_t* out_acc,
                        sherf_stream_t stream, const uint32_t* in_amax = nullptr) {
    const int out_half = (mode & 512) ? 512 : 0;
    const bool single = (mode & 1024) != 0;               // one fp16 product per term (see sconv3_kernel: SP)
    mode &= ~(512 | 1024);
    SHERF_CHECK_ARG(n_rows_out && in_raw && w_packed && out_raw && (mode == 2 || (keys_out && wp_in)));
    SHERF_CHECK_ARG(Cin >= 16 && Cin <= 96 && Cin % 16 == 0 && (Cout == 32 || Cout == 64 || Cout == 96) && max_rows > 0 && mode >= 0 && mode <= 3);
    SHERF_CHECK_ARG(!in_amax || !bin.bnparam);      // (the input scale applies to raw rows only)
    SHERF_CHECK_ARG(bin.acc == nullptr || (bin.n_total && bin.gamma && bin.beta && bin.stats && bin.bnparam));
    // four waves per workgroup.  Eight (sherf_set_debug bit 12; a wave then walks 3-4 taps instead of ~7 and eight partial tiles are summed
    // in LDS) was built and measured in round 3 and LOST: encoder_done 0.66 -> 0.755 ms, frame 1.29 -> 1.39 ms (profiles/
    // r03_bench_f_sconv8_search2.txt) -- 512-thread workgroups fit one per CU where two 256-thread ones shared it, and the 96 -> 96
    // instance spills at the 256-register cap.
    const bool wide = mode != 2 && (g_sherf_debug & 4096) != 0;
    const int nw = wide ? 8 : 4;
    const size_t smem = (size_t)27 * 32 * 4 + (size_t)3 * Cin * 4 + (size_t)nw * 32 * Cout * 4;
    const dim3 grid(cdiv(max_rows, 32)), block(64 * nw);
    SHERF_CHECK_ARG(!in_mult || bin.bnparam);       // (the multiplicity only enters through the BatchNorm transform)
    const bool fold = mode == 2;
    const int bnm = bin.bnparam ? (in_mult ? 2 : 1) : 0;
    const int kmode = mode | ((g_sherf_debug & 128) ? 256 : 0) | (fold ? out_half : 0);
#define SHERF_CONV3___(N, K, F, B, S, W)                                                                                     \
    do {                                                                                                                          \
        if (smem > 64 * 1024) SHERF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv3_kernel<N, K, F, B, S, W>), \
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));       \
        hipLaunchKernelGGL((sconv3_kernel<N, K, F, B, S, W>), grid, block, smem, as_stream(stream), keys_out, n_rows_out, Do, Ho, Wo, \
                           reinterpret_cast<const uint2*>(wp_in), Di, Hi, Wi, in_raw, bin, in_mult,                             \
                           reinterpret_cast<const uint4*>(w_packed), kmode, out_raw, reinterpret_cast<long long*>(out_acc), trace_id, \
                           in_amax);                                                                                            \
    } while (0)
#define SHERF_CONV3__(N, K, F, B, S) do { if (wide && !F) SHERF_CONV3___(N, K, F, B, S, 8); else SHERF_CONV3___(N, K, F, B, S, 4); } while (0)
#define SHERF_CONV3_(N, K, F, B) do { if (single) SHERF_CONV3__(N, K, F, B, true); else SHERF_CONV3__(N, K, F, B, false); } while (0)
#define SHERF_CONV3(N, K)                                                                                                    \
    do { if (fold) { if (bnm == 2) SHERF_CONV3_(N, K, true, 2); else if (bnm) SHERF_CONV3_(N, K, true, 1); else SHERF_CONV3_(N, K, true, 0); } \
         else { if (bnm == 2) SHERF_CONV3_(N, K, false, 2); else if (bnm) SHERF_CONV3_(N, K, false, 1); else SHERF_CONV3_(N, K, false, 0); } } while (0)
    static int trace_launches = 0;                  // (profiling builds: launch ordinal -> the records' first word)
    const int trace_id = SHERF_SCONV_TRACE ? trace_launches++ : 0;
    const int sel = (Cout / 32) * 10 + Cin / 16;
    // column split (SHERF_EXPERIMENT bit 12): the two 96-column single-product instances as three 32-column workgroups per row tile
    // (grid.y = 3; see sconv3_kernel: CS) -- the same bits, a third of the accumulators and weight fragments per wave
    if ((sel == 34 || sel == 36) && !fold && single && !wide && bnm && (sherf_experiment() & 4096)) {
        const size_t smem_cs = (size_t)27 * 32 * 4 + (size_t)3 * Cin * 4 + (size_t)4 * 32 * 32 * 4;
        const dim3 grid_cs(cdiv(max_rows, 32), 3);
#define SHERF_CONV3_CS(K, B)                                                                                                     \
        hipLaunchKernelGGL((sconv3_kernel<3, K, false, B, true, 4, true>), grid_cs, block, smem_cs, as_stream(stream), keys_out, n_rows_out, \
                           Do, Ho, Wo, reinterpret_cast<const uint2*>(wp_in), Di, Hi, Wi, in_raw, bin, in_mult,                  \
                           reinterpret_cast<const uint4*>(w_packed), kmode, out_raw, reinterpret_cast<long long*>(out_acc), trace_id, in_amax)
        if (sel == 34) { if (bnm == 2) SHERF_CONV3_CS(4, 2); else SHERF_CONV3_CS(4, 1); }
        else { if (bnm == 2) SHERF_CONV3_CS(6, 2); else SHERF_CONV3_CS(6, 1); }
#undef SHERF_CONV3_CS
        SHERF_LAUNCH_CHECK();
    }
    switch (sel) {
        case 12: SHERF_CONV3(1, 2); break;     // 32 -> 32
        case 22: SHERF_CONV3(2, 2); break;     // 32 -> 64
        case 24: SHERF_CONV3(2, 4); break;     // 64 -> 64
        case 34: SHERF_CONV3(3, 4); break;     // 64 -> 96
        case 36: SHERF_CONV3(3, 6); break;     // 96 -> 96
        case 32: SHERF_CONV3(3, 2); break;     // 32 -> 96 (fold)
        // the two transposed channel pairs only the input gradient of the stride-2 layers needs (sherf_svox_conv3_dgrad: raw rows, three
        // products, four waves -- one instance each instead of the 24 of a forward pair)
        case 14: if (fold || bnm || single) goto unsupported; SHERF_CONV3___(1, 4, false, 0, false, 4); break;     // 64 -> 32
        case 26: if (fold || bnm || single) goto unsupported; SHERF_CONV3___(2, 6, false, 0, false, 4); break;     // 96 -> 64
        default:
        unsupported:
            snprintf(g_sherf_err, sizeof(g_sherf_err), "sherf_svox_conv3: unsupported channel pair %d -> %d", Cin, Cout);
            return SHERF_EINVAL;
    }
#undef SHERF_CONV3
#undef SHERF_CONV3_
#undef SHERF_CONV3__
#undef SHERF_CONV3___
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_conv3(const int32_t* keys_out, const int32_t* n_rows_out, int Do, int Ho, int Wo,
                                const uint32_t* wp_in, int Di, int Hi, int Wi, const float* in_raw, int Cin,
                                const float* in_bn, const int32_t* in_mult, const void* w_packed, int Cout, int mode,
                                int max_rows, float* out_raw, int64_t* out_acc, sherf_stream_t stream) {
    BnIn bin{nullptr, nullptr, nullptr, nullptr, nullptr, const_cast<float*>(in_bn)};
    return launch_conv3(keys_out, n_rows_out, Do, Ho, Wo, wp_in, Di, Hi, Wi, in_raw, Cin, bin, in_mult, w_packed, Cout, mode,
                        max_rows, out_raw, out_acc, stream);
}

// Input gradient of a sparse convolution on the same MFMA kernel: d_in[i][ci] = sum_k sum_co d_raw[o(i, k)][co] W[co][k][ci] is a sparse
// convolution of d_raw with the taps mirrored and the channel roles exchanged -- w_packed_t = pack_conv_weights of wt[k'][co][ci] =
// W[co][26 - k'][ci] (sherf_amd/backward_dense.py) -- over the neighbour rule of the layer (mode 0) or its transpose (stride 2: mode 3).
extern "C" int sherf_svox_conv3_dgrad(const int32_t* keys_i, const int32_t* n_rows_i, int Di, int Hi, int Wi, const uint32_t* wp_o, int Do,
                                      int Ho, int Wo, const float* d_raw, int Cout, const uint32_t* d_raw_amax, const void* w_packed_t,
                                      int Cin, int down, int max_rows, float* d_in, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d_raw_amax && (down == 0 || down == 1));
    BnIn bin{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return launch_conv3(keys_i, n_rows_i, Di, Hi, Wi, wp_o, Do, Ho, Wo, d_raw, Cout, bin, nullptr, w_packed_t, Cin, down ? 3 : 0, max_rows,
                        d_in, nullptr, stream, d_raw_amax);
}

static int scan_level(const sherf_svox_level_ws& l, sherf_stream_t stream) {
    const int n_chunks = cdiv(l.n_words, 1024);
    hipLaunchKernelGGL(scan_local_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), l.bitmap, l.n_words, l.prefix, l.chunk_ws);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(1), dim3(256), 0, as_stream(stream), l.chunk_ws, n_chunks, l.n_rows);
    hipLaunchKernelGGL(scan_add_kernel, dim3(cdiv(l.n_words, 256)), dim3(256), 0, as_stream(stream), l.bitmap, l.prefix, l.n_words,
                       l.chunk_ws, reinterpret_cast<uint2*>(l.wp), l.keys);
    SHERF_LAUNCH_CHECK();
}

// The whole encoder as one native call: 1 memset + 6 launches for level 0, 4 per down-sampling, 1 per conv, 1 per tapped
// level for the fold (no BatchNorm finalize launches: see BnIn).  Everything is enqueued on `stream`; nothing is read back.
// (A finalize fused into the conv's last workgroup was measured and rejected: the device-scope fence every workgroup
// needs writes back / invalidates its XCD's L2, which slowed the conv 2x and every kernel running next to it.)
// `ev` (optional) is recorded after layer `ev_layer`: lets the frame driver start other work mid-chain.
// `aux` + `lev_ev[4]` (optional): the occupancy structure of levels 1-3 depends only on the voxel coordinates, not on any
// feature, so it is built on a second stream while the level-0 convolutions run (12 launches off the dependent chain).
// With `aux` the three pointwise folds also leave the chain: their output is first read by the gather, so they run on `aux`
// behind an event per tapped layer (lev_ev[4..6]; lev_ev[7] joins aux back into `stream` at the end).  `after_levels`
// (optional) is called once the level builds are queued on aux, before the folds: lets the caller slot in aux work that
// must not wait behind them.
int sherf_svox_encode_impl(const sherf_svox_plan* p, const int32_t* coord, const float* feat, int n, int training,
                           sherf_vox_level* levels_out_host, sherf_stream_t stream, hipEvent_t ev, int ev_layer,
                           sherf_stream_t aux, hipEvent_t* lev_ev, const std::function<int()>* after_levels, int fold_half) {
    SHERF_CHECK_ARG(p && coord && feat && n > 0 && levels_out_host && p->n_layers > 0 && p->n_layers <= SHERF_SVOX_MAX_LAYERS);
    SHERF_CHECK_ARG(p->zero_ptr && p->zero_bytes > 0 && p->acc_fix && p->g0 && p->mult && p->n_total);
    SHERF_CHECK_ARG(aux == nullptr || (lev_ev != nullptr && aux != stream));
    for (int i = 0; i < 4; ++i) {
        const sherf_svox_level_ws& l = p->lev[i];
        SHERF_CHECK_ARG(l.bitmap && l.prefix && l.n_rows && l.chunk_ws && l.wp && l.keys && l.n_words > 0 && l.cap > 0 && l.D > 0 && l.H > 0 && l.W > 0);
    }
    SHERF_HOST_STAMP(sherf_experiment(), "encoder first launch");
    // (SHERF_EXPERIMENT bit 5: timing events along the encoder's stream, printed at the next frame's entry -- tools/host_stamps.py --trail)
    // The trail's state is process-wide and only ever touched with the bit set, under its own lock (it used to be reset by every call of any thread).
    static hipEvent_t xe[32];
    static const char* xw[32];
    static int xn = 0, xmade = 0;
    static std::mutex trail_mu;
    const bool trail = (sherf_experiment() & 32) != 0;
    std::unique_lock<std::mutex> trail_lock(trail_mu, std::defer_lock);
    if (trail) trail_lock.lock();
    if (trail && xn > 0) {
        SHERF_HIP_CHECK(hipEventSynchronize(xe[xn - 1]));
        fprintf(stderr, "[trail]");
        for (int i = 1; i < xn; ++i) {
            float ms = 0.f;
            SHERF_HIP_CHECK(hipEventElapsedTime(&ms, xe[0], xe[i]));
            fprintf(stderr, " %s %.0f", xw[i], ms * 1e3f);
        }
        fprintf(stderr, "\n");
    }
    if (trail) xn = 0;
    auto mark = [&](const char* what) -> int {
        if (!trail || xn >= 32) return SHERF_OK;
        if (xn >= xmade) { SHERF_HIP_CHECK(hipEventCreate(&xe[xn])); xmade = xn + 1; }
        xw[xn] = what;
        SHERF_HIP_CHECK(hipEventRecord(xe[xn++], as_stream(stream)));
        return SHERF_OK;
    };
    SHERF_RUN(mark("entry"));
    SHERF_HIP_CHECK(hipMemsetAsync(p->zero_ptr, 0, (size_t)p->zero_bytes, as_stream(stream)));
    const sherf_svox_level_ws& l0 = p->lev[0];
    hipLaunchKernelGGL(mark_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), coord, n, l0.D, l0.H, l0.W, l0.bitmap);
    SHERF_RUN(scan_level(l0, stream));
    SHERF_RUN(mark("level0"));
    auto build_level = [&](int k, sherf_stream_t st) -> int {       // level k from level k-1 (SparseConv3d k3 s2 p1 output sites)
        const sherf_svox_level_ws& src = p->lev[k - 1];
        hipLaunchKernelGGL(mark_down_kernel, dim3(cdiv(src.cap, 256)), dim3(256), 0, as_stream(st), src.keys, src.n_rows, src.D,
                           src.H, src.W, p->lev[k].bitmap);
        return scan_level(p->lev[k], st);
    };
    const int xp = sherf_experiment();
    const bool levels_inline = aux && (xp & 2);
    if (xp & 1)
        SHERF_RUN(sherf_svox_scatter_rows(coord, feat, n, 32, l0.D, l0.H, l0.W, l0.bitmap, l0.prefix, l0.n_rows, p->acc_fix, p->g0,
                                          p->mult, stream));
    if (aux && !levels_inline) {
        SHERF_HIP_CHECK(hipEventRecord(lev_ev[0], as_stream(stream)));
        SHERF_HIP_CHECK(hipStreamWaitEvent(as_stream(aux), lev_ev[0], 0));
        for (int k = 1; k < 4; ++k) {
            SHERF_RUN(build_level(k, aux));
            SHERF_HIP_CHECK(hipEventRecord(lev_ev[k], as_stream(aux)));
        }
    }
    SHERF_HOST_STAMP(xp, "levels queued");
    if (after_levels) SHERF_RUN((*after_levels)());
    SHERF_HOST_STAMP(xp, "scatter_rows next");
    SHERF_GPU_STAMP(xp, as_stream(stream), "encoder stream reached scatter_rows");
    if (!(xp & 1))
        SHERF_RUN(sherf_svox_scatter_rows(coord, feat, n, 32, l0.D, l0.H, l0.W, l0.bitmap, l0.prefix, l0.n_rows, p->acc_fix, p->g0,
                                          p->mult, stream));
    SHERF_RUN(mark("rows"));
    int lev = 0, ntap = 0;
    const float* cur = p->g0;
    BnIn cur_bn{};                                      // BatchNorm of `cur` (none for the raw level-0 features)
    auto bn_of = [&](const sherf_svox_layer& ly, int dlev) {
        return BnIn{training ? reinterpret_cast<const long long*>(ly.acc) : nullptr, dlev == 0 ? p->n_total : p->lev[dlev].n_rows,
                    ly.gamma, ly.beta, ly.stats, ly.bnparam};
    };
    for (int li = 0; li < p->n_layers; ++li) {
        const sherf_svox_layer& ly = p->layers[li];
        SHERF_CHECK_ARG(ly.wt && ly.gamma && ly.beta && ly.stats && ly.bnparam && ly.out && ly.acc);
        SHERF_CHECK_ARG(lev + (ly.down ? 1 : 0) < 4);
        const int dlev = lev + (ly.down ? 1 : 0);
        const sherf_svox_level_ws& src = p->lev[lev];
        const sherf_svox_level_ws& dst = p->lev[dlev];
        if (ly.down) {
            if (aux && !levels_inline) SHERF_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), lev_ev[dlev], 0));
            else SHERF_RUN(build_level(dlev, stream));
        }
        // training: the conv leaves fixed-point sums of its output in ly.acc, resolved by its consumers' prologues.
        // eval: bnparam comes from the running statistics and was prepared by the caller (sherf_svox_bn_finalize with
        // training = 0) when they last changed.
        SHERF_RUN(launch_conv3(dst.keys, dst.n_rows, dst.D, dst.H, dst.W, src.wp, src.D, src.H, src.W, cur, ly.cin, cur_bn,
                               (lev == 0 && cur_bn.bnparam) ? p->mult : nullptr, ly.wt, ly.cout, (ly.down ? 1 : 0) | ((fold_half & 2) ? 1024 : 0),
                               dst.cap, ly.out, training ? ly.acc : nullptr, stream));
        if (ev && li == ev_layer) SHERF_HIP_CHECK(hipEventRecord(ev, as_stream(stream)));
        SHERF_RUN(mark(ly.down ? "down" : "conv"));
        lev = dlev; cur = ly.out; cur_bn = bn_of(ly, dlev);
        if (ly.tap) {
            SHERF_CHECK_ARG(ntap < 3 && p->fold_mat[ntap] && p->fold_rows[ntap]);
            sherf_stream_t fst = stream;
            if (aux) {
                SHERF_HIP_CHECK(hipEventRecord(lev_ev[4 + ntap], as_stream(stream)));
                SHERF_HIP_CHECK(hipStreamWaitEvent(as_stream(aux), lev_ev[4 + ntap], 0));
                fst = aux;
            }
            SHERF_RUN(launch_conv3(nullptr, dst.n_rows, 1, 1, 1, nullptr, 1, 1, 1, ly.out, ly.cout, cur_bn, nullptr,
                                   p->fold_mat[ntap], 96, 2 | ((fold_half & 1) ? 512 : 0) | ((fold_half & 2) ? 1024 : 0), dst.cap,
                                   p->fold_rows[ntap], nullptr, fst));
            levels_out_host[ntap].wp = dst.wp;
            levels_out_host[ntap].rows = p->fold_rows[ntap];
            levels_out_host[ntap].D = dst.D; levels_out_host[ntap].H = dst.H; levels_out_host[ntap].W = dst.W;
            ++ntap;
        }
    }
    SHERF_HOST_STAMP(xp, "encoder queued");
    SHERF_CHECK_ARG(ntap == 3);
    if (aux) {                                          // the caller's "encoder done" event on `stream` covers aux too
        SHERF_HIP_CHECK(hipEventRecord(lev_ev[7], as_stream(aux)));
        SHERF_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), lev_ev[7], 0));
    }
    return SHERF_OK;
}

extern "C" int sherf_svox_encode(const sherf_svox_plan* p, const int32_t* coord, const float* feat, int n, int training,
                                 sherf_vox_level* levels_out_host, sherf_stream_t stream) {
    return sherf_svox_encode_impl(p, coord, feat, n, training, levels_out_host, stream, nullptr, -1, nullptr, nullptr, nullptr, 0);
}
