// Dense GEMMs of the backward pass (BASELINE config 5) on MFMA: sherf_bwd_gemm, row-major
//     C[M,N] = op(A)[M,K] . op(B)[K,N] + beta C.
// Every call of the backward has one HUGE dimension (the valid samples, ~7e5) and two small ones (layer widths <= 199), in one
// of three patterns (sherf_amd/backward_dense.py):
//     forward recompute   y  = x  . W^T   (transA 0, transB 1)   M = samples
//     data gradient       dx = dy . W     (transA 0, transB 0)   M = samples
//     weight gradient     dW = dy^T . x   (transA 1, transB 0)   K = samples
// Round 1 sent them to rocBLAS sgemm; these are hand-written gfx950 kernels on v_mfma_f32_32x32x16_bf16 with each operand split
// into THREE bf16 parts x = hi + mid + lo (the splits are exact in fp32) and the six products of weight >= 2^-16 kept
// (lo.hi, mid.mid, hi.lo, mid.hi, hi.mid, hi.hi; small terms first; fp32 accumulate): ~2^-23 relative, fp32 grade.  Why not the
// forward's cheaper schemes (measured on the MI355X in round 2):
//   * fp16 hi + lo (the forward's f16x3): gradients are 1e-4 ... 1e-9 in magnitude, where fp16 (min normal 6e-5) has lost its precision
//     or underflowed -- d_tokens_in came out 5e-3 off;
//   * bf16 hi + lo (2^-16): range is fine, but the RECOMPUTED pre-activations carry 5e-5 of error, which flips ~5e-4 of the ReLU masks;
//     a flipped mask changes a sample's gradient by O(1).
// These GEMMs are < 1 % of the training step (section 8 of DESIGN.md), so the doubled MFMA count is free.
//   tall_gemm  (transA 0): the small matrix is converted ONCE per workgroup into B-operand fragments in LDS (<= 104 KiB); persistent
//              4-wave workgroups walk 32-row tiles of the tall operand, each wave its own tile, NT independent accumulator chains.
//   wgrad_gemm (transA 1, transB 0): workgroups walk 16-sample steps of a slab of rows; both operand fragments are read straight
//              from global memory (a lane's 8 K-values are 8 consecutive ROWS: 32 lanes read 32 consecutive floats of one row, coalesced);
//              the waves split the output row tiles; partial sums are added to C with fp32 atomics (the reference does not require
//              a deterministic reduction order, SURVEY section 7 hard part 6).
// Any other shape / transposition goes to a plain fp32 kernel (correctness only: no layer shape of the path reaches it,
// tests/test_hipcpu_kernels.py; a tall product whose B fragments exceed the LDS is cut into column slices instead).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Frag { u32x4 hi, mid, lo; };        // 8 K-values of one lane, three bf16 parts each (two values per dword)

// part = the top 16 bits of what is left (truncation: the remainder x - part is exact in fp32; one v_perm packs a pair)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const uint32_t ua = __builtin_bit_cast(uint32_t, a), ub = __builtin_bit_cast(uint32_t, b);
    hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);          // [a.hi16 | b.hi16 << 16]
    const float ra = a - __builtin_bit_cast(float, ua & 0xFFFF0000u), rb = b - __builtin_bit_cast(float, ub & 0xFFFF0000u);
    const uint32_t va = __builtin_bit_cast(uint32_t, ra), vb = __builtin_bit_cast(uint32_t, rb);
    mid = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __builtin_bit_cast(float, va & 0xFFFF0000u), sb = rb - __builtin_bit_cast(float, vb & 0xFFFF0000u);
    lo = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, sb), __builtin_bit_cast(uint32_t, sa), 0x07060302u);
}
__device__ __forceinline__ Frag split8(const float (&v)[8]) {
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) split2(v[2 * p], v[2 * p + 1], h[p], m[p], l[p]);
    return Frag{u32x4{h[0], h[1], h[2], h[3]}, u32x4{m[0], m[1], m[2], m[3]}, u32x4{l[0], l[1], l[2], l[3]}};
}
__device__ __forceinline__ f32x16 mfma1(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma6(const Frag& a, const Frag& b, f32x16 c) {
    c = mfma1(a.lo, b.hi, c); c = mfma1(a.mid, b.mid, c); c = mfma1(a.hi, b.lo, c);
    c = mfma1(a.mid, b.hi, c); c = mfma1(a.hi, b.mid, c);
    return mfma1(a.hi, b.hi, c);
}
// accumulator register r of lane (j = lane & 31, h = lane >> 5) <-> tile row (r & 3) + 8 (r >> 2) + 4 h, tile column j
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------------------------
// tall: C[M,N] = A[M,K] . S + beta C,   S[k][n] = transB ? B[n * ldb + k] : B[k * ldb + n]
// ---------------------------------------------------------------------------------------------------------------
// Eight waves per workgroup (two per SIMD) share the B fragments: with four, a CU whose LDS holds one workgroup's fragments (96 KiB at
// N = K = 128) ran ONE wave per SIMD -- nothing to overlap a tile's A loads and operand splits with another tile's MFMAs (35 % of the MFMA
// rate its six products per term allow, profiles/r03_train_step_j_rocprofv3_stats.txt).
constexpr int kTallWaves = 8;
template <int NT>
__global__ void __launch_bounds__(64 * kTallWaves) tall_gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int transB,
                                                        float* __restrict__ C, int ldc, int M, int N, int K, float beta,
                                                        const float* __restrict__ bias, int act) {
    extern __shared__ __attribute__((aligned(16))) u32x4 s_frag[];        // [nkb][NT][hi, mid, lo][64 lanes]
    const int nkb = (K + 15) / 16;
    for (int idx = threadIdx.x; idx < nkb * NT * 64; idx += 64 * kTallWaves) {
        const int l = idx & 63, nt = (idx >> 6) % NT, kb = idx / (64 * NT);
        const int n = 32 * nt + (l & 31), k0 = 16 * kb + 8 * (l >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            v[e] = (k < K && n < N) ? (transB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n]) : 0.f;
        }
        const Frag f = split8(v);
        s_frag[((kb * NT + nt) * 3) * 64 + l] = f.hi;
        s_frag[((kb * NT + nt) * 3 + 1) * 64 + l] = f.mid;
        s_frag[((kb * NT + nt) * 3 + 2) * 64 + l] = f.lo;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const bool vec = (lda % 4 == 0) && ((reinterpret_cast<size_t>(A) & 15) == 0);
    const int n_tiles = (M + 31) / 32;
    // bias of this lane's column of every tile, read once: in the store loop it
    // was re-read per element behind the stores (the product got 30 % slower, profiles/r03_train_step_j_rocprofv3_stats.txt)
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (bias && 32 * nt + i < N) ? bias[32 * nt + i] : 0.f;
    const bool relu = act == 1;
    for (int tile = blockIdx.x * kTallWaves + wave; tile < n_tiles; tile += gridDim.x * kTallWaves) {
        const int row = tile * 32 + i;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = tile * 32 + acc_row(r, h), cc = 32 * nt + i;
                acc[nt][r] = (beta != 0.f && rr < M && cc < N) ? beta * C[(size_t)rr * ldc + cc] : 0.f;
            }
        const float* arow = A + (size_t)(row < M ? row : M - 1) * lda;
        for (int kb = 0; kb < nkb; ++kb) {
            const int k0 = 16 * kb + 8 * h;
            float v[8];
            if (vec && k0 + 8 <= K) {
                const float4 a = *reinterpret_cast<const float4*>(arow + k0), b = *reinterpret_cast<const float4*>(arow + k0 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = k0 + e < K ? arow[k0 + e] : 0.f;
            }
            if (row >= M) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            const Frag a = split8(v);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4* f = s_frag + ((kb * NT + nt) * 3) * 64 + lane;
                acc[nt] = mfma6(a, Frag{f[0], f[64], f[128]}, acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = tile * 32 + acc_row(r, h), cc = 32 * nt + i;
                if (rr < M && cc < N) { const float v = acc[nt][r] + bv[nt]; C[(size_t)rr * ldc + cc] = relu ? fmaxf(v, 0.f) : v; }   // epilogue of sherf_bwd_gemm_bias_act
            }
    }
}

// The same product for the shapes the decoder / transformer backward is made of (beta == 0, the number of 16-wide K-blocks known at compile time,
// rows of A 16-byte aligned and padded to whole K-blocks): a streaming kernel.  Round 5: the general kernel above loads a K-block of A right before it splits it -- 1 KiB in flight
// per wave against the ~60 KiB per CU that HBM latency x bandwidth asks for -- and ran 314 us per [750 000, 128] x [128, 128] product where the
// bytes (768 MB) take ~140 us and the six-product MFMA stream ~80 us.  Here a wave fetches the WHOLE next tile of A (NKB x 2 dwordx4 per lane, 16 KiB
// per wave, 128 KiB per CU) before it starts the current tile's MFMAs; the tile loop has no per-element bounds checks (only the last tile's rows and
// the last column tile's columns are guarded).  Same arithmetic, same order of the six products and of the K-blocks: bit-identical to the kernel above.
// FUSE (the data gradient of a layer whose input came out of a ReLU; sherf_bwd_gemm_dgrad_fused): in the store, C += r1_s[row] r1_w[column] (the
// rank-one product of the sigma head joining the feature head's gradient), C = 0 where mask <= 0 (the ReLU of the layer below), colsum[column] += the
// column sums of what was stored (that layer's bias gradient) -- the pass sherf_bwd_relu_mask_colsum made over the result afterwards, and a K = 1 GEMM.
struct DgradFuse { const float* r1_s; int r1_lds; const float* r1_w; const float* mask; int ldm; float* colsum; const float* addend; int ld_add; };
// (addend: a residual stream joining the product in the store -- sherf_bwd_gemm_bias_act_add: C = act(A . op(B) + bias) + addend)

template <int NT, int NKB, bool FUSE = false>
__global__ void __launch_bounds__(64 * kTallWaves) tall_stream_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int transB,
                                                          float* __restrict__ C, int ldc, int M, int N, int K, const float* __restrict__ bias, int act,
                                                          DgradFuse fz) {
    // K in (16 (NKB - 1), 16 NKB]: a row of A is read in whole 16-float blocks (the launcher checks lda >= 16 NKB: the tail of the last block lies in
    // the row's own padding) and the elements past K are zeroed in registers (the padding is never written: it may hold anything, NaN included)
    extern __shared__ __attribute__((aligned(16))) u32x4 s_frag[];        // [NKB][NT][hi, mid, lo][64 lanes]
    for (int idx = threadIdx.x; idx < NKB * NT * 64; idx += 64 * kTallWaves) {
        const int l = idx & 63, nt = (idx >> 6) % NT, kb = idx / (64 * NT);
        const int n = 32 * nt + (l & 31), k0 = 16 * kb + 8 * (l >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (n < N && k0 + e < K) ? (transB ? B[(size_t)n * ldb + k0 + e] : B[(size_t)(k0 + e) * ldb + n]) : 0.f;
        const Frag f = split8(v);
        s_frag[((kb * NT + nt) * 3) * 64 + l] = f.hi;
        s_frag[((kb * NT + nt) * 3 + 1) * 64 + l] = f.mid;
        s_frag[((kb * NT + nt) * 3 + 2) * 64 + l] = f.lo;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int n_tiles = (M + 31) / 32, stride = gridDim.x * kTallWaves;
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (bias && 32 * nt + i < N) ? bias[32 * nt + i] : 0.f;
    float w1[NT], csum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { w1[nt] = (FUSE && fz.r1_w && 32 * nt + i < N) ? fz.r1_w[32 * nt + i] : 0.f; csum[nt] = 0.f; }
    const bool relu = act == 1;
    // A in two halves of the K-blocks: while a half is multiplied the other half (of this tile, then of the next) is in flight -- 8 KiB per wave, 64 KiB per CU.
    // (The whole next tile in registers -- 128 of them beside 64 accumulators -- spilled.)
    constexpr int HA = (NKB + 1) / 2, HB = NKB - HA;
    float4 b0[HA][2], b1[HB > 0 ? HB : 1][2];
    auto rowp = [&](int t) {
        const int row = min(t * 32 + i, M - 1);               // (rows past M repeat the last one: computed, never stored)
        return reinterpret_cast<const float4*>(A + (size_t)row * lda + 8 * h);
    };
    int tile = blockIdx.x * kTallWaves + wave;
    if (tile < n_tiles) {
        const float4* p = rowp(tile);
#pragma unroll
        for (int kb = 0; kb < HA; ++kb) { b0[kb][0] = p[4 * kb]; b0[kb][1] = p[4 * kb + 1]; }
    }
    for (; tile < n_tiles; tile += stride) {
        {
            const float4* p = rowp(tile);
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) { b1[kb][0] = p[4 * (HA + kb)]; b1[kb][1] = p[4 * (HA + kb) + 1]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        auto block = [&](const float4 (&q)[2], int kb) {
            float v[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w};
            if (kb == NKB - 1 && K < 16 * NKB) {                 // (compile-time position, wave-uniform condition)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 16 * kb + 8 * h + e < K ? v[e] : 0.f;
            }
            const Frag a = split8(v);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4* f = s_frag + ((kb * NT + nt) * 3) * 64 + lane;
                acc[nt] = mfma6(a, Frag{f[0], f[64], f[128]}, acc[nt]);
            }
        };
#pragma unroll
        for (int kb = 0; kb < HA; ++kb) block(b0[kb], kb);
        __builtin_amdgcn_sched_barrier(0);
        if (tile + stride < n_tiles) {
            const float4* p = rowp(tile + stride);
#pragma unroll
            for (int kb = 0; kb < HA; ++kb) { b0[kb][0] = p[4 * kb]; b0[kb][1] = p[4 * kb + 1]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) block(b1[kb], HA + kb);
        const bool full = tile * 32 + 32 <= M;                  // wave-uniform
        if constexpr (FUSE) {
            // (one branch around each GROUP of loads: `ptr ? load : constant` per element made the compiler branch around every single load and wait for it)
            float s1[16];                                       // the rank-one term's row factors of this lane's 16 rows
            if (fz.r1_s) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s1[r] = fz.r1_s[(size_t)min(tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2), M - 1) * fz.r1_lds];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s1[r] = 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cc = 32 * nt + i;
                if (cc < N) {
                    const int row0 = tile * 32 + 4 * h;
                    float mv[16];
                    if (fz.mask) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) mv[r] = fz.mask[(size_t)min(row0 + (r & 3) + 8 * (r >> 2), M - 1) * fz.ldm + cc];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) mv[r] = 1.f;
                    }
                    float av[16];
                    if (fz.addend) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) av[r] = fz.addend[(size_t)min(row0 + (r & 3) + 8 * (r >> 2), M - 1) * fz.ld_add + cc];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) av[r] = 0.f;
                    }
                    float ov[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[nt][r] + bv[nt];
                        v = relu ? fmaxf(v, 0.f) : v;
                        v = fmaf(s1[r], w1[nt], v) + av[r];
                        ov[r] = mv[r] > 0.f ? v : 0.f;
                    }
                    float* cp = C + (size_t)row0 * ldc + cc;
                    if (full) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) { cp[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = ov[r]; csum[nt] += ov[r]; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dr = (r & 3) + 8 * (r >> 2);
                            if (row0 + dr < M) { cp[(size_t)dr * ldc] = ov[r]; csum[nt] += ov[r]; }
                        }
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cc = 32 * nt + i;
            if (cc < N) {
                float* cp = C + (size_t)(tile * 32 + 4 * h) * ldc + cc;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[nt][r] + bv[nt];
                        cp[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = relu ? fmaxf(v, 0.f) : v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        const float v = acc[nt][r] + bv[nt];
                        if (tile * 32 + 4 * h + dr < M) cp[(size_t)dr * ldc] = relu ? fmaxf(v, 0.f) : v;
                    }
                }
            }
        }
    }
    if constexpr (FUSE) {
        if (fz.colsum) {                                        // (uniform over the grid)
            __syncthreads();                                    // every wave is past its last read of the fragment image: reuse it
            float* red = reinterpret_cast<float*>(s_frag);      // [kTallWaves][32 NT]
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float t = csum[nt];
                t += __shfl_xor(t, 32);                         // the two row halves of the tile
                if (h == 0) red[wave * 32 * NT + 32 * nt + i] = t;
            }
            __syncthreads();
            for (int c = threadIdx.x; c < 32 * NT; c += 64 * kTallWaves)
                if (c < N) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < kTallWaves; ++w) t += red[w * 32 * NT + c];
                    unsafeAtomicAdd(fz.colsum + c, t);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad: C[M,N] += A[Kbig,M]^T . B[Kbig,N]   (C pre-scaled by beta by the caller-side kernel below); M <= 256, N <= 32 NT
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSlab = 512;          // rows of the huge dimension per workgroup visit
// steps of raw operand rows a wave of the round-5 kernels keeps in flight: as many as fit under the 63 outstanding loads the wave's counter can tell apart
// (four stages of 16 dword loads made the compiler wait for the oldest stage at every refill)
constexpr int wg_ring(int loads_per_stage) { return 56 / loads_per_stage < 1 ? 1 : 56 / loads_per_stage > 4 ? 4 : 56 / loads_per_stage; }

template <int NT, int MTW>          // MTW = row tiles of C per wave (the 4 waves take tiles w, w + 4)
__global__ void __launch_bounds__(256) wgrad_gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int N, int Kbig) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int MT = (M + 31) / 32;
    f32x16 acc[MTW][NT];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][nt][r] = 0.f;
    for (int slab = blockIdx.x; slab * kSlab < Kbig; slab += gridDim.x) {
        const int r_end = min(Kbig, (slab + 1) * kSlab);
        for (int r0 = slab * kSlab; r0 < r_end; r0 += 16) {
            Frag b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = 32 * nt + i;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = r0 + 8 * h + e;
                    v[e] = (r < r_end && n < N) ? B[(size_t)r * ldb + n] : 0.f;
                }
                b[nt] = split8(v);
            }
#pragma unroll
            for (int a = 0; a < MTW; ++a) {
                const int mt = wave + 4 * a;
                if (mt < MT) {                                  // wave-uniform
                    const int m = 32 * mt + i;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = r0 + 8 * h + e;
                        v[e] = (r < r_end && m < M) ? A[(size_t)r * lda + m] : 0.f;
                    }
                    const Frag af = split8(v);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[a][nt] = mfma6(af, b[nt], acc[a][nt]);
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < MTW; ++a) {
        const int mt = wave + 4 * a;
        if (mt >= MT) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = 32 * mt + acc_row(r, h), cc = 32 * nt + i;
                if (mm < M && cc < N && acc[a][nt][r] != 0.f) unsafeAtomicAdd(C + (size_t)mm * ldc + cc, acc[a][nt][r]);
            }
    }
}

// Round 5: the kernel above has every one of its four waves split ALL of B's column tiles for the one row tile it owns -- at 128 x 128 that is
// five operand splits (1 120 VALU cycles) per 24 MFMAs (768 cycles) per wave and 16-row step, four times the same B work per workgroup -- and leaves
// waves idle when C has fewer than four row tiles.  Two kernels replace it on the shapes of the path (same six products per term, fp32 atomics into C):
//   wgrad_shared_kernel<MT, NT> (two or more row tiles): wave w owns row tiles w, w + 4; each B column tile is split ONCE per workgroup (by wave
//     nt % 4) into a double-buffered LDS image every wave reads (one barrier per step); a wave's raw operand rows of step s + 1 are in flight while
//     step s is multiplied;
//   wgrad_solo_kernel<NT> (one row tile: M <= 32): the four waves take different 16-row steps of the slab (no sharing, no barrier).
// an operand element that is loaded unconditionally (clamped address) and zeroed by its bit mask when its row / column is not real: with
// `ok ? load : 0` the compiler branched around every single load and waited for it before the next (706 us per 128 x 128)
__device__ __forceinline__ float masked(float v, uint32_t m) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & m); }

template <int MT, int NT>
__global__ void __launch_bounds__(256) wgrad_shared_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                           float* __restrict__ C, int ldc, int M, int N, int Kbig) {
    constexpr int MTW = (MT + 3) / 4, BW = (NT + 3) / 4;             // row tiles / B column tiles (to split) per wave
    constexpr int kWgRing = wg_ring(8 * (MTW + BW));
    __shared__ __attribute__((aligned(16))) u32x4 s_b[2][NT][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    f32x16 acc[MTW][NT];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][nt][r] = 0.f;
    // this lane's column of every operand tile it loads, clamped into the matrix (loads stay in bounds; the value is zeroed when the column is not real)
    int am[MTW], bn[BW];
    bool aok[MTW], bok[BW];
#pragma unroll
    for (int a = 0; a < MTW; ++a) { const int m = 32 * (wave + 4 * a) + i; aok[a] = wave + 4 * a < MT && m < M; am[a] = aok[a] ? m : 0; }
#pragma unroll
    for (int b = 0; b < BW; ++b) { const int n = 32 * (wave + 4 * b) + i; bok[b] = wave + 4 * b < NT && n < N; bn[b] = bok[b] ? n : 0; }
    uint32_t amask[MTW], bmask[BW];
#pragma unroll
    for (int a = 0; a < MTW; ++a) amask[a] = aok[a] ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int b = 0; b < BW; ++b) bmask[b] = bok[b] ? 0xFFFFFFFFu : 0u;
    // raw operand rows of the next kWgRing steps: dword loads (a lane's 8 K-values are 8 different ROWS of memory) are latency-bound one step ahead
    // (measured: 676 us per 128 x 128 against round 2's 325): several steps ahead (wg_ring)
    float ra[kWgRing][MTW][8], rb[kWgRing][BW][8];
    auto fetch = [&](float (&da)[MTW][8], float (&db)[BW][8], int r0, int r_end) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = r0 + 8 * h + e;
            const bool in = r < r_end;
            const size_t rr = in ? r : r_end - 1;
#pragma unroll
            for (int a = 0; a < MTW; ++a) da[a][e] = A[rr * lda + am[a]];          // raw: the masks are applied where the value is USED (applied here,
#pragma unroll
            for (int b = 0; b < BW; ++b) db[b][e] = B[rr * ldb + bn[b]];          // every load is waited for on the spot)
        }
    };
    auto clean = [&](float (&v)[8], uint32_t cm, int rs, int r_end) {             // zero what is not a real row / column of the operand
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = masked(v[e], (rs + 8 * h + e < r_end ? 0xFFFFFFFFu : 0u) & cm);
    };
    int buf = 0;
    for (int slab = blockIdx.x; slab * kSlab < Kbig; slab += gridDim.x) {
        const int r_end = min(Kbig, (slab + 1) * kSlab);
#pragma unroll
        for (int d = 0; d < kWgRing; ++d)
            if (slab * kSlab + 16 * d < r_end) fetch(ra[d], rb[d], slab * kSlab + 16 * d, r_end);
        for (int r0 = slab * kSlab; r0 < r_end; r0 += 16 * kWgRing) {
#pragma unroll
            for (int d = 0; d < kWgRing; ++d) {
                const int rs = r0 + 16 * d;
                if (rs >= r_end) break;                              // (uniform over the workgroup: the barrier below stays convergent)
                Frag af[MTW];
#pragma unroll
                for (int a = 0; a < MTW; ++a) { clean(ra[d][a], amask[a], rs, r_end); af[a] = split8(ra[d][a]); }
#pragma unroll
                for (int b = 0; b < BW; ++b)
                    if (wave + 4 * b < NT) {                         // wave-uniform
                        clean(rb[d][b], bmask[b], rs, r_end);
                        const Frag f = split8(rb[d][b]);
                        s_b[buf][wave + 4 * b][0][lane] = f.hi; s_b[buf][wave + 4 * b][1][lane] = f.mid; s_b[buf][wave + 4 * b][2][lane] = f.lo;
                    }
                if (rs + 16 * kWgRing < r_end) fetch(ra[d], rb[d], rs + 16 * kWgRing, r_end);      // refill this stage
                __syncthreads();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const Frag bf{s_b[buf][nt][0][lane], s_b[buf][nt][1][lane], s_b[buf][nt][2][lane]};
#pragma unroll
                    for (int a = 0; a < MTW; ++a)
                        if (wave + 4 * a < MT) acc[a][nt] = mfma6(af[a], bf, acc[a][nt]);
                }
                buf ^= 1;     // (the image of step s is rewritten at step s + 2, behind the barrier of step s + 1 every reader of step s has passed)
            }
        }
    }
#pragma unroll
    for (int a = 0; a < MTW; ++a) {
        const int mt = wave + 4 * a;
        if (mt >= MT) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = 32 * mt + acc_row(r, h), cc = 32 * nt + i;
                if (mm < M && cc < N && acc[a][nt][r] != 0.f) unsafeAtomicAdd(C + (size_t)mm * ldc + cc, acc[a][nt][r]);
            }
    }
}

template <int NT>
__global__ void __launch_bounds__(256) wgrad_solo_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int N, int Kbig) {
    constexpr int kWgRing = wg_ring(8 * (1 + NT));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const bool aok = i < M;
    const int am = aok ? i : 0;
    int bn[NT];
    bool bok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { const int n = 32 * nt + i; bok[nt] = n < N; bn[nt] = bok[nt] ? n : 0; }
    const uint32_t amask = aok ? 0xFFFFFFFFu : 0u;
    uint32_t bmask[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bmask[nt] = bok[nt] ? 0xFFFFFFFFu : 0u;
    float ra[kWgRing][8], rb[kWgRing][NT][8];
    auto fetch = [&](float (&da)[8], float (&db)[NT][8], int r0, int r_end) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = r0 + 8 * h + e;
            const bool in = r < r_end;
            const size_t rr = in ? r : r_end - 1;
            da[e] = A[rr * lda + am];                                 // raw (masked where used, see wgrad_shared_kernel)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) db[nt][e] = B[rr * ldb + bn[nt]];
        }
    };
    for (int slab = blockIdx.x; slab * kSlab < Kbig; slab += gridDim.x) {
        const int r_end = min(Kbig, (slab + 1) * kSlab);
        const int rw = slab * kSlab + 16 * wave;                     // this wave's steps: rw, rw + 64, ...
#pragma unroll
        for (int d = 0; d < kWgRing; ++d)
            if (rw + 64 * d < r_end) fetch(ra[d], rb[d], rw + 64 * d, r_end);
        for (int r0 = rw; r0 < r_end; r0 += 64 * kWgRing) {
#pragma unroll
            for (int d = 0; d < kWgRing; ++d) {
                const int rs = r0 + 64 * d;
                if (rs >= r_end) break;
                auto clean = [&](float (&v)[8], uint32_t cm) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = masked(v[e], (rs + 8 * h + e < r_end ? 0xFFFFFFFFu : 0u) & cm);
                };
                clean(ra[d], amask);
                const Frag af = split8(ra[d]);
                Frag bf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { clean(rb[d][nt], bmask[nt]); bf[nt] = split8(rb[d][nt]); }
                if (rs + 64 * kWgRing < r_end) fetch(ra[d], rb[d], rs + 64 * kWgRing, r_end);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma6(af, bf[nt], acc[nt]);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = acc_row(r, h), cc = 32 * nt + i;
            if (mm < M && cc < N && acc[nt][r] != 0.f) unsafeAtomicAdd(C + (size_t)mm * ldc + cc, acc[nt][r]);
        }
}

__global__ void scale_kernel(float* __restrict__ C, int ldc, int M, int N, float beta) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N) return;
    float* p = C + (size_t)(idx / N) * ldc + idx % N;
    *p = beta == 0.f ? 0.f : *p * beta;
}

// any shape: one thread per output element (fp32 FMAs)
__global__ void plain_gemm_kernel(int transA, int transB, int M, int N, int K, const float* __restrict__ A, int lda,
                                  const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc, float beta) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        s = fmaf(transA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k], transB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n], s);
    float* p = C + (size_t)m * ldc + n;
    *p = beta == 0.f ? s : s + beta * *p;
}

// y += bias[column], ReLU (act 1): the epilogue as its own pass, for the products that do not take the tall path
__global__ void bias_act_tail_kernel(float* __restrict__ y, int ldy, const float* __restrict__ bias, int64_t n, int C, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * C) return;
    const int64_t r = i / C;
    const int c = (int)(i % C);
    const float v = y[r * ldy + c] + (bias ? bias[c] : 0.f);
    y[r * ldy + c] = act == 1 ? fmaxf(v, 0.f) : v;
}

// fallback of sherf_bwd_gemm_dgrad_fused: y = 0 where mask <= 0; colsum += column sums (256 rows per workgroup, a thread per column round-robin)
__global__ void __launch_bounds__(256) mask_colsum_tail_kernel(float* __restrict__ y, int ldy, const float* __restrict__ mask, int ldm, int64_t n, int C,
                                                               float* __restrict__ colsum) {
    const int64_t r0 = (int64_t)blockIdx.x * 256, r1 = r0 + 256 < n ? r0 + 256 : n;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            float v = y[r * ldy + c];
            if (mask && !(mask[r * ldm + c] > 0.f)) { v = 0.f; y[r * ldy + c] = 0.f; }
            a += v;
        }
        if (colsum) unsafeAtomicAdd(colsum + c, a);
    }
}

// fallback of sherf_bwd_gemm_bias_act_add: y += addend
__global__ void add2d_tail_kernel(float* __restrict__ y, int ldy, const float* __restrict__ a, int lda, int64_t n, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * C) return;
    const int64_t r = i / C;
    const int c = (int)(i % C);
    y[r * ldy + c] += a[r * lda + c];
}

}  // namespace

static int g_last_path = -1;          // which kernel the last sherf_bwd_gemm call took: 1 = tall MFMA (general), 3 = tall MFMA (streaming), 2 = weight-gradient MFMA (round 2's kernel), 4 = weight-gradient MFMA (shared-B / solo kernels), 5 = streaming tall MFMA with the fused data-gradient store, 0 = plain
extern "C" int sherf_bwd_gemm_last_path() { return g_last_path; }

static int gemm_impl(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, float beta, const float* bias, int act, sherf_stream_t stream);

extern "C" int sherf_bwd_gemm(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                              float* C, int ldc, float beta, sherf_stream_t stream) {
    return gemm_impl(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, beta, nullptr, 0, stream);
}

extern "C" int sherf_bwd_gemm_bias_act(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                       float* C, int ldc, float beta, const float* bias, int act, sherf_stream_t stream) {
    SHERF_CHECK_ARG(act == 0 || act == 1);
    return gemm_impl(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, beta, bias, act, stream);
}

// C[M,N] = A[M,K] . B[K,N] (+ r1_s[row] r1_w[column]) masked by `mask` (> 0 keeps), colsum[column] += the column sums of the result: the data gradient of a
// layer behind a ReLU in ONE kernel when the shape takes the streaming kernel (N, K = 128: every hidden layer of the decoder); otherwise the same result
// from the separate kernels (product, K = 1 product with beta 1, mask + column sums).  r1_s / r1_w, mask, colsum: each optional (null).
extern "C" int sherf_bwd_gemm_dgrad_fused(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                          const float* r1_s, int r1_lds, const float* r1_w, const float* mask, int ldm, float* colsum,
                                          sherf_stream_t stream) {
    SHERF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= N && ldc >= N && (!r1_s == !r1_w) && (!r1_s || r1_lds > 0) && (!mask || ldm >= N));
    hipStream_t st = as_stream(stream);
    const int nkb = (K + 15) / 16, NT = (N + 31) / 32;
    if (NT == 4 && nkb == 8 && lda >= 128 && lda % 4 == 0 && (reinterpret_cast<size_t>(A) & 15) == 0 && !(sherf_experiment() & 64)) {
        const int tiles = (M + 31) / 32, grid = min((tiles + kTallWaves - 1) / kTallWaves, n_cus());
        const size_t smem = (size_t)nkb * NT * 3072;
        SHERF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tall_stream_kernel<4, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((tall_stream_kernel<4, 8, true>), dim3(grid), dim3(64 * kTallWaves), smem, st, A, lda, B, ldb, 0, C, ldc, M, N, K, nullptr, 0,
                           DgradFuse{r1_s, r1_lds, r1_w, mask, ldm, colsum, nullptr, 0});
        g_last_path = 5;
        SHERF_LAUNCH_CHECK();
    }
    SHERF_RUN(gemm_impl(0, 0, M, N, K, A, lda, B, ldb, C, ldc, 0.f, nullptr, 0, stream));
    if (r1_s) SHERF_RUN(gemm_impl(0, 0, M, N, 1, r1_s, r1_lds, r1_w, N, C, ldc, 1.f, nullptr, 0, stream));
    if (mask || colsum) hipLaunchKernelGGL(mask_colsum_tail_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, C, ldc, mask, ldm, (int64_t)M, N, colsum);
    SHERF_LAUNCH_CHECK();
}

// C[M,N] = act(A[M,K] . op(B) + bias) + addend[M,N]: a Linear whose output joins a residual stream (the transformer's to_out and net.3), the residual
// added in the product's store on the streaming kernel's shapes (N <= 32, K = 32 / 48); otherwise the product followed by an add pass.
extern "C" int sherf_bwd_gemm_bias_act_add(int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                           const float* bias, int act, const float* addend, int ld_add, sherf_stream_t stream) {
    SHERF_CHECK_ARG(A && B && C && addend && M > 0 && N > 0 && K > 0 && lda >= K && ldc >= N && ld_add >= N && (act == 0 || act == 1) && addend != C);
    hipStream_t st = as_stream(stream);
    const int nkb = (K + 15) / 16, NT = (N + 31) / 32;
    if (NT == 1 && (nkb == 2 || nkb == 3) && lda >= 16 * nkb && lda % 4 == 0 && (reinterpret_cast<size_t>(A) & 15) == 0 && !(sherf_experiment() & 64)) {
        const int tiles = (M + 31) / 32, grid = min((tiles + kTallWaves - 1) / kTallWaves, n_cus());
        const size_t smem = (size_t)nkb * NT * 3072 < 1024 ? 1024 : (size_t)nkb * NT * 3072;
        const DgradFuse fz{nullptr, 0, nullptr, nullptr, 0, nullptr, addend, ld_add};
        if (nkb == 2) hipLaunchKernelGGL((tall_stream_kernel<1, 2, true>), dim3(grid), dim3(64 * kTallWaves), smem, st, A, lda, B, ldb, transB, C, ldc, M, N, K, bias, act, fz);
        else hipLaunchKernelGGL((tall_stream_kernel<1, 3, true>), dim3(grid), dim3(64 * kTallWaves), smem, st, A, lda, B, ldb, transB, C, ldc, M, N, K, bias, act, fz);
        g_last_path = 5;
        SHERF_LAUNCH_CHECK();
    }
    SHERF_RUN(gemm_impl(0, transB, M, N, K, A, lda, B, ldb, C, ldc, 0.f, bias, act, stream));
    hipLaunchKernelGGL(add2d_tail_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, st, C, ldc, addend, ld_add, (int64_t)M, N);
    SHERF_LAUNCH_CHECK();
}

static int gemm_impl(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, float beta, const float* bias, int act, sherf_stream_t stream) {
    SHERF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N);
    hipStream_t st = as_stream(stream);
    const int NT = (N + 31) / 32, nkb = (K + 15) / 16;
    // tall product: the small matrix B lives in LDS as MFMA fragments (3072 bytes per (K-block, 32-column tile): hi + lo + ...).  When
    // the whole of B does not fit (the skip layer's data gradient dx = dy[n,128] . W[128,199]: 8 K-blocks x 7 tiles = 168 KiB), the
    // columns are cut into slices that do and the kernel is launched once per slice (offset B and C) -- every decoder shape of the
    // backward stays on the MFMA path (tests/test_backward_dense.py: test_decoder_shapes_stay_on_mfma).
    const int nt_fit = min(8, (int)((156 * 1024) / ((size_t)nkb * 3072)));
    if (!transA && nt_fit >= 1 && NT <= 4 * nt_fit) {
        const int tiles = (M + 31) / 32, grid = min((tiles + kTallWaves - 1) / kTallWaves, n_cus());
        bool streamed = false;
        for (int n0 = 0; n0 < N; n0 += 32 * nt_fit) {
            const int Ns = min(N - n0, 32 * nt_fit), NTs = (Ns + 31) / 32;
            const size_t smem = (size_t)nkb * NTs * 3072;
            const float* Bs = transB ? B + (size_t)n0 * ldb : B + n0;
            float* Cs = C + n0;
            // the streaming kernel for the shapes of the path (every (column tiles, K-blocks) pair the decoder / transformer backward produces)
            if (beta == 0.f && lda >= 16 * nkb && lda % 4 == 0 && (reinterpret_cast<size_t>(A) & 15) == 0 && !(sherf_experiment() & 64)) {      // (SHERF_EXPERIMENT bit 6: the general kernel, A/B runs)
                bool hit = true;
#define SHERF_STREAM(n, k) case (n) * 16 + (k): \
                if (smem > 64 * 1024) SHERF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tall_stream_kernel<n, k>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
                hipLaunchKernelGGL((tall_stream_kernel<n, k>), dim3(grid), dim3(64 * kTallWaves), smem, st, A, lda, Bs, ldb, transB, Cs, ldc, M, Ns, K, bias ? bias + n0 : nullptr, act, DgradFuse{}); break
                switch (NTs * 16 + nkb) {
                    SHERF_STREAM(4, 8); SHERF_STREAM(3, 8); SHERF_STREAM(6, 8); SHERF_STREAM(1, 8); SHERF_STREAM(5, 2); SHERF_STREAM(1, 3);
                    SHERF_STREAM(1, 2); SHERF_STREAM(2, 2); SHERF_STREAM(1, 9); SHERF_STREAM(6, 4); SHERF_STREAM(2, 8);
                    SHERF_STREAM(4, 5); SHERF_STREAM(4, 13); SHERF_STREAM(2, 12);          // K = 71, 199, 187 (rows padded to whole blocks by the caller)
                    default: hit = false;
                }
#undef SHERF_STREAM
                if (hit) { streamed = true; continue; }
            }
#define SHERF_TALL(n) case n: \
            if (smem > 64 * 1024) SHERF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tall_gemm_kernel<n>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
            hipLaunchKernelGGL((tall_gemm_kernel<n>), dim3(grid), dim3(64 * kTallWaves), smem, st, A, lda, Bs, ldb, transB, Cs, ldc, M, Ns, K, beta, bias ? bias + n0 : nullptr, act); break
            switch (NTs) { SHERF_TALL(1); SHERF_TALL(2); SHERF_TALL(3); SHERF_TALL(4); SHERF_TALL(5); SHERF_TALL(6); SHERF_TALL(7); SHERF_TALL(8); }
#undef SHERF_TALL
        }
        g_last_path = streamed ? 3 : 1;
        SHERF_LAUNCH_CHECK();
    }
    const int MT = (M + 31) / 32;
    if (transA && !transB && NT <= 8 && MT <= 8 && ((MT + 3) / 4) * NT <= 8) {
        if (beta != 1.f) hipLaunchKernelGGL(scale_kernel, dim3(cdiv((int64_t)M * N, 256)), dim3(256), 0, st, C, ldc, M, N, beta);
        const int slabs = (K + kSlab - 1) / kSlab, grid = min(slabs, 2 * n_cus());
        // round 5: B split once per workgroup / the waves on different steps, for the (row tiles, column tiles) pairs of the path (SHERF_EXPERIMENT bit 7: the old kernel)
        if (!(sherf_experiment() & 128)) {
            bool hit = true;
#define SHERF_WS(m, n) case (m) * 16 + (n): hipLaunchKernelGGL((wgrad_shared_kernel<m, n>), dim3(grid), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K); break
#define SHERF_WO(n) case 16 + (n): hipLaunchKernelGGL((wgrad_solo_kernel<n>), dim3(grid), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K); break
            switch (MT * 16 + NT) {
                SHERF_WS(4, 4); SHERF_WS(4, 3); SHERF_WS(4, 7); SHERF_WS(2, 6); SHERF_WS(5, 1);
                SHERF_WO(1); SHERF_WO(2); SHERF_WO(4);
                default: hit = false;
            }
#undef SHERF_WS
#undef SHERF_WO
            if (hit) {
                g_last_path = 4;
                if (bias || act) hipLaunchKernelGGL(bias_act_tail_kernel, dim3(cdiv((int64_t)M * N, 256)), dim3(256), 0, st, C, ldc, bias, (int64_t)M, N, act);
                SHERF_LAUNCH_CHECK();
            }
        }
#define SHERF_WG(n, w) hipLaunchKernelGGL((wgrad_gemm_kernel<n, w>), dim3(grid), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K)
#define SHERF_WGN(n) case n: if (MT <= 4) SHERF_WG(n, 1); else SHERF_WG(n, 2); break
        switch (NT) {
            SHERF_WGN(1); SHERF_WGN(2); SHERF_WGN(3); SHERF_WGN(4);
            case 5: SHERF_WG(5, 1); break; case 6: SHERF_WG(6, 1); break; case 7: SHERF_WG(7, 1); break; case 8: SHERF_WG(8, 1); break;
        }
#undef SHERF_WGN
#undef SHERF_WG
        g_last_path = 2;
        if (bias || act) hipLaunchKernelGGL(bias_act_tail_kernel, dim3(cdiv((int64_t)M * N, 256)), dim3(256), 0, st, C, ldc, bias, (int64_t)M, N, act);
        SHERF_LAUNCH_CHECK();
    }
    g_last_path = 0;
    hipLaunchKernelGGL(plain_gemm_kernel, dim3(cdiv((int64_t)M * N, 256)), dim3(256), 0, st, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, beta);
    if (bias || act) hipLaunchKernelGGL(bias_act_tail_kernel, dim3(cdiv((int64_t)M * N, 256)), dim3(256), 0, st, C, ldc, bias, (int64_t)M, N, act);
    SHERF_LAUNCH_CHECK();
}
