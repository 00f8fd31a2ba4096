// Fused per-sample network (gfx950 MFMA): rows a13 + a14 of SURVEY.md section 8.
//
//   slot-2 token += W_b . PE5(rgb)[:32]                         (rest of conv1d_reprojection, renderer.py:423-424)
//   3-token pre-norm transformer, dim 32, 3 heads x 16            (renderer.py:949-993)
//   PE6(x_c), PE4(v_c)                                            (renderer.py:875-916)
//   NeRFDecoder 8x128 with skip, sigma / feature / view / rgb     (triplane.py:285-316)
//
// Formulation: D^T = W . X^T -- the weight matrix is the MFMA A operand (32 output features per tile) and 32
// samples are the B operand columns, using v_mfma_f32_32x32x16_bf16.  A wave owns 32 samples for the whole
// network; a layer's fp32 accumulator tile (lane = sample, 16 regs = 16 of the tile's 32 features) is
// converted in registers into two B operand K-blocks of the next layer: the K permutation this implies,
//     k-slot (kb, h, e)  <->  feature 16*kb + (e&3) + 8*(e>>2) + 4*h,
// is baked into the packed weight stream (sherf_amd/mlp_pack.py), so activations never leave the register
// file and never cross lanes (except the 3x3 attention dot products and LayerNorm sums: one lane^32 exchange).
// Weights stream L2 -> LDS one output tile ("chunk") at a time in MFMA fragment order (ds_read_b128,
// lane-linear, conflict free), double buffered, one barrier per chunk, shared by the 8 waves of a workgroup.
//
// prec 0: bf16 operands, fp32 accumulate.   prec 1 ("bf16x3"): operands split hi+lo, three MFMAs per product
// (lo*hi + hi*lo + hi*hi), ~2^-16 relative error -- the mode that meets the 1e-3 parity tolerance.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int N_CHUNKS = 49;
constexpr int MAX_NKB = 13;
constexpr int BIAS_LN = N_CHUNKS;   // index of the first LayerNorm table in wbias ([idx][2][16] floats)

// K-blocks (16 inputs each) per chunk; a chunk is one 32-row output tile of one layer.
__host__ __device__ constexpr int chunk_nkb(int c) {
    return c == 0 ? 2        // rgb PE -> slot-2 token
         : c <= 5 ? 2        // to_qkv tiles: [q0|q1] [q2|pad] [k0|k1] [k2|v0] [v1|v2]
         : c == 6 ? 3        // to_out (48 -> 32)
         : c <= 8 ? 2        // FF 32->32, 32->32
         : c <= 12 ? 5       // pts_linears.0 : PE6 (3 kb) + z0 (2 kb)
         : c <= 28 ? 8       // pts_linears.1-4
         : c <= 32 ? 13      // pts_linears.5 : PE6 + z0 + h
         : c <= 40 ? 8       // pts_linears.6-7
         : c <= 45 ? 8       // feature_linear (4 tiles) + alpha_linear (1 tile, row 0)
         : c <= 47 ? 12      // views_linear : feature (8) + PE4 (2) + z1 (2)
         : 4;                // rgb_linear (rows 0..2)
}
__host__ __device__ constexpr int chunk_off_kb(int c) {   // stream offset in KiB; every chunk stores hi then lo
    int o = 0;
    for (int i = 0; i < c; ++i) o += 2 * chunk_nkb(i);
    return o;
}

template <int PREC> struct BFrag;
template <> struct BFrag<0> { uint4 hi; };
template <> struct BFrag<1> { uint4 hi, lo; };

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_rt(float a) { return (float)((__bf16)a); }

// hi/lo split of a pair for the bf16x3 mode: hi = the top 16 bits (truncation; one v_perm for the pair), lo = x - hi is
// exact in fp32 and then rounded to bf16, so hi + lo carries ~17 bits whichever way hi was rounded.
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const uint32_t ua = __builtin_bit_cast(uint32_t, a), ub = __builtin_bit_cast(uint32_t, b);
    hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);          // [a.hi16 | b.hi16 << 16]
    lo = pack2(a - __builtin_bit_cast(float, ua & 0xFFFF0000u), b - __builtin_bit_cast(float, ub & 0xFFFF0000u));
}

template <int PREC>
__device__ __forceinline__ BFrag<PREC> make_frag(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    BFrag<PREC> f;
    if constexpr (PREC == 1) {
        split2(v0, v1, f.hi.x, f.lo.x); split2(v2, v3, f.hi.y, f.lo.y);
        split2(v4, v5, f.hi.z, f.lo.z); split2(v6, v7, f.hi.w, f.lo.w);
    } else {
        f.hi = make_uint4(pack2(v0, v1), pack2(v2, v3), pack2(v4, v5), pack2(v6, v7));
    }
    return f;
}

// ReLU as an integer max: one instruction, no NaN-canonicalising v_max pair (inputs are finite)
__device__ __forceinline__ float relu(float x) {
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}

// one fp32 accumulator tile -> the two K-blocks it becomes as an input of the next layer
template <int PREC>
__device__ __forceinline__ void split_tile(const f32x16& a, BFrag<PREC>& k0, BFrag<PREC>& k1) {
    k0 = make_frag<PREC>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
    k1 = make_frag<PREC>(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
}

__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel shape: NW waves per workgroup, NTL column tiles (32 samples each) per wave.
//   <NW=8, NTL=1>: two waves per SIMD (<= 256 VGPRs), each wave one tile.
//   <NW=4, NTL=2>: one wave per SIMD (<= 512 VGPRs), each wave two tiles: every weight fragment read from LDS feeds two
//                  MFMA chains, half the barriers and LDS traffic per sample.
// The weight stream is walked in STEPS: the small transformer chunks 0..8 are replayed once per tile (keeps the
// attention state of only one tile live), then the decoder chunks 9..48 run once for all tiles of the wave.
// ---------------------------------------------------------------------------------------------------------------
// PHASE 0: the fused kernel (transformer + decoder).  PHASE 1 / 2 (shape '8x1split', experimental): the VALU-bound
// transformer prologue and the MFMA-bound decoder as two launches -- 1 streams only chunks 0..8 (tiny ring slots => several
// workgroups per CU, its waves no longer phase-locked to MFMA-bound ones) and leaves z_0, z_1 as ready-made bf16 hi/lo
// B-fragments in the first 8 KiB of the tile's `tokens` block; 2 streams chunks 9..48 and starts from those fragments.
// PHASE 3 = PHASE 2 with the 128-input layers walked TWO output tiles per step (one DMA of both chunks into a 32 KiB slot, two
// independent accumulator chains, half the workgroup barriers of the trunk): 26 steps instead of 40.
// PHASE -1 (shape '8x1persist', experimental) = the fused kernel as PERSISTENT workgroups: one per CU, each walking tile groups
// blockIdx.x, blockIdx.x + gridDim.x, ...; the bias tables stay in LDS and the weight ring never drains -- the last three steps
// of a group already stream steps 0..2 of the next one into the slots they free (slot(s) = s % 3 holds for every group) -- so
// a group pays neither the launch gap nor the exposed first-chunk latency of a fresh workgroup.
template <int NTL, int PHASE> __host__ __device__ constexpr int n_steps() {
    return PHASE == 1 ? 9 * NTL : PHASE == 2 ? N_CHUNKS - 9 : PHASE == 3 ? 26 : 9 * NTL + (N_CHUNKS - 9);
}
template <int NTL, int PHASE> __host__ __device__ constexpr int step_chunk(int s) {       // first chunk of a step
    if (PHASE == 3)
        return s < 4 ? 9 + s : s < 12 ? 13 + 2 * (s - 4) : s < 16 ? 29 + (s - 12) : s < 20 ? 33 + 2 * (s - 16) : s < 22 ? 41 + 2 * (s - 20)
             : s == 22 ? 45 : s < 25 ? 46 + (s - 23) : 48;
    return PHASE == 2 ? s + 9 : s < 9 * NTL ? s % 9 : s - 9 * (NTL - 1);
}
template <int PHASE> __host__ __device__ constexpr int step_count(int s) {                // chunks streamed in that step
    return PHASE == 3 && ((s >= 4 && s < 12) || (s >= 16 && s < 22)) ? 2 : 1;
}

template <int PREC, int NW, int NTL, int PHASE = 0> struct Ctx {
    const char* ws;          // packed weight stream (global)
    const float* wbias;      // [N_CHUNKS + 4][2][16] D-layout bias / LayerNorm tables (LDS copy)
    char* lds;               // NSLOT ring slots
    int lane, h, dbg, wave, pending;
#if SHERF_MLP_TRACE
    uint32_t* trace;         // this wave's [64][4] stamps in LDS
#endif
    bool more;               // PHASE -1: this workgroup has another tile group after the current one (uniform)
    static constexpr int SLOT = (PHASE == 1 ? 3 : PHASE == 3 ? 16 : MAX_NKB) * 1024 * (PREC + 1);     // chunks 0..8: <= 3 K-blocks; pairs: 2 x 8
    // <NW=4, NTL=1> (shape '4x1'): half-size workgroups, TWO co-resident per CU (one wave of each per SIMD) that are not coupled by
    // each other's barriers; their rings must fit the 160 KiB LDS together -> two slots, one step of prefetch distance.
    static constexpr int NSLOT = (NW == 4 && NTL == 1) ? 2 : 3;
    __device__ __forceinline__ const char* slot(int step) const { return lds + (step % NSLOT) * SLOT + lane * 16; }
};

// Weight stream L2 -> LDS by LDS-DMA (global_load_lds, 1 KiB per wave-instruction, no VGPR round trip), three ring
// slots, two steps of prefetch distance.  Completion: a wave waits (counted vmcnt) until only the pieces of the most
// recently issued step are still in flight, then the workgroup barrier makes every wave's pieces visible.
typedef __attribute__((address_space(3))) void* lptr_t;

template <int PREC, int NW, int NTL, int PHASE>
__device__ __forceinline__ int dma_issue(Ctx<PREC, NW, NTL, PHASE>& cx, int step) {
    if (step >= n_steps<NTL, PHASE>() || (cx.dbg & 32)) return 0;
    const int c = step_chunk<NTL, PHASE>(step);
    const int pieces = chunk_nkb(c) * (PREC + 1) * step_count<PHASE>(step);       // a paired step streams two equal, adjacent chunks
    const char* src = cx.ws + (size_t)chunk_off_kb(c) * 1024 + cx.lane * 16;
    char* dst = cx.lds + (step % Ctx<PREC, NW, NTL, PHASE>::NSLOT) * Ctx<PREC, NW, NTL, PHASE>::SLOT;
    int n = 0;
    constexpr int kMaxPieces = (PHASE == 3 ? 16 : MAX_NKB) * (PREC + 1);
#pragma unroll
    for (int i = 0; i < (kMaxPieces + NW - 1) / NW; ++i) {
        const int p = cx.wave + i * NW;                        // wave-uniform
        if (p < pieces) {
            // Inline asm on purpose: hipcc drains an LDS-DMA it knows about (vmcnt(0)) before every ds_read that might alias
            // it, which would serialise the ring.  M0 = LDS byte address of the piece (saved/restored: compiler-reserved).
            const char* g = src + p * 1024;
            uint32_t l = (uint32_t)(size_t)(lptr_t)(dst + p * 1024);
            if constexpr (PHASE != 0) l = __builtin_amdgcn_readfirstlane(l);   // wave-uniform by construction; the split kernels' control
                                                                               // flow hides that from the compiler ("s" constraint)
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(l) : "memory");
            ++n;
        }
    }
    return n;
}

__device__ __forceinline__ void wait_vm(int n) {              // s_waitcnt vmcnt(n) needs an immediate
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// end of step s: step s+1 must have landed (everything but the newest issue), then step s's slot is recycled for s+3
// SHERF_MLP_TRACE (profiling builds only, tools/mlp_trace.py): every wave stamps s_memtime at the end of its MFMA stream, after the
// weight-DMA wait and after the workgroup barrier of every step into LDS; selected workgroups copy the stamps out at the end.
#ifndef SHERF_MLP_TRACE
#define SHERF_MLP_TRACE 0
#endif
#if SHERF_MLP_TRACE
__device__ uint32_t* g_mlp_trace = nullptr;           // [slot][wave 0..7][64 steps][4] u32
__device__ int g_mlp_trace_every = 0;
#define SHERF_TRACE_STAMP(cx, step, k) do { if ((cx).lane == 0) (cx).trace[(step) * 4 + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SHERF_TRACE_STAMP(cx, step, k) do { } while (0)
#endif

template <int PREC, int NW, int NTL, int PHASE>
__device__ __forceinline__ void advance(Ctx<PREC, NW, NTL, PHASE>& cx, int step) {
    constexpr int NSLOT = Ctx<PREC, NW, NTL, PHASE>::NSLOT;
    SHERF_TRACE_STAMP(cx, step, 0);
    wait_vm(NSLOT == 2 ? 0 : cx.pending);    // two slots: the newest issue IS step s+1
    SHERF_TRACE_STAMP(cx, step, 1);
    if (!(cx.dbg & 64)) wg_barrier();
    SHERF_TRACE_STAMP(cx, step, 2);
    if constexpr (PHASE == -1) {             // past the end of this group: the freed slot takes step (step + 3) % 3 of the next group
        if (step + NSLOT >= n_steps<NTL, PHASE>()) { cx.pending = cx.more ? dma_issue(cx, (step + NSLOT) % NSLOT) : 0; return; }
    }
    cx.pending = dma_issue(cx, step + NSLOT);
}

template <class C>
__device__ __forceinline__ f32x16 bias_tile(const C& cx, int idx) {
    const float4* p = reinterpret_cast<const float4*>(cx.wbias + (idx * 2 + cx.h) * 16);
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    f32x16 r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    return r;
}

// SHERF_MLP_FASTMATH: hardware transcendentals (v_sin/v_cos/v_exp/v_rsq/v_rcp, ~1e-6 absolute) instead of the
// correctly rounded libm sequences in the VALU-bound transformer / positional-encoding prologue (sincosf alone is ~130
// instructions + branches, six times per tile).
#ifndef SHERF_MLP_FASTMATH
#define SHERF_MLP_FASTMATH 1
#endif
#if SHERF_MLP_FASTMATH
__device__ __forceinline__ void sincos_(float x, float* s, float* c) { *s = __sinf(x); *c = __cosf(x); }
__device__ __forceinline__ float exp_(float x) { return __expf(x); }
__device__ __forceinline__ float rcp_(float x) { return __frcp_rn(x); }
__device__ __forceinline__ float rsqrt_(float x) { return __frsqrt_rn(x); }
#else
__device__ __forceinline__ void sincos_(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ float exp_(float x) { return expf(x); }
__device__ __forceinline__ float rcp_(float x) { return 1.0f / x; }
__device__ __forceinline__ float rsqrt_(float x) { return 1.0f / sqrtf(x); }
#endif
// SHERF_MLP_FAST_ERF (off; to be measured and parity-checked on hardware): Abramowitz-Stegun 7.1.26 for the exact-GELU erf,
// |error| <= 1.5e-7, ~14 VALU instead of libm's ~38 -- the 32 erf evaluations per lane are the largest VALU block left
// in the prologue (~1.2 K of 5.3 K instructions per tile).  tests/test_mlp_pack.py checks the formula in float32.
#ifndef SHERF_MLP_FAST_ERF
#define SHERF_MLP_FAST_ERF 0
#endif
__device__ __forceinline__ float erf_(float x) {
#if SHERF_MLP_FAST_ERF
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    return copysignf(1.0f - poly * __expf(-ax * ax), x);
#else
    return erff(x);
#endif
}
#ifndef SHERF_MLP_P1_WAVES
#define SHERF_MLP_P1_WAVES 2     // waves per SIMD the transformer-only launch is compiled for (VGPR cap 512 / this)
#endif
#ifndef SHERF_MLP_INTERLEAVE
#define SHERF_MLP_INTERLEAVE 0
#endif
// SHERF_MLP_WAVE_PRIO (off; to be measured): the two waves a SIMD hosts leave every workgroup barrier together, interleave their
// MFMA chains and then run their VALU epilogues at the same time with the MFMA pipe idle.  Giving the first half of the waves
// (one per SIMD) issue priority lets that wave finish its chain first and do its epilogue under the other wave's MFMAs.
#ifndef SHERF_MLP_WAVE_PRIO
#define SHERF_MLP_WAVE_PRIO 0
#endif
// acc[col] += W_step[kb0 .. kb0+NK) . B[col]: one segment of a chunk's K range, NCOL column sets sharing the A fragments
// SHERF_MLP_SPLITK (experiment): the 8-K-block segments accumulate even / odd K-blocks in two independent chains (summed at the
// end), so that a wave's MFMA stream is not one 24-deep dependent chain.
#ifndef SHERF_MLP_SPLITK
#define SHERF_MLP_SPLITK 0
#endif
template <int PREC, int NK, int NCOL, int IL>
__device__ __forceinline__ void mma_seg(const char* s, int kb0, int nkb_total, const BFrag<PREC> (&b)[NCOL][NK], f32x16 (&acc)[NCOL]) {
#if SHERF_MLP_SPLITK
    if constexpr (PREC == 1 && NK == 8 && NCOL == 1) {
        f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NK; kb += 2) {
            const uint4 ah0 = *reinterpret_cast<const uint4*>(s + (kb0 + kb) * 1024);
            const uint4 al0 = *reinterpret_cast<const uint4*>(s + (nkb_total + kb0 + kb) * 1024);
            const uint4 ah1 = *reinterpret_cast<const uint4*>(s + (kb0 + kb + 1) * 1024);
            const uint4 al1 = *reinterpret_cast<const uint4*>(s + (nkb_total + kb0 + kb + 1) * 1024);
            acc[0] = mfma(al0, b[0][kb].hi, acc[0]);
            acc2 = mfma(al1, b[0][kb + 1].hi, acc2);
            acc[0] = mfma(ah0, b[0][kb].lo, acc[0]);
            acc2 = mfma(ah1, b[0][kb + 1].lo, acc2);
            acc[0] = mfma(ah0, b[0][kb].hi, acc[0]);
            acc2 = mfma(ah1, b[0][kb + 1].hi, acc2);
        }
        acc[0] += acc2;
        return;
    }
#endif
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
        const uint4 ah = *reinterpret_cast<const uint4*>(s + (kb0 + kb) * 1024);
        if constexpr (PREC == 1) {
            const uint4 al = *reinterpret_cast<const uint4*>(s + (nkb_total + kb0 + kb) * 1024);
#pragma unroll
            for (int t = 0; t < NCOL; ++t) {
                acc[t] = mfma(al, b[t][kb].hi, acc[t]);
                acc[t] = mfma(ah, b[t][kb].lo, acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < NCOL; ++t) acc[t] = mfma(ah, b[t][kb].hi, acc[t]);
        // Software pipeline across chunks: the previous chunk's epilogue (ReLU + bf16 hi/lo split, ~60 VALU) is independent
        // of this chunk's MFMAs; left alone the compiler sinks all four epilogues of a layer in front of the next layer's
        // first MFMA (240 VALU during which this wave's MFMA pipe idles).  Ask for a few of them after every K-block.
        if constexpr (IL > 0 && PREC == 1 && NK >= 4) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NCOL, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, IL, 0);
        }
    }
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }   // partner lane holds the other 16 features

// LayerNorm over the 32 features of a token (16 here, 16 in lane^32), eps 1e-5 (renderer.py:931)
template <int PREC, class C>
__device__ __forceinline__ void layer_norm(const C& cx, const f32x16& x, int ln_idx, BFrag<PREC>& k0, BFrag<PREC>& k1) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[r];
    s += xhalf(s);
    const float mean = s * (1.0f / 32.0f);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { float d = x[r] - mean; q += d * d; }
    q += xhalf(q);
    const float inv = rsqrt_(q * (1.0f / 32.0f) + 1e-5f);
    const f32x16 g = bias_tile(cx, BIAS_LN + 2 * ln_idx), bt = bias_tile(cx, BIAS_LN + 2 * ln_idx + 1);
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = (x[r] - mean) * inv * g[r] + bt[r];
    split_tile<PREC>(y, k0, k1);
}

// NeRF positional encoding in "natural" K-block order: feature f = 16*kb + 8*h + e of
// [x(3), sin(2^0 x)(3), cos(2^0 x)(3), sin(2^1 x)(3), ...]; entries >= 3 + 6*NF are zero.
template <int PREC, int NF, int NKB>
__device__ __forceinline__ void pe_frags(int h, float x, float y, float z, BFrag<PREC> (&out)[NKB]) {
    float f[NKB * 16];
#pragma unroll
    for (int i = 0; i < NKB * 16; ++i) f[i] = 0.f;
    f[0] = x; f[1] = y; f[2] = z;
    float s[3], c[3];
    sincos_(x, &s[0], &c[0]); sincos_(y, &s[1], &c[1]); sincos_(z, &s[2], &c[2]);
#pragma unroll
    for (int q = 0; q < NF; ++q) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (3 + 6 * q + a < NKB * 16) f[3 + 6 * q + a] = s[a];
            if (6 + 6 * q + a < NKB * 16) f[6 + 6 * q + a] = c[a];
            float s2 = 2.f * s[a] * c[a], c2 = 1.f - 2.f * s[a] * s[a];    // double the angle
            s[a] = s2; c[a] = c2;
        }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = h ? f[16 * kb + 8 + e] : f[16 * kb + e];
        out[kb] = make_frag<PREC>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
}


// VAR: scheduling variant of the same arithmetic (results bit-identical): low byte = SHERF_MLP_INTERLEAVE count, next byte =
// SHERF_MLP_WAVE_PRIO level.  Runtime-selectable through `shape` 5-7 of sherf_nerf_mlp so that sherf_amd.tune can time them on
// the hardware it runs on; the -D macros only move the default.
template <int PREC, int NW, int NTL, int PHASE = 0, int VAR = (SHERF_MLP_INTERLEAVE | (SHERF_MLP_WAVE_PRIO << 8))>      // (bits 16+: PHASE_PRIO)
__global__ void __launch_bounds__(NW * 64, PHASE == 1 ? SHERF_MLP_P1_WAVES : NTL == 1 ? 2 : 1)
nerf_mlp_kernel(const int32_t* __restrict__ counters, const float4* __restrict__ tokens, const float* __restrict__ extras,
                const char* __restrict__ ws, const float* __restrict__ wbias, int64_t capacity, float4* __restrict__ out, int dbg) {
    using CX = Ctx<PREC, NW, NTL, PHASE>;
    constexpr int NT = NW * 64;
    constexpr int IL = VAR & 0xff, PRIO = (VAR >> 8) & 0xff, PHASE_PRIO = (VAR >> 16) & 0xff;
    __shared__ __attribute__((aligned(16))) char lds[CX::NSLOT * CX::SLOT + (N_CHUNKS + 4) * 32 * 4 + (SHERF_MLP_TRACE ? NW * 64 * 16 : 0)];
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t n_tiles = (nv + 31) / 32;
    if ((int64_t)blockIdx.x * NW * NTL >= n_tiles) return;           // whole workgroup beyond the data
    CX cx;
    float* lbias = reinterpret_cast<float*>(lds + CX::NSLOT * CX::SLOT);
    for (int i = threadIdx.x; i < (N_CHUNKS + 4) * 32; i += NT) lbias[i] = wbias[i];   // visible after the prologue barrier
    cx.ws = ws; cx.wbias = lbias; cx.lds = lds;
    cx.lane = threadIdx.x & 63; cx.h = cx.lane >> 5; cx.dbg = dbg;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if SHERF_MLP_TRACE
    cx.trace = reinterpret_cast<uint32_t*>(lds + CX::NSLOT * CX::SLOT + (N_CHUNKS + 4) * 32 * 4) + cx.wave * 256;
    if (cx.lane == 0) { cx.trace[63 * 4] = (uint32_t)__builtin_amdgcn_s_memtime(); cx.trace[63 * 4 + 1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }   // HW_ID
#endif
    if constexpr (PRIO > 0) { if (cx.wave < NW / 2) __builtin_amdgcn_s_setprio(PRIO); }
    int j = cx.lane & 31, h = cx.h;                                  // (not const: PHASE -1 launders them per group)
    int64_t tile[NTL];
    bool live[NTL];
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
        if constexpr (PHASE == -1) tile[u] = ((int64_t)blockIdx.x * NW + cx.wave) * NTL + u;      // scalar: lives across the group loop
        else tile[u] = ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * NTL + u;
        live[u] = tile[u] < n_tiles;
        if (!live[u]) tile[u] = n_tiles - 1;                         // dead tiles still take part in every barrier
    }

    dma_issue(cx, 0);
    const int n1 = dma_issue(cx, 1);
    wait_vm(n1);                              // step 0 (this wave's pieces) landed
    __syncthreads();
    cx.pending = CX::NSLOT == 2 ? n1 : dma_issue(cx, 2);

    [[maybe_unused]] int64_t grp = blockIdx.x;                       // PHASE -1: tile group of this pass
    bool again = false;
    do {                                                             // one pass unless PHASE == -1 (persistent workgroups)
    if constexpr (PHASE == -1) {
        cx.more = grp + gridDim.x < (n_tiles + NW * NTL - 1) / (NW * NTL);
        // keep the per-chunk source addresses from being hoisted out of the group loop (dozens of live 64-bit VGPR pairs)
        asm volatile("" : "+s"(cx.ws));
        asm volatile("" : "+v"(cx.lane));         // same for everything derived from the lane id: one live VGPR, re-derived per group
        j = cx.lane & 31; h = cx.lane >> 5; cx.h = h;
    }
    BFrag<PREC> z0b[NTL][2], z1b[NTL][2];                            // fused tokens z_0, z_1 as K-blocks, per tile
    float xc[NTL][3], vc[NTL][3];
    int step = 0;

    // z_0 / z_1 hand-over between the two launches of the split shape: 8 uint4 per lane per tile, lane-major, written over the
    // first 8 KiB of the tile's own `tokens` block (the wave has read its tokens into registers long before): fragment
    // q = 4 * (0: z_0, 1: z_1) + 2 * kb + (0: hi, 1: lo).
    uint4* const zfrag = reinterpret_cast<uint4*>(const_cast<float4*>(tokens));
    if constexpr (PHASE >= 2) {
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
            const float* ex = extras + tile[u] * 12 * 32 + j;
            xc[u][0] = ex[0]; xc[u][1] = ex[32]; xc[u][2] = ex[64]; vc[u][0] = ex[96]; vc[u][1] = ex[128]; vc[u][2] = ex[160];
            const uint4* zp = zfrag + tile[u] * 768 + cx.lane;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                z0b[u][kb].hi = zp[(2 * kb) * 64];
                z1b[u][kb].hi = zp[(4 + 2 * kb) * 64];
                if constexpr (PREC == 1) { z0b[u][kb].lo = zp[(2 * kb + 1) * 64]; z1b[u][kb].lo = zp[(4 + 2 * kb + 1) * 64]; }
            }
        }
    }
    // ================= transformer, one tile at a time (steps 9u .. 9u+8 replay chunks 0..8) =================
#pragma unroll
    for (int u = 0; u < (PHASE >= 2 ? 0 : NTL); ++u) {
        // ---- inputs: tokens in D layout (quad q = 2i+h -> regs 4i..4i+3), extras ----
        f32x16 tok[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 v = tokens[((tile[u] * 3 + t) * 8 + (2 * i + h)) * 32 + j];
                tok[t][4 * i] = v.x; tok[t][4 * i + 1] = v.y; tok[t][4 * i + 2] = v.z; tok[t][4 * i + 3] = v.w;
            }
        const float* ex = extras + tile[u] * 12 * 32 + j;
        xc[u][0] = ex[0]; xc[u][1] = ex[32]; xc[u][2] = ex[64]; vc[u][0] = ex[96]; vc[u][1] = ex[128]; vc[u][2] = ex[160];

        // ---- chunk 0: slot-2 token += W_b . PE5(rgb)[:32] ----
        {
            BFrag<PREC> b[1][2];
            pe_frags<PREC, 5, 2>(h, ex[192], ex[224], ex[256], b[0]);
            if constexpr (PHASE == -1) {
                if (u == 0 && grp != (int64_t)blockIdx.x) {          // steps 0..2 were streamed by the previous group's last steps
                    wait_vm(0);
                    __syncthreads();
                    cx.pending = 0;
                }
            }
            f32x16 acc[1] = {bias_tile(cx, 0)};
            mma_seg<PREC, 2, 1, IL>(cx.slot(step), 0, 2, b, acc);
            advance(cx, step); ++step;
            tok[2] += acc[0];
        }
        // ---- LN1 + to_qkv (chunks 1..5), attention, to_out (6) ----
        BFrag<PREC> ln[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) layer_norm<PREC>(cx, tok[t], 0, ln[t][0], ln[t][1]);
        float qa[2][16], qb[2][8];
        {
            BFrag<PREC> b2[2][2] = {{ln[0][0], ln[0][1]}, {ln[1][0], ln[1][1]}};
            f32x16 acc[2] = {bias_tile(cx, 1), bias_tile(cx, 1)};
            mma_seg<PREC, 2, 2, IL>(cx.slot(step), 0, 2, b2, acc);          // [q head0 | q head1]
            advance(cx, step); ++step;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) qa[i][r] = acc[i][r];
            f32x16 acc2[2] = {bias_tile(cx, 2), bias_tile(cx, 2)};
            mma_seg<PREC, 2, 2, IL>(cx.slot(step), 0, 2, b2, acc2);         // [q head2 | pad]
            advance(cx, step); ++step;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) qb[i][r] = acc2[i][r];
        }
        float dot[2][3][3];                                     // [query token][head][key token]
        float o[2][3][8];                                       // attention output [query][head][8 of 16 dims]
        float v0[3][8];
        {
            f32x16 acc[3] = {bias_tile(cx, 3), bias_tile(cx, 3), bias_tile(cx, 3)};
            mma_seg<PREC, 2, 3, IL>(cx.slot(step), 0, 2, ln, acc);          // [k head0 | k head1]
            advance(cx, step); ++step;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    float d0 = 0.f, d1 = 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) { d0 += qa[i][r] * acc[t][r]; d1 += qa[i][8 + r] * acc[t][8 + r]; }
                    dot[i][0][t] = d0; dot[i][1][t] = d1;
                }
        }
        {
            f32x16 acc[3] = {bias_tile(cx, 4), bias_tile(cx, 4), bias_tile(cx, 4)};
            mma_seg<PREC, 2, 3, IL>(cx.slot(step), 0, 2, ln, acc);          // [k head2 | v head0]
            advance(cx, step); ++step;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    float d2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) d2 += qb[i][r] * acc[t][r];
                    dot[i][2][t] = d2;
                }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 8; ++r) v0[t][r] = acc[t][8 + r];
        }
        // softmax over the 3 keys, scale 16^-0.5 (renderer.py:956,971-973)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int hd = 0; hd < 3; ++hd) {
                float d[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) d[t] = (dot[i][hd][t] + xhalf(dot[i][hd][t])) * 0.25f;
                float m = fmaxf(d[0], fmaxf(d[1], d[2]));
                float e0 = exp_(d[0] - m), e1 = exp_(d[1] - m), e2 = exp_(d[2] - m);
                float inv = rcp_(e0 + e1 + e2);
                dot[i][hd][0] = e0 * inv; dot[i][hd][1] = e1 * inv; dot[i][hd][2] = e2 * inv;
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) o[i][0][r] = dot[i][0][0] * v0[0][r] + dot[i][0][1] * v0[1][r] + dot[i][0][2] * v0[2][r];
        {
            f32x16 acc[3] = {bias_tile(cx, 5), bias_tile(cx, 5), bias_tile(cx, 5)};
            mma_seg<PREC, 2, 3, IL>(cx.slot(step), 0, 2, ln, acc);          // [v head1 | v head2]
            advance(cx, step); ++step;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    o[i][1][r] = dot[i][1][0] * acc[0][r] + dot[i][1][1] * acc[1][r] + dot[i][1][2] * acc[2][r];
                    o[i][2][r] = dot[i][2][0] * acc[0][8 + r] + dot[i][2][1] * acc[1][8 + r] + dot[i][2][2] * acc[2][8 + r];
                }
        }
        f32x16 y[2];
        {
            BFrag<PREC> ob[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int hd = 0; hd < 3; ++hd)
                    ob[i][hd] = make_frag<PREC>(o[i][hd][0], o[i][hd][1], o[i][hd][2], o[i][hd][3], o[i][hd][4], o[i][hd][5],
                                                o[i][hd][6], o[i][hd][7]);
            f32x16 acc[2] = {bias_tile(cx, 6), bias_tile(cx, 6)};
            mma_seg<PREC, 3, 2, IL>(cx.slot(step), 0, 3, ob, acc);          // to_out + bias
            advance(cx, step); ++step;
            y[0] = acc[0] + tok[0]; y[1] = acc[1] + tok[1];             // residual (renderer.py:925)
        }
        // ---- FF: LN2 -> Linear -> GELU(erf) -> Linear, residual (chunks 7, 8) ----
        {
            BFrag<PREC> l2[2][2];
            layer_norm<PREC>(cx, y[0], 1, l2[0][0], l2[0][1]);
            layer_norm<PREC>(cx, y[1], 1, l2[1][0], l2[1][1]);
            f32x16 acc[2] = {bias_tile(cx, 7), bias_tile(cx, 7)};
            mma_seg<PREC, 2, 2, IL>(cx.slot(step), 0, 2, l2, acc);
            advance(cx, step); ++step;
            BFrag<PREC> gb[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { float a = acc[i][r]; acc[i][r] = 0.5f * a * (1.0f + erf_(a * 0.70710678118654752f)); }
                split_tile<PREC>(acc[i], gb[i][0], gb[i][1]);
            }
            f32x16 acc2[2] = {bias_tile(cx, 8), bias_tile(cx, 8)};
            mma_seg<PREC, 2, 2, IL>(cx.slot(step), 0, 2, gb, acc2);
            advance(cx, step); ++step;
            f32x16 za = acc2[0] + y[0], zb = acc2[1] + y[1];
            split_tile<PREC>(za, z0b[u][0], z0b[u][1]);
            split_tile<PREC>(zb, z1b[u][0], z1b[u][1]);
        }
    }
    if constexpr (PHASE == 1) {                                      // hand z_0, z_1 to the decoder launch and stop
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
            if (!live[u]) continue;
            uint4* zp = zfrag + tile[u] * 768 + cx.lane;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                zp[(2 * kb) * 64] = z0b[u][kb].hi;
                zp[(4 + 2 * kb) * 64] = z1b[u][kb].hi;
                if constexpr (PREC == 1) { zp[(2 * kb + 1) * 64] = z0b[u][kb].lo; zp[(4 + 2 * kb + 1) * 64] = z1b[u][kb].lo; }
            }
        }
        return;
    }

    // ================= NeRF decoder: all NTL tiles of the wave share every weight fragment =================
    // PHASE_PRIO (shape '4x1phase'): from here on the wave is MFMA-bound -- it gets issue priority over the co-resident workgroup's wave on
    // this SIMD whenever that one is still in its VALU-bound transformer / encoding prologue (a new wave starts at priority 0)
    if constexpr (PHASE_PRIO > 0) __builtin_amdgcn_s_setprio(PHASE_PRIO);
    BFrag<PREC> ha[NTL][8], hb[NTL][8], pe[NTL][3];
#pragma unroll
    for (int u = 0; u < NTL; ++u) pe_frags<PREC, 6, 3>(h, xc[u][0], xc[u][1], xc[u][2], pe[u]);
    // one output tile of a trunk layer: bias, MFMAs over the listed input segments, ReLU, split into the next layer's K-blocks
#define SHERF_TRUNK_TILE(CHUNK, OUT, T, ...)                                                          \
    {                                                                                                 \
        f32x16 acc[NTL];                                                                              \
        _Pragma("unroll") for (int u = 0; u < NTL; ++u) acc[u] = bias_tile(cx, CHUNK);                \
        const char* s_ = cx.slot(step);                                                               \
        __VA_ARGS__                                                                                   \
        advance(cx, step); ++step;                                                                    \
        _Pragma("unroll") for (int u = 0; u < NTL; ++u) {                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[u][r] = relu(acc[u][r]);               \
            split_tile<PREC>(acc[u], OUT[u][2 * (T)], OUT[u][2 * (T) + 1]);                          \
        }                                                                                             \
    }
    // two output tiles of a 128-input layer in ONE step (PHASE 3): the slot holds chunk CHUNK (hi, lo) then chunk CHUNK + 1
#define SHERF_TRUNK_PAIR(CHUNK, OUT, T, IN, RELU)                                                     \
    {                                                                                                 \
        static_assert(PREC == 1, "paired steps rely on the hi+lo chunk layout being contiguous");     \
        f32x16 acc0[NTL], acc1[NTL];                                                                  \
        _Pragma("unroll") for (int u = 0; u < NTL; ++u) { acc0[u] = bias_tile(cx, CHUNK); acc1[u] = bias_tile(cx, (CHUNK) + 1); } \
        const char* s_ = cx.slot(step);                                                               \
        mma_seg<PREC, 8, NTL, IL>(s_, 0, 8, IN, acc0);                                                    \
        mma_seg<PREC, 8, NTL, IL>(s_ + 2 * 8 * 1024, 0, 8, IN, acc1);                                     \
        advance(cx, step); ++step;                                                                    \
        _Pragma("unroll") for (int u = 0; u < NTL; ++u) {                                             \
            if (RELU) { _Pragma("unroll") for (int r = 0; r < 16; ++r) { acc0[u][r] = relu(acc0[u][r]); acc1[u][r] = relu(acc1[u][r]); } } \
            split_tile<PREC>(acc0[u], OUT[u][2 * (T)], OUT[u][2 * (T) + 1]);                          \
            split_tile<PREC>(acc1[u], OUT[u][2 * (T) + 2], OUT[u][2 * (T) + 3]);                      \
        }                                                                                             \
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)     // pts_linears.0 : [PE6(x_c) (3 kb) | z_0 (2 kb)]
        SHERF_TRUNK_TILE(9 + T, ha, T, mma_seg<PREC, 3, NTL, IL>(s_, 0, 5, pe, acc); mma_seg<PREC, 2, NTL, IL>(s_, 3, 5, z0b, acc);)
#pragma unroll
    for (int L = 0; L < 4; ++L) {   // pts_linears.1-4 (ping-pong ha -> hb -> ha ...)
        if constexpr (PHASE == 3) {
#pragma unroll
            for (int T = 0; T < 4; T += 2) {
                if (L & 1) SHERF_TRUNK_PAIR(13 + 4 * L + T, ha, T, hb, true)
                else SHERF_TRUNK_PAIR(13 + 4 * L + T, hb, T, ha, true)
            }
        } else {
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                if (L & 1) SHERF_TRUNK_TILE(13 + 4 * L + T, ha, T, mma_seg<PREC, 8, NTL, IL>(s_, 0, 8, hb, acc);)
                else SHERF_TRUNK_TILE(13 + 4 * L + T, hb, T, mma_seg<PREC, 8, NTL, IL>(s_, 0, 8, ha, acc);)
            }
        }
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)     // pts_linears.5 : [PE6 | z_0 | h(128)] ; after 4 layers the activations are back in ha
        SHERF_TRUNK_TILE(29 + T, hb, T, mma_seg<PREC, 3, NTL, IL>(s_, 0, 13, pe, acc); mma_seg<PREC, 2, NTL, IL>(s_, 3, 13, z0b, acc);
                         mma_seg<PREC, 8, NTL, IL>(s_, 5, 13, ha, acc);)
    if constexpr (PHASE == 3) {
#pragma unroll
        for (int T = 0; T < 4; T += 2) SHERF_TRUNK_PAIR(33 + T, ha, T, hb, true)                              // pts_linears.6
#pragma unroll
        for (int T = 0; T < 4; T += 2) SHERF_TRUNK_PAIR(37 + T, hb, T, ha, true)                              // pts_linears.7
    } else {
#pragma unroll
        for (int T = 0; T < 4; ++T) SHERF_TRUNK_TILE(33 + T, ha, T, mma_seg<PREC, 8, NTL, IL>(s_, 0, 8, hb, acc);)   // pts_linears.6
#pragma unroll
        for (int T = 0; T < 4; ++T) SHERF_TRUNK_TILE(37 + T, hb, T, mma_seg<PREC, 8, NTL, IL>(s_, 0, 8, ha, acc);)   // pts_linears.7
    }
    // ---- heads: feature_linear (4 tiles, no activation) into ha, alpha_linear (tile 45, row 0), both from hb ----
    float sigma[NTL];
    if constexpr (PHASE == 3) {
#pragma unroll
        for (int T = 0; T < 4; T += 2) SHERF_TRUNK_PAIR(41 + T, ha, T, hb, false)
    } else {
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x16 acc[NTL];
#pragma unroll
            for (int u = 0; u < NTL; ++u) acc[u] = bias_tile(cx, 41 + T);
            mma_seg<PREC, 8, NTL, IL>(cx.slot(step), 0, 8, hb, acc);
            advance(cx, step); ++step;
#pragma unroll
            for (int u = 0; u < NTL; ++u) split_tile<PREC>(acc[u], ha[u][2 * T], ha[u][2 * T + 1]);
        }
    }
    {
        f32x16 acc[NTL];
#pragma unroll
        for (int u = 0; u < NTL; ++u) acc[u] = bias_tile(cx, 45);
        mma_seg<PREC, 8, NTL, IL>(cx.slot(step), 0, 8, hb, acc);
        advance(cx, step); ++step;
#pragma unroll
        for (int u = 0; u < NTL; ++u) sigma[u] = acc[u][0];             // row 0 lives in reg 0 of the h == 0 lanes
    }
    // ---- views_linear : [feature (8 kb) | PE4(v_c) (2 kb) | z_1 (2 kb)] -> 64, ReLU ; rgb_linear -> sigmoid ----
    BFrag<PREC> pv[NTL][2], gb[NTL][4];
#pragma unroll
    for (int u = 0; u < NTL; ++u) pe_frags<PREC, 4, 2>(h, vc[u][0], vc[u][1], vc[u][2], pv[u]);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        f32x16 acc[NTL];
#pragma unroll
        for (int u = 0; u < NTL; ++u) acc[u] = bias_tile(cx, 46 + T);
        const char* s_ = cx.slot(step);
        mma_seg<PREC, 8, NTL, IL>(s_, 0, 12, ha, acc);
        mma_seg<PREC, 2, NTL, IL>(s_, 8, 12, pv, acc);
        mma_seg<PREC, 2, NTL, IL>(s_, 10, 12, z1b, acc);
        advance(cx, step); ++step;
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = relu(acc[u][r]);
            split_tile<PREC>(acc[u], gb[u][2 * T], gb[u][2 * T + 1]);
        }
    }
    {
        f32x16 acc[NTL];
#pragma unroll
        for (int u = 0; u < NTL; ++u) acc[u] = bias_tile(cx, 48);
        mma_seg<PREC, 4, NTL, IL>(cx.slot(step), 0, 4, gb, acc);
        if constexpr (PHASE == -1) { if (cx.more) advance(cx, step); }       // frees slot (n_steps - 1) % 3 for the next group
#pragma unroll
        for (int u = 0; u < NTL; ++u)
            if (live[u] && h == 0) {
                const int64_t c = tile[u] * 32 + j;
                if (c < nv) {
                    float r = rcp_(1.0f + exp_(-acc[u][0])), g = rcp_(1.0f + exp_(-acc[u][1])), b = rcp_(1.0f + exp_(-acc[u][2]));
                    out[c] = make_float4(r * 1.002f - 0.001f, g * 1.002f - 0.001f, b * 1.002f - 0.001f, sigma[u]);   // triplane.py:314
                }
            }
    }
#if SHERF_MLP_TRACE
    if (g_mlp_trace && g_mlp_trace_every > 0 && blockIdx.x % g_mlp_trace_every == 0 && cx.lane < 64) {
        if (cx.lane == 0) cx.trace[63 * 4 + 2] = (uint32_t)__builtin_amdgcn_s_memtime();
        const size_t slot = blockIdx.x / g_mlp_trace_every;
        uint32_t* dst = g_mlp_trace + (slot * 8 + cx.wave) * 256;
        for (int i = cx.lane; i < 256; i += 64) dst[i] = cx.trace[i];
    }
#endif
    if constexpr (PHASE == -1) {
        again = cx.more;
        if (again) {
            grp += gridDim.x;
#pragma unroll
            for (int u = 0; u < NTL; ++u) {
                tile[u] = (grp * NW + cx.wave) * NTL + u;
                live[u] = tile[u] < n_tiles;
                if (!live[u]) tile[u] = n_tiles - 1;
            }
        }
    }
    } while (again);   // tile groups
#undef SHERF_TRUNK_TILE
#undef SHERF_TRUNK_PAIR
}

}  // namespace

#if SHERF_MLP_TRACE
extern "C" int sherf_mlp_set_trace(void* buf, int every) {
    SHERF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_trace), &buf, sizeof(buf)));
    SHERF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_trace_every), &every, sizeof(every)));
    return SHERF_OK;
}
#endif

extern "C" int sherf_mlp_stream_layout(int32_t* n_chunks, int32_t* nkb_host, int32_t max_chunks) {
    SHERF_CHECK_ARG(n_chunks && nkb_host && max_chunks >= N_CHUNKS);
    *n_chunks = N_CHUNKS;
    for (int c = 0; c < N_CHUNKS; ++c) nkb_host[c] = chunk_nkb(c);
    return SHERF_OK;
}

extern "C" int sherf_nerf_mlp(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                              const float* wbias, int prec, int shape, int64_t capacity, float* out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && tokens && extras && wstream && wbias && out);
    SHERF_CHECK_ARG((prec == 0 || prec == 1) && shape >= 0 && shape <= 11 && capacity > 0);
    const int64_t tiles = (capacity + 31) / 32;
    const bool wide = shape == 1;                             // <NW=4, NTL=2>: one wave per SIMD, two tiles per wave
#define SHERF_MLP(P, W, L)                                                                                                 \
    hipLaunchKernelGGL((nerf_mlp_kernel<P, W, L>), dim3((unsigned)((tiles + (W) * (L) - 1) / ((W) * (L)))), dim3((W) * 64), 0,  \
                       as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,                        \
                       reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), g_sherf_debug)
    if (shape == 10) {                        // 4x1 with the decoder phase at raised issue priority
        SHERF_CHECK_ARG(prec == 1);
        hipLaunchKernelGGL((nerf_mlp_kernel<1, 4, 1, 0, (2 << 16)>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, as_stream(stream), counters,
                           reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,
                           reinterpret_cast<float4*>(out), g_sherf_debug);
        SHERF_LAUNCH_CHECK();
    }
    if (shape == 8 || shape == 9) {           // 4 waves x 1 tile: two independent workgroups per CU (two-slot weight rings); 9 = + interleave 8
        SHERF_CHECK_ARG(prec == 1);
        if (shape == 8) SHERF_MLP(1, 4, 1);
        else
            hipLaunchKernelGGL((nerf_mlp_kernel<1, 4, 1, 0, 8>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, as_stream(stream), counters,
                               reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,
                               reinterpret_cast<float4*>(out), g_sherf_debug);
        SHERF_LAUNCH_CHECK();
    }
    if (shape >= 5) {                         // scheduling variants of the default kernel (same arithmetic, bit-identical results):
        SHERF_CHECK_ARG(prec == 1);           // 5 = MFMA/VALU interleave 8, 6 = wave priority 2, 7 = both; chosen by sherf_amd.tune
#define SHERF_MLP_VAR(V)                                                                                                     \
    hipLaunchKernelGGL((nerf_mlp_kernel<1, 8, 1, 0, V>), dim3((unsigned)((tiles + 7) / 8)), dim3(512), 0, as_stream(stream), counters, \
                       reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,  \
                       reinterpret_cast<float4*>(out), g_sherf_debug)
        if (shape == 5) SHERF_MLP_VAR(8); else if (shape == 6) SHERF_MLP_VAR(2 << 8); else SHERF_MLP_VAR(8 | (2 << 8));
#undef SHERF_MLP_VAR
        SHERF_LAUNCH_CHECK();
    }
    if (shape == 4 || shape == 11) {          // experimental: persistent workgroups (PHASE -1): 4 = 8 waves, one per CU; 11 = 4 waves, two per CU
        SHERF_CHECK_ARG(prec == 1);
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0, v = 0;
            SHERF_HIP_CHECK(hipGetDevice(&dev));
            SHERF_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
            n_cu = v > 0 ? v : 256;
        }
        if (shape == 11) {
            const int64_t groups4 = (tiles + 3) / 4;
            hipLaunchKernelGGL((nerf_mlp_kernel<1, 4, 1, -1>), dim3((unsigned)(groups4 < 2 * n_cu ? groups4 : 2 * n_cu)), dim3(256), 0,
                               as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                               reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), g_sherf_debug);
            SHERF_LAUNCH_CHECK();
        }
        const int64_t groups = (tiles + 7) / 8;
        hipLaunchKernelGGL((nerf_mlp_kernel<1, 8, 1, -1>), dim3((unsigned)(groups < n_cu ? groups : n_cu)), dim3(512), 0, as_stream(stream),
                           counters, reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,
                           reinterpret_cast<float4*>(out), g_sherf_debug);
        SHERF_LAUNCH_CHECK();
    }
    if (shape == 3) {                         // experimental: shape 2 with the decoder walking two output tiles per step (PHASE 3)
        SHERF_CHECK_ARG(prec == 1);
        hipLaunchKernelGGL((nerf_mlp_kernel<1, 8, 1, 1>), dim3((unsigned)((tiles + 7) / 8)), dim3(512), 0, as_stream(stream), counters,
                           reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,
                           reinterpret_cast<float4*>(out), g_sherf_debug);
        hipLaunchKernelGGL((nerf_mlp_kernel<1, 8, 1, 3>), dim3((unsigned)((tiles + 7) / 8)), dim3(512), 0, as_stream(stream), counters,
                           reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,
                           reinterpret_cast<float4*>(out), g_sherf_debug);
        SHERF_LAUNCH_CHECK();
    }
    if (shape == 2) {                         // experimental: transformer prologue and decoder as two launches (see PHASE)
        SHERF_CHECK_ARG(prec == 1);
        hipLaunchKernelGGL((nerf_mlp_kernel<1, 8, 1, 1>), dim3((unsigned)((tiles + 7) / 8)), dim3(512), 0, as_stream(stream), counters,
                           reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,
                           reinterpret_cast<float4*>(out), g_sherf_debug);
        hipLaunchKernelGGL((nerf_mlp_kernel<1, 8, 1, 2>), dim3((unsigned)((tiles + 7) / 8)), dim3(512), 0, as_stream(stream), counters,
                           reinterpret_cast<const float4*>(tokens), extras, reinterpret_cast<const char*>(wstream), wbias, capacity,
                           reinterpret_cast<float4*>(out), g_sherf_debug);
        SHERF_LAUNCH_CHECK();
    }
    if (prec == 0) { if (wide) SHERF_MLP(0, 4, 2); else SHERF_MLP(0, 8, 1); }
    else { if (wide) SHERF_MLP(1, 4, 2); else SHERF_MLP(1, 8, 1); }
    SHERF_LAUNCH_CHECK();
}
