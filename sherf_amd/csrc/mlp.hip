// Fused per-sample network (gfx950 MFMA): rows a13 + a14 of SURVEY.md section 8.
//
//   slot-2 token += W_b . PE5(rgb)[:32]                         (rest of conv1d_reprojection, renderer.py:423-424)
//   3-token pre-norm transformer, dim 32, 3 heads x 16            (renderer.py:949-993)
//   PE6(x_c), PE4(v_c)                                            (renderer.py:875-916)
//   NeRFDecoder 8x128 with skip, sigma / feature / view / rgb     (triplane.py:285-316)
//
// Formulation: D^T = W . X^T -- the weight matrix is the MFMA A operand (32 output features per tile, a "chunk") and 32
// samples are the B operand columns of v_mfma_f32_32x32x16_{f16,bf16}.  A wave owns 32 samples for the whole network; a
// layer's fp32 accumulator tile (lane = sample, 16 regs = 16 of the tile's 32 features) is converted in registers into two B
// operand K-blocks of the next layer: the K permutation this implies,
//     k-slot (kb, h, e)  <->  feature 16*kb + (e&3) + 8*(e>>2) + 4*h,
// is baked into the packed weight stream (sherf_amd/mlp_pack.py), so activations never leave the register file and never cross
// lanes (except the 3x3 attention dot products and LayerNorm sums: one lane^32 exchange).
//
// Schedule (round 2; measured with the in-kernel timeline of tools/mlp_trace.py, profiles/r02_mlp_trace_v1.txt): the round-1
// kernel walked one chunk per step as ONE dependent chain of 24 MFMAs; any instruction between two MFMAs on the same
// accumulator (the next fragment's ds_read, its s_waitcnt) forfeits the back-to-back forwarding, so every MFMA took ~81 cycles
// of its wave's time and a SIMD's two waves together kept the matrix pipe < 60 % busy (waits on the weight DMA and the
// workgroup barrier were only 12 %).  Here every step feeds TWO independent accumulator chains -- a pair of output chunks
// sharing the B operands (or the even / odd K-blocks of a single chunk) -- issued alternately, and the weight stream is cut
// into uniform steps of <= 20 KiB (a pair of chunks x 4-5 K-blocks x hi,lo) so that a 3-slot ring (two steps of prefetch
// distance) fits twice into a CU's LDS: two 4-wave workgroups per CU, one wave of each per SIMD, never phase-locked by each
// other's barriers -- one runs its VALU-bound transformer prologue under the other's MFMA-bound decoder.
//
// prec 1 ("f16x3", default): operands split hi + lo in fp16 (11 + 11 significant bits), three MFMAs per product
//   (lo*hi + hi*lo + hi*hi, fp32 accumulate): ~2^-21 relative, the mode that meets the 1e-3 per-sample tolerance.
// prec 0 ("bf16"): one bf16 product (north_star's nominal precision; 8 significant bits).
// prec 2 ("f16"):  one fp16 product, operands rounded to nearest even (v_cvt_pk_f16_f32): 11 significant bits at a third of the
//   MFMA issue of prec 1.  Within 1e-3 on networks of the reference's own initialisation scale (tests: the "_ri" fixtures) and an
//   order of magnitude outside it on the adversarial seeded weights -- sherf_amd/renderer.py: mlp_precision='auto' measures which.
#include "common.h"

#include <algorithm>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));     // a 16-byte MFMA operand fragment (native vector: usable as an asm "v" operand)

constexpr int N_CHUNKS = 49;
constexpr int N_STEPS = 43;
constexpr int NW = 4;               // waves per workgroup
constexpr int NSLOT = 3;            // ring slots: two steps of prefetch distance
constexpr int BIAS_LN = N_CHUNKS;   // index of the first LayerNorm table in wbias ([idx][2][16] floats)

// ---- the weight stream: N_STEPS steps of `step_units` units; a unit = the A fragments (hi [, lo]: 1 KiB each) of one
//      (chunk, K-block); a chunk = one 32-row output tile of one layer (49 of them, sherf_amd/mlp_pack.py).  Steps: 0-1
//      transformer (chunks 0-4 | 5-8 + one padding unit), 2-3 pts_linears.0 (a pair of chunks x 5 kb each), 4-19
//      pts_linears.1-4 (pair x K-half), 20-25 pts_linears.5 (pair x {5, 4, 4} kb), 26-33 pts_linears.6-7, 34-37
//      feature_linear, 38 alpha_linear (8 kb), 39-41 views_linear (pair x 4 kb), 42 rgb_linear (4 kb). ----
__host__ __device__ constexpr int step_units(int s) {
    return s < 4 ? 10 : s < 20 ? 8 : s < 26 ? ((s - 20) % 3 == 0 ? 10 : 8) : s < 42 ? 8 : 4;
}
// unit u of step s -> chunk * 16 + kb  (-1 = padding); the order in which the kernel consumes them
__host__ __device__ constexpr int step_unit(int s, int u) {
    int c = -1, kb = 0;
    if (s == 0) { if (u < 2) { c = 0; kb = u; } else { c = 1 + (u - 2) / 2; kb = (u - 2) % 2; } }
    else if (s == 1) {
        if (u < 2) { c = 5; kb = u; } else if (u < 5) { c = 6; kb = u - 2; } else if (u < 7) { c = 7; kb = u - 5; } else if (u < 9) { c = 8; kb = u - 7; }
    }
    else if (s < 4) { c = 9 + 2 * (s - 2) + (u & 1); kb = u / 2; }
    else if (s < 20) { const int q = s - 4; c = 13 + 4 * (q / 4) + 2 * ((q % 4) / 2) + (u & 1); kb = 4 * (q % 2) + u / 2; }
    else if (s < 26) { const int q = s - 20, seg = q % 3; c = 29 + 2 * (q / 3) + (u & 1); kb = (seg == 0 ? 0 : seg == 1 ? 5 : 9) + u / 2; }
    else if (s < 34) { const int q = s - 26; c = 33 + 4 * (q / 4) + 2 * ((q % 4) / 2) + (u & 1); kb = 4 * (q % 2) + u / 2; }
    else if (s < 38) { const int q = s - 34; c = 41 + 2 * (q / 2) + (u & 1); kb = 4 * (q % 2) + u / 2; }
    else if (s == 38) { c = 45; kb = u; }
    else if (s < 42) { c = 46 + (u & 1); kb = 4 * (s - 39) + u / 2; }
    else { c = 48; kb = u; }
    return c < 0 ? -1 : c * 16 + kb;
}
template <int PREC> constexpr int NPIECE = PREC == 1 ? 2 : 1;                     // 1 KiB fragments per unit: hi [, lo]
template <int PREC> __host__ __device__ constexpr int step_pieces(int s) {      // 1 KiB pieces, padded to one DMA round of the 4 waves
    return s < 0 || s >= N_STEPS ? 0 : (step_units(s) * NPIECE<PREC> + NW - 1) / NW * NW;
}
template <int PREC> __host__ __device__ constexpr int step_off_kib(int s) {
    int o = 0;
    for (int i = 0; i < s; ++i) o += step_pieces<PREC>(i);
    return o;
}

template <int PREC> struct BFrag;
template <> struct BFrag<0> { u32x4 hi; };
template <> struct BFrag<1> { u32x4 hi, lo; };
template <> struct BFrag<2> { u32x4 hi; };

__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pack2_f16(float a, float b) {      // round to nearest even: one v_cvt_pk_f16_f32
    f16x2 v;
    v[0] = (_Float16)a; v[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, v);
}
// f16 hi/lo split of a pair: hi = the pair rounded toward zero (one v_cvt_pkrtz), lo = x - hi exactly in fp32 (hi keeps x's
// leading 11 bits) and then rounded to fp16: hi + lo carries 22 significant bits.  |x| must stay below 65504 (fp16 range): the
// activations of an 8 x 128 ReLU network fed with encodings in [-1, 1] and O(1) tokens sit 3-4 orders of magnitude below that.
// SHERF_MLP_FMA_MIX: the residuals x - hi as ONE v_fma_mix_f32 each (f16 half of `hi` times -1.0 plus the f32 value) instead of
// v_cvt_f32_f16 + v_sub_f32: 4 instead of 6 VALU per pair -- hipcc does not form it.  Verified on the MI355X: bit-identical output,
// 0.665 vs 0.676 ms (profiles/r02_mlp_trace_v4.txt).  The host build of the test shim takes the plain-C form below.
#ifndef SHERF_MLP_FMA_MIX
#define SHERF_MLP_FMA_MIX 1
#endif
__device__ __forceinline__ void split2_f16(float a, float b, uint32_t& hi, uint32_t& lo) {
#if SHERF_MLP_FMA_MIX
    float ra, rb;
    // (the conversion itself is the builtin, NOT asm: an asm-written register feeding a compiler-visible MFMA gets no VALU -> MFMA
    //  wait states from hipcc's hazard recognizer -- tools/mfma_hazard_check.py found three such sites, one wait state short, in round 3's build)
    hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(ra, rb));
#else
    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    const float ra = __builtin_fmaf((float)h[0], -1.0f, a), rb = __builtin_fmaf((float)h[1], -1.0f, b);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(ra, rb));
#endif
}

template <int PREC>
__device__ __forceinline__ BFrag<PREC> make_frag(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    BFrag<PREC> f;
    if constexpr (PREC == 1) {
        uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
        split2_f16(v0, v1, h0, l0); split2_f16(v2, v3, h1, l1);
        split2_f16(v4, v5, h2, l2); split2_f16(v6, v7, h3, l3);
        f.hi = u32x4{h0, h1, h2, h3}; f.lo = u32x4{l0, l1, l2, l3};
    } else if constexpr (PREC == 2) {
        f.hi = u32x4{pack2_f16(v0, v1), pack2_f16(v2, v3), pack2_f16(v4, v5), pack2_f16(v6, v7)};
    } else {
        f.hi = u32x4{pack2_bf16(v0, v1), pack2_bf16(v2, v3), pack2_bf16(v4, v5), pack2_bf16(v6, v7)};
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

template <int PREC>
__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (PREC != 0)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// SHERF_MLP_TRACE (profiling builds only, tools/mlp_trace.py): every wave stamps s_memtime at the end of its MFMA stream, after the
// weight-DMA wait and after the workgroup barrier of every step into LDS; selected workgroups copy the stamps out at the end.
#ifndef SHERF_MLP_TRACE
#define SHERF_MLP_TRACE 0
#endif
#if SHERF_MLP_TRACE
__device__ uint32_t* g_mlp_trace = nullptr;           // [slot][wave 0..7][64 steps][4] u32
__device__ int g_mlp_trace_every = 0;
// (two-tile kernel: its LDS is full -- stamps go into lanes of three VGPRs, index 8 + 3 (step - 2) + k)
#define SHERF_TRACE_STAMP(cx, step, k) do { if ((cx).treg) trace_reg((cx), 8 + 3 * ((step) - 2) + (k)); \
                                            else if ((cx).lane == 0) (cx).trace[(step) * 4 + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SHERF_TRACE_STAMP(cx, step, k) do { } while (0)
#endif

// SHERF_MLP_ABLATE (profiling builds only; results are garbage): 32 = no weight DMA, 64 = no workgroup barriers, 256 = no transformer arithmetic (z = the raw tokens), 512 = no positional
// encodings (zero fragments), 1024 = layer epilogues without conversion / ReLU (16 moves instead of 32 VALU), 2048 (pipelined decoder; round 6) = NO layer
// epilogue at all: every layer's B operands stay the fp16 fused tokens z_0 / z_1 (real data, real weights, the full MFMA / ring / barrier stream):
// with 256 + 512 the "MFMA-only" instruction stream whose duration is what the power cap leaves the matrix pipe on realistic operands
#ifndef SHERF_MLP_ABLATE
#define SHERF_MLP_ABLATE 0
#endif

template <int PREC> struct Ctx {
    const char* ws;          // packed weight stream (global)
    const float* wbias;      // [N_CHUNKS + 4][2][16] D-layout bias / LayerNorm tables (LDS copy)
    const char* lds;         // NSLOT ring slots (generic pointer, + this lane's 16 bytes: what the ds_reads use)
    uint32_t lds_addr;       // LDS byte address of the ring (what the DMA's M0 takes), wave-uniform
    int lane, h, wave;
    int notrans;             // use_trans = False (renderer.py:261, 427): the decoder reads the fused tokens 0 / 1 as they are
#if SHERF_MLP_TRACE
    uint32_t* trace;         // this wave's [64][4] stamps in LDS
    bool treg;               // stamps in registers instead (two-tile kernel)
    uint32_t tv[3];
#endif
    static constexpr int UNIT = NPIECE<PREC> * 1024;                       // bytes per unit: hi [, lo]
    static constexpr int SLOT = (PREC == 1 ? 20 : 12) * 1024;              // the largest step, padded
    __device__ __forceinline__ const char* slot(int step) const { return lds + (step % NSLOT) * SLOT; }
};

#if SHERF_MLP_TRACE
template <class C>
__device__ __forceinline__ void trace_put(C& cx, int idx, uint32_t v) {      // lane (idx % 64) of register idx / 64 (a compare + a select)
    if (idx < 64) cx.tv[0] = cx.lane == idx ? v : cx.tv[0];
    else if (idx < 128) cx.tv[1] = cx.lane == idx - 64 ? v : cx.tv[1];
    else cx.tv[2] = cx.lane == idx - 128 ? v : cx.tv[2];
}
template <class C>
__device__ __forceinline__ void trace_reg(C& cx, int idx) {
    trace_put(cx, idx, (uint32_t)__builtin_amdgcn_s_memtime());
}
#define SHERF_TRACE_REG(cx, idx) trace_reg((cx), (idx))
#else
#define SHERF_TRACE_REG(cx, idx) do { } while (0)
#endif

// Weight stream L2 -> LDS by LDS-DMA (global_load_lds, 1 KiB per wave-instruction, no VGPR round trip).  Every step is a whole
// number of rounds of the four waves (the stream is padded), so the issue is branch free.  Completion: a wave waits (counted
// vmcnt) until only the pieces of the most recently issued step are still in flight, then the workgroup barrier makes every
// wave's pieces visible.
typedef __attribute__((address_space(3))) void* lptr_t;

template <int PREC>
__device__ __forceinline__ void dma_issue(Ctx<PREC>& cx, int step) {
    if (step >= N_STEPS || (SHERF_MLP_ABLATE & 32)) return;
    const char* src = cx.ws + (size_t)step_off_kib<PREC>(step) * 1024;          // (+ this lane's 16 bytes: folded into cx.ws)
    const uint32_t dst = cx.lds_addr + (step % NSLOT) * Ctx<PREC>::SLOT;
#pragma unroll
    for (int i = 0; i < step_pieces<PREC>(step) / NW; ++i) {
        // piece p = wave + 4 i (the wave's own 1 KiB offset is folded into cx.ws / cx.lds_addr).  Inline asm on purpose: hipcc
        // drains an LDS-DMA it knows about (vmcnt(0)) before every ds_read that might alias it, which would serialise the ring.
        // M0 = LDS byte address of the piece (saved/restored: compiler-reserved).
        const char* g = src + i * NW * 1024;
        const uint32_t l = dst + i * NW * 1024;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(l) : "memory");
    }
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

// end of step s: step s+1 must have landed (everything but the newest issue, step s+2), then step s's slot is recycled for s+3
template <int PREC>
__device__ __forceinline__ void step_wait(Ctx<PREC>& cx, int step) {
    SHERF_TRACE_STAMP(cx, step, 0);
    wait_vm((SHERF_MLP_ABLATE & 32) ? 0 : step_pieces<PREC>(step + 2) / NW);
    SHERF_TRACE_STAMP(cx, step, 1);
    if (!(SHERF_MLP_ABLATE & 64)) wg_barrier();
    SHERF_TRACE_STAMP(cx, step, 2);
}
template <int PREC>
__device__ __forceinline__ void advance(Ctx<PREC>& cx, int step) {
    step_wait(cx, step);
    dma_issue(cx, step + NSLOT);
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
__device__ __forceinline__ float exp_(float x) { return __expf(x); }
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32 (1 ulp); __frcp_rn is a full IEEE division: ~10 instructions
__device__ __forceinline__ float rsqrt_(float x) { return __frsqrt_rn(x); }
#else
__device__ __forceinline__ float exp_(float x) { return expf(x); }
__device__ __forceinline__ float rcp_(float x) { return 1.0f / x; }
__device__ __forceinline__ float rsqrt_(float x) { return 1.0f / sqrtf(x); }
#endif
// SHERF_MLP_FAST_ERF: Abramowitz-Stegun 7.1.26 for the exact-GELU erf, |error| <= 1.5e-7 (fp32 evaluation: tests/test_mlp_pack.py),
// ~14 straight-line VALU instead of libm's ~38 + branches -- the 32 erf evaluations per lane were the largest VALU block of the prologue.
#ifndef SHERF_MLP_FAST_ERF
#define SHERF_MLP_FAST_ERF 1
#endif
__device__ __forceinline__ float erf_(float x) {
#if SHERF_MLP_FAST_ERF
    const float ax = fabsf(x);
    const float t = rcp_(fmaf(0.3275911f, ax, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    return copysignf(1.0f - poly * __expf(-ax * ax), x);
#else
    return erff(x);
#endif
}
// exact GELU 0.5 a (1 + erf(a / sqrt 2)) (renderer.py:940: nn.GELU()).
//   FAST = false: through erf_ above (|error| <= 1.5e-7): the fp32-grade form prec 1 keeps.
//   FAST = true (single-product precisions, round 4): gelu = relu(a) - |a| / 2 * y,  y = (a1 t + a2 t^2 + a3 t^3) exp(-a^2 / 2),
//     t = 1 / (1 + p |a| / sqrt 2)  (Abramowitz-Stegun 7.1.25, |erf error| <= 2.5e-5 -> |gelu error| <= 1.3e-5 |a|: a twentieth of the
//     fp16 rounding its consumer applies to it); written so that the sign of `a` never needs restoring: 11 VALU + 2 transcendentals
//     instead of 16 + 2 -- the 32 evaluations per lane are the largest single VALU block of the transformer.
template <bool FAST>
__device__ __forceinline__ float gelu_(float a) {
    if constexpr (!FAST) return 0.5f * a * (1.0f + erf_(a * 0.70710678118654752f));
    const float aa = fabsf(a);
    const float t = rcp_(fmaf(0.47047f * 0.70710678118654752f, aa, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, 0.7478556f, -0.0958798f), 0.3480242f);
#if SHERF_MLP_FASTMATH
    const float e = __builtin_amdgcn_exp2f(a * a * -0.72134752044448170f);           // exp(-a^2 / 2) = 2^(-a^2 log2(e) / 2)
#else
    const float e = expf(-0.5f * a * a);
#endif
    const float hy = 0.5f * poly * e;
    return fmaf(-aa, hy, fmaxf(a, 0.0f));
}
// SHERF_MLP_DECODER_PRIO: issue priority of a wave once it enters the MFMA-bound decoder (over the co-resident workgroup's wave on
// the same SIMD whenever that one is in its VALU-bound prologue; a new wave starts at priority 0)
#ifndef SHERF_MLP_DECODER_PRIO
#define SHERF_MLP_DECODER_PRIO 2
#endif

// ---- MFMA segments.  `s` = the step's slot (+ this lane's 16 bytes); unit u of the step lives at s + u * UNIT: hi, then lo. ----
// NCOL column sets (tokens) sharing the A fragments of ONE chunk: the transformer's independent chains
template <int PREC, int NK, int NCOL>
__device__ __forceinline__ void mma_cols(const char* s, int u0, const BFrag<PREC> (&b)[NCOL][NK], f32x16 (&acc)[NCOL]) {
    constexpr int UNIT = Ctx<PREC>::UNIT;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
        const u32x4 ah = *reinterpret_cast<const u32x4*>(s + (u0 + kb) * UNIT);
        if constexpr (PREC == 1) {
            const u32x4 al = *reinterpret_cast<const u32x4*>(s + (u0 + kb) * UNIT + 1024);
#pragma unroll
            for (int t = 0; t < NCOL; ++t) acc[t] = mfma<PREC>(al, b[t][kb].hi, acc[t]);
#pragma unroll
            for (int t = 0; t < NCOL; ++t) acc[t] = mfma<PREC>(ah, b[t][kb].lo, acc[t]);
        }
#pragma unroll
        for (int t = 0; t < NCOL; ++t) acc[t] = mfma<PREC>(ah, b[t][kb].hi, acc[t]);
    }
}
// Six (PREC 1) / two (PREC 0) MFMAs of one K-block on TWO accumulator chains, alternating, as ONE ordered asm block: hipcc's
// scheduler otherwise mixes the chains at will and drops the next fragments' ds_reads and their waits between dependent MFMAs
// (the "memory" clobber keeps the compiler's LDS reads on their side of the block: a fetch written before it stays before it).
// Hazards hipcc cannot see inside an asm statement (cdna_hip_programming.md 5.7): the block opens with `s_nop 1` (a VALU-written B
// fragment -> MFMA operand) and every READER of the accumulators other than the next block goes through mfma_settle() first.
template <int PREC>
__device__ __forceinline__ void mfma_block(f32x16& acc0, f32x16& acc1, const u32x4& ah0, const u32x4& al0, const u32x4& ah1, const u32x4& al1,
                                           const u32x4& bh0, const u32x4& bl0, const u32x4& bh1, const u32x4& bl1) {
    if constexpr (PREC == 1)
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %3, %6, %0\n\t"
                     "v_mfma_f32_32x32x16_f16 %1, %5, %8, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %2, %7, %0\n\t"
                     "v_mfma_f32_32x32x16_f16 %1, %4, %9, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %2, %6, %0\n\t"
                     "v_mfma_f32_32x32x16_f16 %1, %4, %8, %1"
                     : "+v"(acc0), "+v"(acc1) : "v"(ah0), "v"(al0), "v"(ah1), "v"(al1), "v"(bh0), "v"(bl0), "v"(bh1), "v"(bl1) : "memory");
    else if constexpr (PREC == 2)
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %2, %4, %0\n\t"
                     "v_mfma_f32_32x32x16_f16 %1, %3, %5, %1"
                     : "+v"(acc0), "+v"(acc1) : "v"(ah0), "v"(ah1), "v"(bh0), "v"(bh1) : "memory");
    else
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %3, %5, %1"
                     : "+v"(acc0), "+v"(acc1) : "v"(ah0), "v"(ah1), "v"(bh0), "v"(bh1) : "memory");
}
// an MFMA's result -> any reader other than the next MFMA taking it whole as C: 12 wait states for the 8-pass XDL ops
__device__ __forceinline__ void mfma_settle(f32x16& acc0, f32x16& acc1) {
    asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc0), "+v"(acc1));
}

// the A fragments of two units (p, p + UNIT): a pair of chunks at one K-block, or one chunk at two consecutive K-blocks
template <int PREC> struct AFrag { u32x4 h0, l0, h1, l1; };
template <int PREC>
__device__ __forceinline__ AFrag<PREC> load_units(const char* p) {
    constexpr int UNIT = Ctx<PREC>::UNIT;
    AFrag<PREC> f;
    f.h0 = *reinterpret_cast<const u32x4*>(p); f.h1 = *reinterpret_cast<const u32x4*>(p + UNIT);
    if constexpr (PREC == 1) { f.l0 = *reinterpret_cast<const u32x4*>(p + 1024); f.l1 = *reinterpret_cast<const u32x4*>(p + UNIT + 1024); }
    else { f.l0 = f.h0; f.l1 = f.h1; }
    return f;
}
// NB blocks over consecutive unit pairs (u0 + 2 i, u0 + 2 i + 1) of the step's slot.  `cur` holds the fragments of the first pair
// (loaded by the caller: right after the step's barrier, so that their LDS latency hides under the DMA issue / the previous
// pair's epilogue); every block's successor is fetched BEFORE the block is issued -- the asm's "memory" clobber pins that order
// -- and MORE says that another segment of the same step follows at u0 + 2 NB (its first pair is then left in `cur`).
//   PAIR = true : units = (chunk 0, chunk 1) at K-block i, both chains take b[i]             (a pair of output chunks)
//   PAIR = false: units = (kb 2i, kb 2i+1) of ONE chunk, chain 0 takes b[2i], chain 1 b[2i+1] (split-K: the caller adds the chains)
template <int PREC, int NB, bool PAIR, bool MORE>
__device__ __forceinline__ void mma_chains(const char* s, int u0, const BFrag<PREC>* b, f32x16& acc0, f32x16& acc1, AFrag<PREC>& cur) {
    constexpr int UNIT = Ctx<PREC>::UNIT;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        AFrag<PREC> nxt;
        const bool pre = i + 1 < NB || MORE;
        if (pre) nxt = load_units<PREC>(s + (u0 + 2 * (i + 1)) * UNIT);
        const BFrag<PREC>& b0 = b[PAIR ? i : 2 * i];
        const BFrag<PREC>& b1 = b[PAIR ? i : 2 * i + 1];
        if constexpr (PREC == 1) mfma_block<PREC>(acc0, acc1, cur.h0, cur.l0, cur.h1, cur.l1, b0.hi, b0.lo, b1.hi, b1.lo);
        else mfma_block<PREC>(acc0, acc1, cur.h0, cur.h0, cur.h1, cur.h1, b0.hi, b0.hi, b1.hi, b1.hi);
        if (pre) cur = nxt;
    }
}
template <int PREC, int NK, bool MORE = false>
__device__ __forceinline__ void mma_pair(const char* s, int u0, const BFrag<PREC>* b, f32x16& acc0, f32x16& acc1, AFrag<PREC>& cur) {
    mma_chains<PREC, NK, true, MORE>(s, u0, b, acc0, acc1, cur);
}
template <int PREC, int NK>
__device__ __forceinline__ void mma_splitk(const char* s, int u0, const BFrag<PREC>* b, f32x16& acc0, f32x16& acc1, AFrag<PREC>& cur) {
    static_assert(NK % 2 == 0, "even / odd K-block chains");
    mma_chains<PREC, NK / 2, false, false>(s, u0, b, acc0, acc1, cur);
}

// v + (the partner lane's v): the partner lane (lane ^ 32) holds the other 16 features of the sample.  SHERF_MLP_PERMLANE (round 4):
// one v_permlane32_swap_b32 (VALU, no LDS round trip) instead of ds_bpermute_b32 + s_waitcnt lgkmcnt(0) (36 exchanges per tile).  The swap
// leaves (lo, lo) in one register and (hi, hi) in the other; their sum is lo + hi in every lane: the same two addends as before.
// Measured: no difference in kernel time (profiles/r04_call_a_*.txt, build variant `noperm`); kept for the 36 LDS operations it removes.
#ifndef SHERF_MLP_PERMLANE
#define SHERF_MLP_PERMLANE 1
#endif
__device__ __forceinline__ float xhalf_sum(float v) {
#if SHERF_MLP_PERMLANE
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
#else
    return v + __shfl_xor(v, 32);
#endif
}

// LayerNorm over the 32 features of a token (16 here, 16 in lane^32), eps 1e-5 (renderer.py:931).
// prec 1 (fp32-grade): the two-pass form of round 1-3 (mean, then the sum of squared deviations).  Single-product precisions (round 4):
// one pass -- sum and sum of squares together, var = E[x^2] - mean^2 (32 O(1) values in fp32: the cancellation costs ~1e-7; tokens whose
// mean dwarfs their spread fall back to the two-pass variance, see the guard below) -- and the
// normalisation as two fmas per feature ((x * inv - mean * inv) * g + b): 64 instead of 96 VALU per token, five tokens per tile.
template <int PREC, class C>
__device__ __forceinline__ void layer_norm(const C& cx, const f32x16& x, int ln_idx, BFrag<PREC>& k0, BFrag<PREC>& k1) {
    const f32x16 g = bias_tile(cx, BIAS_LN + 2 * ln_idx), bt = bias_tile(cx, BIAS_LN + 2 * ln_idx + 1);
    f32x16 y;
    if constexpr (PREC == 1) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x[r];
        s = xhalf_sum(s);
        const float mean = s * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { float d = x[r] - mean; q += d * d; }
        q = xhalf_sum(q);
        const float inv = rsqrt_(q * (1.0f / 32.0f) + 1e-5f);
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = (x[r] - mean) * inv * g[r] + bt[r];
    } else {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s += x[r]; q = __builtin_fmaf(x[r], x[r], q); }
        s = xhalf_sum(s); q = xhalf_sum(q);
        const float mean = s * (1.0f / 32.0f), ex2 = q * (1.0f / 32.0f);
        float var = fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.0f);
        // Cancellation guard (round 6): E[x^2] - mean^2 loses log2(E[x^2] / var) bits.  When some sample of
        // the wave has a mean above 16 standard deviations (more than eight of the 24 bits gone) the whole wave takes the two-pass variance --
        // a wave-uniform branch that tokens of O(1) mean never enter (3 VALU + a ballot per token in the common case).
        if (__ballot(ex2 > 257.0f * var) != 0) {
            float q2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = x[r] - mean; q2 = __builtin_fmaf(d, d, q2); }
            var = xhalf_sum(q2) * (1.0f / 32.0f);
        }
        const float inv = rsqrt_(var + 1e-5f), off = -mean * inv;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = __builtin_fmaf(__builtin_fmaf(x[r], inv, off), g[r], bt[r]);
    }
    split_tile<PREC>(y, k0, k1);
}

// sin / cos with the exact phase reduction: csrc/common.h (shared with the gather, which can hand the encodings over as fragments)
__device__ __forceinline__ void sincos_exact_phase(float a, float* s, float* c) { sherf_sincos_exact_phase(a, s, c); }

// NeRF positional encoding in "natural" K-block order: feature f = 16*kb + 8*h + e of
// [x(3), sin(2^0 x)(3), cos(2^0 x)(3), sin(2^1 x)(3), ...]; entries >= 3 + 6*NF are zero.  Every octave is evaluated DIRECTLY from
// 2^q x (exact in fp32), like the reference's sin(phase + x * f) (renderer.py:906): round 1 doubled the angle octave by octave
// (s' = 2 s c, c' = 1 - 2 s^2), which doubles the error per octave as well -- 6e-6 on the top octave of PE6 from fp32 rounding alone,
// 6e-5 with a 5e-7 error of the first sine -- and the decoder multiplies that by its gain (the tail of the per-sample sigma error).
template <int PREC, int NF, int NKB>
__device__ __forceinline__ void pe_frags(int h, float x, float y, float z, BFrag<PREC> (&out)[NKB]) {
    if constexpr ((SHERF_MLP_ABLATE & 512) != 0) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) out[kb] = make_frag<PREC>(x, y, z, 0.f, 0.f, 0.f, 0.f, 0.f);
        return;
    }
    float f[NKB * 16];
#pragma unroll
    for (int i = 0; i < NKB * 16; ++i) f[i] = 0.f;
    f[0] = x; f[1] = y; f[2] = z;
    const float v3[3] = {x, y, z};
#pragma unroll
    for (int q = 0; q < NF; ++q) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float s, c;
            sincos_exact_phase(v3[a] * (float)(1 << q), &s, &c);
            if (3 + 6 * q + a < NKB * 16) f[3 + 6 * q + a] = s;
            if (6 + 6 * q + a < NKB * 16) f[6 + 6 * q + a] = c;
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

// SHERF_MLP_PK_RELU (prec 2): ReLU on the PACKED fp16 pairs after the conversion (v_pk_max_f16: one instruction per two values)
// instead of one integer max per fp32 value before it -- relu(round(x)) == round(relu(x)), 272 VALU less per tile of a kernel that
// issues ~2 900 of them beside 374 MFMAs (profiles/r03_mlp_isa_mix.txt).
// Measured on the MI355X (profiles/r03_mlp_variants.txt): 0.306 -> 0.292 ms, bit-identical; together with -fno-slp-vectorize
// (packed fp32 VALU beside MFMAs costs more than the pair it replaces: MI355X_MICROARCH.md) 0.288 ms.
#ifndef SHERF_MLP_PK_RELU
#define SHERF_MLP_PK_RELU 1
#endif
__device__ __forceinline__ uint32_t relu2_f16(uint32_t p) {
    f16x2 v = __builtin_bit_cast(f16x2, p);
    v = __builtin_elementwise_max(v, f16x2{(_Float16)0.0f, (_Float16)0.0f});
    return __builtin_bit_cast(uint32_t, v);
}

// two finished accumulator tiles (a pair of chunks) -> four K-blocks of the next layer; pinned in place: left alone the compiler
// sinks every epilogue of a layer in front of the next layer's first MFMA and keeps all the raw accumulators alive until then
template <int PREC, bool RELU>
__device__ __forceinline__ void finish_pair(f32x16& acc0, f32x16& acc1, BFrag<PREC>* out) {
    mfma_settle(acc0, acc1);
    if constexpr ((SHERF_MLP_ABLATE & 1024) != 0) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[f].hi[e] = __builtin_bit_cast(uint32_t, f < 2 ? acc0[4 * f + e] : acc1[4 * (f - 2) + e]);
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    constexpr bool PK = RELU && PREC == 2 && SHERF_MLP_PK_RELU;
    if constexpr (RELU && !PK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = relu(acc0[r]); acc1[r] = relu(acc1[r]); }
    }
    split_tile<PREC>(acc0, out[0], out[1]);
    split_tile<PREC>(acc1, out[2], out[3]);
    if constexpr (PK) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[f].hi[e] = relu2_f16(out[f].hi[e]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// NOT fused with the taps of a10-a12 in front of it: round 2 built that (the gather as this kernel's prologue, every wave filling
// its own tile's token registers while the co-resident workgroup's decoder keeps the MFMA pipe busy; commit 42beabf).  It is
// bit-identical to the two launches on the CPU build and took the frame from 2.00 to 1.85 ms, but on the MI355X ~12 % of the tiles
// came out wrong, a different set every launch -- with correct tokens in memory, with the taps ahead of any weight DMA, with every
// counted wait replaced by vmcnt(0), even with the taps replaced by plain loads of the stored tokens (profiles/
// r02_fused_gather_mlp_experiment_*.txt).  Unexplained, so not shipped; sherf_gather_tokens stays its own launch.
// One launch (nerf_mlp_kernel): transformer (steps 0-1) + decoder (steps 2-42).  Round 2 measured the two as SEPARATE launches on the
// three-product precision: 0.23 + 0.62 ms against 0.69 ms fused (profiles/r02_kernel_trace_v2_split.txt).  Round 4 measured the
// single-product precision every way round (profiles/r04_call_[a-e]_*.txt, cfg2_dense_ri, 1.27 M samples):
//   one launch 0.51 ms; two launches (sherf_nerf_mlp_split below) 0.62 ms = transformer 0.20 + decoder 0.40 (the decoder ALONE keeps the
//   matrix pipe 51 % busy: it is not the transformer that holds it back); with the weight DMA, the barriers and the per-step fragment
//   reads all ablated 0.46 ms; without the transformer's arithmetic 0.42 ms; SQ counters: MFMA busy 46 % + VALU issue 57 % of the
//   cycles -- the two add up, a tile's 2 900 VALU and 374 MFMAs barely overlap.  A microbenchmark (tools/ubench) shows they overlap
//   perfectly inside ONE wave's stream (8 MFMAs + 64 VALU per iteration run at the 8 MFMAs' speed), so a persistent kernel that runs
//   tile-group g's decoder and group g+1's transformer in the same stream was built (commits 0b79420..23ea087: builtin MFMAs,
//   sched_group_barrier pipelines, fenced parts; two waves per SIMD at 255 registers without a spill once the weight DMA took its
//   address from SGPRs): 0.55-0.56 ms in every variant -- no gain, and not bit-identical on the hardware (fma contraction order), so
//   it is not in the tree.  What stays from round 4: the cheaper single-product LayerNorm / GELU (-250 VALU per tile), the lane^32
//   exchange as v_permlane32_swap, the static hazard audit (tools/mfma_hazard_check.py), the two-launch form as a tested option.
// ---- the two halves of the network as device functions: one launch runs both (nerf_mlp_kernel), the two-launch form runs them as
//      nerf_tokens_kernel + nerf_decoder_kernel (below) ----
// Transformer (chunks 0..8 = steps 0-1 of the weight stream) of one 32-sample tile -> the fused tokens z_0, z_1 as K-blocks.
//   RING = true : the weights walk the 3-slot ring (step 0 / 1 in slots 0 / 1; the two advance() calls recycle them)
//   RING = false: the weights are resident at cx.lds (steps 0-1 back to back), no barrier, no DMA: waves run independently
// a tile's three tokens in D layout (quad q = 2i+h -> regs 4i..4i+3)
__device__ __forceinline__ void load_tokens(const float4* __restrict__ tokens, int64_t tile, int j, int h, f32x16 (&tok)[3]) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = tokens[((tile * 3 + t) * 8 + (2 * i + h)) * 32 + j];
            tok[t][4 * i] = v.x; tok[t][4 * i + 1] = v.y; tok[t][4 * i + 2] = v.z; tok[t][4 * i + 3] = v.w;
        }
}
//   PRE = false: the tile's tokens are read here, and tokens 0 / 1 RE-read for the residual (the one-tile kernels: 32 registers less at
//                their pressure peak);  PRE = true: `tok` arrives loaded (the two-tile kernel prefetches the next tile's tokens under
//                this tile's arithmetic and keeps tokens 0 / 1 for the residual: it has the 256-register budget)
//   PEM = true (round 6, prec 2 only): PE5(rgb) arrives as two ready-made fp16 fragments (`pef`: sherf_gather_tokens_pe's pefrag[tile][5..6][lane],
//                the same bits pe_frags<2, 5, 2> would produce) instead of 15 sin / cos pairs + selects + conversions per lane
template <int PREC, bool RING, bool PRE = false, bool PEM = false>
__device__ __forceinline__ void transformer_tile(Ctx<PREC>& cx, const float4* __restrict__ tokens, const float* __restrict__ extras, int64_t tile,
                                                 BFrag<PREC> (&z0b)[2], BFrag<PREC> (&z1b)[2], f32x16 (&tok)[3], const u32x4* __restrict__ pef = nullptr) {
    const int j = cx.lane & 31, h = cx.h;
    if (cx.notrans) {
        // use_trans = False: no slot-2 completion, no transformer -- `sampled_features` go to the decoder unchanged (renderer.py:427, 432); the
        // weight stream's first two steps are walked past (wave-uniform branch: a kernel argument)
        if constexpr (!PRE) load_tokens(tokens, tile, j, h, tok);
        split_tile<PREC>(tok[0], z0b[0], z0b[1]);
        split_tile<PREC>(tok[1], z1b[0], z1b[1]);
        if constexpr (RING) { advance(cx, 0); advance(cx, 1); }
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    {
        // ---- inputs: tokens in D layout, extras ----
        if constexpr (!PRE) load_tokens(tokens, tile, j, h, tok);
        const float* ex = extras + tile * 12 * 32 + j;
        if constexpr ((SHERF_MLP_ABLATE & 256) != 0) {
            split_tile<PREC>(tok[0] + tok[2], z0b[0], z0b[1]);
            split_tile<PREC>(tok[1] + tok[2], z1b[0], z1b[1]);
            if constexpr (RING) { advance(cx, 0); advance(cx, 1); }
            return;
        }

        const char* s = RING ? cx.slot(0) : cx.lds;               // step 0: chunks 0..4
        // ---- chunk 0: slot-2 token += W_b . PE5(rgb)[:32] ----
        {
            BFrag<PREC> b[1][2];
            if constexpr (PEM) { b[0][0].hi = pef[(tile * 7 + 5) * 64 + cx.lane]; b[0][1].hi = pef[(tile * 7 + 6) * 64 + cx.lane]; }
            else pe_frags<PREC, 5, 2>(h, ex[192], ex[224], ex[256], b[0]);
            f32x16 acc[1] = {bias_tile(cx, 0)};
            mma_cols<PREC, 2, 1>(s, 0, b, acc);
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
            mma_cols<PREC, 2, 2>(s, 2, b2, acc);                      // [q head0 | q head1]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) qa[i][r] = acc[i][r];
            f32x16 acc2[2] = {bias_tile(cx, 2), bias_tile(cx, 2)};
            mma_cols<PREC, 2, 2>(s, 4, b2, acc2);                     // [q head2 | pad]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) qb[i][r] = acc2[i][r];
        }
        float dot[2][3][3];                                     // [query token][head][key token]
        float o[2][3][8];                                       // attention output [query][head][8 of 16 dims]
        float v0[3][8];
        // keys / values token by token (one 16-register accumulator instead of three: the kernel's register peak sits here)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const BFrag<PREC> (&lt)[1][2] = reinterpret_cast<const BFrag<PREC> (&)[1][2]>(ln[t]);
            f32x16 acc[1] = {bias_tile(cx, 3)};
            mma_cols<PREC, 2, 1>(s, 6, lt, acc);                      // [k head0 | k head1]
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) { d0 += qa[i][r] * acc[0][r]; d1 += qa[i][8 + r] * acc[0][8 + r]; }
                dot[i][0][t] = d0; dot[i][1][t] = d1;
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const BFrag<PREC> (&lt)[1][2] = reinterpret_cast<const BFrag<PREC> (&)[1][2]>(ln[t]);
            f32x16 acc[1] = {bias_tile(cx, 4)};
            mma_cols<PREC, 2, 1>(s, 8, lt, acc);                      // [k head2 | v head0]
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float d2 = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) d2 += qb[i][r] * acc[0][r];
                dot[i][2][t] = d2;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) v0[t][r] = acc[0][8 + r];
        }
        if constexpr (RING) advance(cx, 0);
        // softmax over the 3 keys, scale 16^-0.5 (renderer.py:956,971-973)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int hd = 0; hd < 3; ++hd) {
                float d[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) d[t] = xhalf_sum(dot[i][hd][t]) * 0.25f;
                float m = fmaxf(d[0], fmaxf(d[1], d[2]));
                float e0 = exp_(d[0] - m), e1 = exp_(d[1] - m), e2 = exp_(d[2] - m);
                float inv = rcp_(e0 + e1 + e2);
                dot[i][hd][0] = e0 * inv; dot[i][hd][1] = e1 * inv; dot[i][hd][2] = e2 * inv;
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) o[i][0][r] = dot[i][0][0] * v0[0][r] + dot[i][0][1] * v0[1][r] + dot[i][0][2] * v0[2][r];
        s = RING ? cx.slot(1) : cx.lds + step_pieces<PREC>(0) * 1024;     // step 1: chunks 5..8
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) { o[i][1][r] = 0.f; o[i][2][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const BFrag<PREC> (&lt)[1][2] = reinterpret_cast<const BFrag<PREC> (&)[1][2]>(ln[t]);
            f32x16 acc[1] = {bias_tile(cx, 5)};
            mma_cols<PREC, 2, 1>(s, 0, lt, acc);                      // [v head1 | v head2]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    o[i][1][r] += dot[i][1][t] * acc[0][r];
                    o[i][2][r] += dot[i][2][t] * acc[0][8 + r];
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
            mma_cols<PREC, 3, 2>(s, 2, ob, acc);                      // to_out + bias
            // residual (renderer.py:925).  tok[0], tok[1] are RE-READ (L2-hot, coalesced) instead of carried through the attention:
            // 32 registers less at the kernel's pressure peak
            if constexpr (PRE) {
                y[0] = acc[0] + tok[0]; y[1] = acc[1] + tok[1];
            } else {
                const float4* tp = tokens;
                asm volatile("" : "+v"(tp));
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 tk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 v = tp[((tile * 3 + t) * 8 + (2 * i + h)) * 32 + j];
                        tk[4 * i] = v.x; tk[4 * i + 1] = v.y; tk[4 * i + 2] = v.z; tk[4 * i + 3] = v.w;
                    }
                    y[t] = acc[t] + tk;
                }
            }
        }
        // ---- FF: LN2 -> Linear -> GELU(erf) -> Linear, residual (chunks 7, 8) ----
        {
            BFrag<PREC> l2[2][2];
            layer_norm<PREC>(cx, y[0], 1, l2[0][0], l2[0][1]);
            layer_norm<PREC>(cx, y[1], 1, l2[1][0], l2[1][1]);
            f32x16 acc[2] = {bias_tile(cx, 7), bias_tile(cx, 7)};
            mma_cols<PREC, 2, 2>(s, 5, l2, acc);
            BFrag<PREC> gb[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = gelu_<PREC != 1>(acc[i][r]);
                split_tile<PREC>(acc[i], gb[i][0], gb[i][1]);
            }
            f32x16 acc2[2] = {bias_tile(cx, 8), bias_tile(cx, 8)};
            mma_cols<PREC, 2, 2>(s, 7, gb, acc2);
            if constexpr (RING) advance(cx, 1);
            f32x16 za = acc2[0] + y[0], zb = acc2[1] + y[1];
            split_tile<PREC>(za, z0b[0], z0b[1]);
            split_tile<PREC>(zb, z1b[0], z1b[1]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// NeRF decoder (steps 2..42 of the weight stream through the ring) of one tile; on entry steps 2-4 have been issued and step 2 has landed
// (the caller's prologue, or the transformer's second advance()).
template <int PREC>
__device__ __forceinline__ void decoder_tile(Ctx<PREC>& cx, const int32_t* __restrict__ counters, const BFrag<PREC> (&z0b)[2], const BFrag<PREC> (&z1b)[2],
                                             const float (&xc)[3], const float (&vc)[3], int64_t tile, bool live, int64_t nv, float4* __restrict__ out) {
    const int j = cx.lane & 31, h = cx.h;
    // ================= NeRF decoder =================
    if constexpr (SHERF_MLP_DECODER_PRIO > 0) __builtin_amdgcn_s_setprio(SHERF_MLP_DECODER_PRIO);
    int step = 2;
    BFrag<PREC> ha[8], hb[8];
    AFrag<PREC> cur = load_units<PREC>(cx.slot(step));               // first fragments of the next step: fetched right behind its barrier
    // (the fetch goes out BEFORE the DMA issue of the slot just freed: its LDS latency hides under those ~30 instructions)
#define SHERF_NEXT_STEP() do { step_wait(cx, step); cur = load_units<PREC>(cx.slot(step + 1)); dma_issue(cx, step + NSLOT); ++step; } while (0)
    // a 128 -> 128 layer: two pairs of output chunks x two K-halves = four steps; chunk C0 + T -> OUT[2T], OUT[2T+1]
#define SHERF_LAYER128(C0, IN, OUT, RELU)                                                             \
    _Pragma("unroll") for (int P = 0; P < 2; ++P) {                                                   \
        f32x16 acc0 = bias_tile(cx, (C0) + 2 * P), acc1 = bias_tile(cx, (C0) + 2 * P + 1);            \
        mma_pair<PREC, 4>(cx.slot(step), 0, IN, acc0, acc1, cur);                                     \
        SHERF_NEXT_STEP();                                                                            \
        mma_pair<PREC, 4>(cx.slot(step), 0, IN + 4, acc0, acc1, cur);                                 \
        SHERF_NEXT_STEP();                                                                            \
        finish_pair<PREC, RELU>(acc0, acc1, OUT + 4 * P);                                             \
    }
    {   // pts_linears.0 : [PE6(x_c) (3 kb) | z_0 (2 kb)]
        BFrag<PREC> pe[3];
        pe_frags<PREC, 6, 3>(h, xc[0], xc[1], xc[2], pe);
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            f32x16 acc0 = bias_tile(cx, 9 + 2 * P), acc1 = bias_tile(cx, 10 + 2 * P);
            const char* s = cx.slot(step);
            mma_pair<PREC, 3, true>(s, 0, pe, acc0, acc1, cur);
            mma_pair<PREC, 2>(s, 6, z0b, acc0, acc1, cur);
            SHERF_NEXT_STEP();
            finish_pair<PREC, true>(acc0, acc1, ha + 4 * P);
        }
    }
    SHERF_LAYER128(13, ha, hb, true)     // pts_linears.1-4 (ping-pong ha -> hb -> ha ...)
    SHERF_LAYER128(17, hb, ha, true)
    SHERF_LAYER128(21, ha, hb, true)
    SHERF_LAYER128(25, hb, ha, true)
    {   // pts_linears.5 : [PE6 | z_0 | h(128)]: the encoding is recomputed (24 registers not carried through four layers)
        float x0 = xc[0], x1 = xc[1], x2 = xc[2];
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
        BFrag<PREC> pe[3];
        pe_frags<PREC, 6, 3>(h, x0, x1, x2, pe);
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            f32x16 acc0 = bias_tile(cx, 29 + 2 * P), acc1 = bias_tile(cx, 30 + 2 * P);
            const char* s = cx.slot(step);
            mma_pair<PREC, 3, true>(s, 0, pe, acc0, acc1, cur);
            mma_pair<PREC, 2>(s, 6, z0b, acc0, acc1, cur);
            SHERF_NEXT_STEP();
            mma_pair<PREC, 4>(cx.slot(step), 0, ha, acc0, acc1, cur);
            SHERF_NEXT_STEP();
            mma_pair<PREC, 4>(cx.slot(step), 0, ha + 4, acc0, acc1, cur);
            SHERF_NEXT_STEP();
            finish_pair<PREC, true>(acc0, acc1, hb + 4 * P);
        }
    }
    SHERF_LAYER128(33, hb, ha, true)     // pts_linears.6
    SHERF_LAYER128(37, ha, hb, true)     // pts_linears.7
    // ---- heads: feature_linear (no activation) into ha, alpha_linear (chunk 45, row 0), both from hb ----
    SHERF_LAYER128(41, hb, ha, false)
    float sigma;
    {
        f32x16 acc0 = bias_tile(cx, 45), acc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        mma_splitk<PREC, 8>(cx.slot(step), 0, hb, acc0, acc1, cur);
        SHERF_NEXT_STEP();
        mfma_settle(acc0, acc1);
        sigma = acc0[0] + acc1[0];                                   // row 0 lives in reg 0 of the h == 0 lanes
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- views_linear : [feature (8 kb) | PE4(v_c) (2 kb) | z_1 (2 kb)] -> 64, ReLU ; rgb_linear -> sigmoid ----
    BFrag<PREC> gb[4];
    {
        BFrag<PREC> pv[2];
        pe_frags<PREC, 4, 2>(h, vc[0], vc[1], vc[2], pv);
        f32x16 acc0 = bias_tile(cx, 46), acc1 = bias_tile(cx, 47);
        mma_pair<PREC, 4>(cx.slot(step), 0, ha, acc0, acc1, cur);
        SHERF_NEXT_STEP();
        mma_pair<PREC, 4>(cx.slot(step), 0, ha + 4, acc0, acc1, cur);
        SHERF_NEXT_STEP();
        const char* s = cx.slot(step);
        mma_pair<PREC, 2, true>(s, 0, pv, acc0, acc1, cur);
        mma_pair<PREC, 2>(s, 4, z1b, acc0, acc1, cur);
        SHERF_NEXT_STEP();
        finish_pair<PREC, true>(acc0, acc1, gb);
    }
    {
        f32x16 acc0 = bias_tile(cx, 48), acc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        mma_splitk<PREC, 4>(cx.slot(step), 0, gb, acc0, acc1, cur);
        mfma_settle(acc0, acc1);
        SHERF_TRACE_STAMP(cx, step, 0);
        if (live && h == 0) {
            const int64_t c = tile * 32 + j;
            if (c < nv) {
                float r = rcp_(1.0f + exp_(-(acc0[0] + acc1[0]))), g = rcp_(1.0f + exp_(-(acc0[1] + acc1[1]))), b = rcp_(1.0f + exp_(-(acc0[2] + acc1[2])));
                out[c] = make_float4(r * 1.002f - 0.001f, g * 1.002f - 0.001f, b * 1.002f - 0.001f, sigma);   // triplane.py:314
                // fp16 operand range guard (prec 1, 2): an activation beyond 65504 turns into inf - inf = NaN and reaches the outputs;
                // counters[3] (reset by the sampler every frame) then reads 1: sherf_amd.ImportanceRenderer.check_finite()
                if (!(fabsf(sigma) <= 3.0e38f) || !(r + g + b <= 4.0f)) atomicOr(reinterpret_cast<unsigned*>(const_cast<int32_t*>(counters)) + 3, 1u);
            }
        }
    }
#undef SHERF_NEXT_STEP
#undef SHERF_LAYER128
}

// =====================================================================================================================================
// Round 5: the decoder with its epilogues INSIDE the MFMA stream (single-product precisions, one tile per wave).
// What the in-kernel timeline of this round says (profiles/r05_call_c_*): a decoder step that only issues MFMAs runs at ~70 % of the matrix
// pipe, a step that also carries a pair's epilogue (fp32 -> fp16 repack + ReLU of 32 accumulator registers, the next pair's bias reads) takes
// ~620 cycles more -- during which the pipe idles: the co-resident workgroup is in its VALU-bound transformer then (the start offsets of
// co-resident workgroups sit at half a period) and has no MFMAs to offer.  The microbenchmark (tools/ubench/issue_model.hip) shows where such
// VALU work is free: a handful of instructions between two MFMAs of the SAME wave (8 MFMAs + 24 / 40 VALU: 35 / 36 cycles per MFMA against
// 33.5 bare).  So here a finished pair of accumulator tiles is NOT converted at once: the next ring step runs on a SECOND pair of accumulators
// and carries the conversion as four 8-instruction pieces pinned behind its four MFMA blocks; the piece order never crosses a dependency
// (a layer's K-blocks 0-3 come from the pair finished two steps earlier, 4-7 from the one finished one step earlier, and the weight stream
// already walks [pair 0: K lo, K hi][pair 1: K lo, K hi]); the bias tiles of a pair are read one step ahead, into the accumulators the
// previous conversion has just released.  Same MFMA order per accumulator, same conversions: bit-identical to decoder_tile.
template <int PREC, bool RELU>
__device__ __forceinline__ void epi_piece(const f32x16& a0, const f32x16& a1, BFrag<PREC>* out, int i) {
    static_assert(PREC != 1, "single-product precisions");
    if constexpr ((SHERF_MLP_ABLATE & 2048) != 0) return;
    const f32x16& a = i < 2 ? a0 : a1;
    const int o = (i & 1) * 8;
    constexpr bool PK = RELU && PREC == 2 && SHERF_MLP_PK_RELU;
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (RELU && !PK) ? relu(a[o + r]) : a[o + r];
    out[i] = make_frag<PREC>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    if constexpr (PK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[i].hi[e] = relu2_f16(out[i].hi[e]);
    }
}
// mma_chains with a filler pinned behind every block: fill(f0 + i) runs between block i and block i + 1
template <int PREC, int NB, bool PAIR, bool MORE, class F>
__device__ __forceinline__ void mma_chains_f(const char* s, int u0, const BFrag<PREC>* b, f32x16& acc0, f32x16& acc1, AFrag<PREC>& cur, int f0, F&& fill) {
    constexpr int UNIT = Ctx<PREC>::UNIT;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        AFrag<PREC> nxt;
        const bool pre = i + 1 < NB || MORE;
        if (pre) nxt = load_units<PREC>(s + (u0 + 2 * (i + 1)) * UNIT);
        const BFrag<PREC>& b0 = b[PAIR ? i : 2 * i];
        const BFrag<PREC>& b1 = b[PAIR ? i : 2 * i + 1];
        mfma_block<PREC>(acc0, acc1, cur.h0, cur.h0, cur.h1, cur.h1, b0.hi, b0.hi, b1.hi, b1.hi);
        __builtin_amdgcn_sched_barrier(0);
        fill(f0 + i);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) cur = nxt;
    }
}

// PEM = true (round 6, prec 2): PE6(x_c) / PE4(v_c) are LOADED as fp16 fragments (pef = pefrag + this lane: [tile][q = 0-2 | 3-4][lane]) where
// pe_frags computed them -- ~420 VALU per tile less in a kernel whose time follows its instruction energy.  The loads are
// compiler-visible: their s_waitcnt can only be STRONGER than needed beside the asm-issued weight DMA (vmcnt retires in order).
template <int PREC, bool PEM = false>
__device__ __forceinline__ void decoder_tile_p(Ctx<PREC>& cx, const int32_t* __restrict__ counters, const BFrag<PREC> (&z0b)[2], const BFrag<PREC> (&z1b)[2],
                                               const float (&xc)[3], const float (&vc)[3], int64_t tile, bool live, int64_t nv, float4* __restrict__ out,
                                               const u32x4* __restrict__ pef = nullptr) {
    const int j = cx.lane & 31, h = cx.h;
    if constexpr (SHERF_MLP_DECODER_PRIO > 0) __builtin_amdgcn_s_setprio(SHERF_MLP_DECODER_PRIO);
    int step = 2;
    BFrag<PREC> ha[8], hb[8];
    if constexpr ((SHERF_MLP_ABLATE & 2048) != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { ha[i] = z0b[i & 1]; hb[i] = z1b[i & 1]; }
    }
    f32x16 X0, X1, Y0, Y1;                                           // the two accumulator pairs
    AFrag<PREC> cur = load_units<PREC>(cx.slot(step));
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto none = [](int) {};
#define SHERF_NEXT_STEP() do { step_wait(cx, step); cur = load_units<PREC>(cx.slot(step + 1)); dma_issue(cx, step + NSLOT); ++step; } while (0)
    // a 128 -> 128 layer (chunks C0..C0+3, IN -> OUT): pair 0 on X, pair 1 on Y.  On entry X holds pair 0's bias tiles and Y the previous
    // layer's finished pair 1, whose conversion (-> IN[4..7], PREV_RELU) rides on the first step; on exit Y holds this layer's finished
    // pair 1 and X the bias tiles NX0, NX1 (-1: zeros) of whatever runs on X next.
#define SHERF_LAYER128_P(C0, IN, OUT, RELU, PREV_RELU, NX0, NX1)                                                                              \
    mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, IN, X0, X1, cur, 0, [&](int i) { epi_piece<PREC, PREV_RELU>(Y0, Y1, IN + 4, i); }); \
    SHERF_NEXT_STEP();                                                                                                                         \
    mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, IN + 4, X0, X1, cur, 0, [&](int i) { if (i == 0) { Y0 = bias_tile(cx, (C0) + 2); Y1 = bias_tile(cx, (C0) + 3); } }); \
    SHERF_NEXT_STEP();                                                                                                                         \
    mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, IN, Y0, Y1, cur, 0, [&](int i) { epi_piece<PREC, RELU>(X0, X1, OUT, i); });           \
    SHERF_NEXT_STEP();                                                                                                                         \
    mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, IN + 4, Y0, Y1, cur, 0, [&](int i) { if (i == 0) { X0 = bias_tile(cx, NX0); X1 = (NX1) < 0 ? zero : bias_tile(cx, (NX1) < 0 ? 0 : (NX1)); } }); \
    SHERF_NEXT_STEP();
    {   // pts_linears.0 : [PE6(x_c) (3 kb) | z_0 (2 kb)], one step per pair
        BFrag<PREC> pe[3];
        if constexpr (PEM) {
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) pe[kb].hi = pef[(tile * 7 + kb) * 64];
        } else pe_frags<PREC, 6, 3>(h, xc[0], xc[1], xc[2], pe);
        X0 = bias_tile(cx, 9); X1 = bias_tile(cx, 10);
        const char* s = cx.slot(step);
        mma_chains_f<PREC, 3, true, true>(s, 0, pe, X0, X1, cur, 0, [&](int i) { if (i == 0) { Y0 = bias_tile(cx, 11); Y1 = bias_tile(cx, 12); } });
        mma_chains_f<PREC, 2, true, false>(s, 6, z0b, X0, X1, cur, 3, none);
        SHERF_NEXT_STEP();
        s = cx.slot(step);
        auto carry = [&](int i) { if (i < 4) epi_piece<PREC, true>(X0, X1, ha, i); else { X0 = bias_tile(cx, 13); X1 = bias_tile(cx, 14); } };
        mma_chains_f<PREC, 3, true, true>(s, 0, pe, Y0, Y1, cur, 0, carry);
        mma_chains_f<PREC, 2, true, false>(s, 6, z0b, Y0, Y1, cur, 3, carry);
        SHERF_NEXT_STEP();
    }
    SHERF_LAYER128_P(13, ha, hb, true, true, 17, 18)      // pts_linears.1-4
    SHERF_LAYER128_P(17, hb, ha, true, true, 21, 22)
    SHERF_LAYER128_P(21, ha, hb, true, true, 25, 26)
    SHERF_LAYER128_P(25, hb, ha, true, true, 29, 30)
    {   // pts_linears.5 : [PE6 | z_0 | h(128)], three steps per pair; on entry Y = pts_linears.4's pair 1 (-> ha[4..7])
        float x0 = xc[0], x1 = xc[1], x2 = xc[2];
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
        BFrag<PREC> pe[3];
        if constexpr (PEM) {                                         // re-read (L2): 12 registers not carried through four layers
            const u32x4* pp = pef;
            asm volatile("" : "+v"(pp));
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) pe[kb].hi = pp[(tile * 7 + kb) * 64];
        } else pe_frags<PREC, 6, 3>(h, x0, x1, x2, pe);
        const char* s = cx.slot(step);
        auto carry0 = [&](int i) { if (i < 4) epi_piece<PREC, true>(Y0, Y1, ha + 4, i); };
        mma_chains_f<PREC, 3, true, true>(s, 0, pe, X0, X1, cur, 0, carry0);
        mma_chains_f<PREC, 2, true, false>(s, 6, z0b, X0, X1, cur, 3, carry0);
        SHERF_NEXT_STEP();
        mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, ha, X0, X1, cur, 0, [&](int i) { if (i == 0) { Y0 = bias_tile(cx, 31); Y1 = bias_tile(cx, 32); } });
        SHERF_NEXT_STEP();
        mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, ha + 4, X0, X1, cur, 0, none);
        SHERF_NEXT_STEP();
        s = cx.slot(step);
        auto carry1 = [&](int i) { if (i < 4) epi_piece<PREC, true>(X0, X1, hb, i); };
        mma_chains_f<PREC, 3, true, true>(s, 0, pe, Y0, Y1, cur, 0, carry1);
        mma_chains_f<PREC, 2, true, false>(s, 6, z0b, Y0, Y1, cur, 3, carry1);
        SHERF_NEXT_STEP();
        mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, ha, Y0, Y1, cur, 0, [&](int i) { if (i == 0) { X0 = bias_tile(cx, 33); X1 = bias_tile(cx, 34); } });
        SHERF_NEXT_STEP();
        mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, ha + 4, Y0, Y1, cur, 0, none);
        SHERF_NEXT_STEP();
    }
    SHERF_LAYER128_P(33, hb, ha, true, true, 37, 38)      // pts_linears.6
    SHERF_LAYER128_P(37, ha, hb, true, true, 41, 42)      // pts_linears.7
    SHERF_LAYER128_P(41, hb, ha, false, true, 45, -1)     // feature_linear (no activation) -> ha; X <- alpha_linear's bias / zeros
    float sigma;
    BFrag<PREC> gb[4];
    {
        // alpha_linear (chunk 45, split-K over hb) on X; carries feature_linear's pair 1 (Y -> ha[4..7]), then reads views_linear's bias tiles into Y
        mma_chains_f<PREC, 4, false, false>(cx.slot(step), 0, hb, X0, X1, cur, 0, [&](int i) {
            epi_piece<PREC, false>(Y0, Y1, ha + 4, i);
            if (i == 3) { Y0 = bias_tile(cx, 46); Y1 = bias_tile(cx, 47); }
        });
        SHERF_NEXT_STEP();
        // views_linear : [feature (8 kb) | PE4(v_c) (2 kb) | z_1 (2 kb)] -> 64, ReLU, on Y; sigma leaves X behind the first block
        BFrag<PREC> pv[2];
        if constexpr (PEM) { pv[0].hi = pef[(tile * 7 + 3) * 64]; pv[1].hi = pef[(tile * 7 + 4) * 64]; }
        else pe_frags<PREC, 4, 2>(h, vc[0], vc[1], vc[2], pv);
        mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, ha, Y0, Y1, cur, 0, [&](int i) {
            // (the WHOLE tuples stay live up to here: only element 0 of each is read, and hipcc would hand the other fifteen registers of an
            //  accumulator to the bias reads above while the MFMA writing them is still in flight -- tools/mfma_hazard_check.py, R1)
            if (i == 0) { asm volatile("" : "+v"(X0), "+v"(X1)); sigma = X0[0] + X1[0]; }               // row 0: reg 0 of the h == 0 lanes
        });
        SHERF_NEXT_STEP();
        mma_chains_f<PREC, 4, true, false>(cx.slot(step), 0, ha + 4, Y0, Y1, cur, 0, [&](int i) { if (i == 0) { X0 = bias_tile(cx, 48); X1 = zero; } });
        SHERF_NEXT_STEP();
        const char* s = cx.slot(step);
        mma_chains_f<PREC, 2, true, true>(s, 0, pv, Y0, Y1, cur, 0, none);
        mma_chains_f<PREC, 2, true, false>(s, 4, z1b, Y0, Y1, cur, 2, none);
        SHERF_NEXT_STEP();
        if constexpr ((SHERF_MLP_ABLATE & 2048) != 0) { mfma_settle(Y0, Y1); gb[0] = z0b[0]; gb[1] = z0b[1]; gb[2] = z1b[0]; gb[3] = z1b[1]; }
        else
        finish_pair<PREC, true>(Y0, Y1, gb);                         // rgb_linear needs all of it: nothing to ride on
    }
    {
        mma_chains_f<PREC, 2, false, false>(cx.slot(step), 0, gb, X0, X1, cur, 0, none);
        mfma_settle(X0, X1);
        SHERF_TRACE_STAMP(cx, step, 0);
        if (live && h == 0) {
            const int64_t c = tile * 32 + j;
            if (c < nv) {
                float r = rcp_(1.0f + exp_(-(X0[0] + X1[0]))), g = rcp_(1.0f + exp_(-(X0[1] + X1[1]))), b = rcp_(1.0f + exp_(-(X0[2] + X1[2])));
                out[c] = make_float4(r * 1.002f - 0.001f, g * 1.002f - 0.001f, b * 1.002f - 0.001f, sigma);   // triplane.py:314
                if (!(fabsf(sigma) <= 3.0e38f) || !(r + g + b <= 4.0f)) atomicOr(reinterpret_cast<unsigned*>(const_cast<int32_t*>(counters)) + 3, 1u);
            }
        }
    }
#undef SHERF_NEXT_STEP
#undef SHERF_LAYER128_P
}

// =====================================================================================================================================
// Round 5: TWO 32-sample tiles per wave (single-product precisions).  Why: round 4's counters and timeline say the one-tile kernel sits
// at the SUM of its issue streams -- per tile a wave issues 374 MFMAs beside ~4 800 other instructions, of which ~1 950 are per-STEP
// overhead of the weight ring (the A-fragment ds_reads and their waits, the DMA issue, the barrier, the block-opening s_nops) paid once
// per 8 MFMAs.  With two tiles in a wave every A fragment read from LDS feeds TWO MFMAs, a step carries 16 MFMAs in FOUR independent
// accumulator chains (consecutive MFMAs of a chain are three MFMAs = 96 pipe cycles apart: nothing the compiler drops between blocks
// sits between dependent MFMAs any more), and ring traffic, barriers and waits per tile halve.  Registers: two tiles' activations
// (2 x 64) + four accumulators (64) + fragments need the 256-register budget = two waves per SIMD (two 4-wave workgroups per CU), each
// holding two tiles: four tiles in flight per SIMD instead of three.  The transformer runs tile after tile with its weights (steps 0-1)
// resident in ring slots 0-1 -- no barrier, no DMA wait while it runs -- and parks the fused tokens z_0 / z_1 as B fragments in LDS.
// Same MFMA order per accumulator as the one-tile kernel: bit-identical outputs.
template <int PREC>
__device__ __forceinline__ void mfma_block4(f32x16& t0c0, f32x16& t0c1, f32x16& t1c0, f32x16& t1c1, const u32x4& a0, const u32x4& a1,
                                            const u32x4& b00, const u32x4& b01, const u32x4& b10, const u32x4& b11) {
    // chain (tile t, unit u) <- A fragment a_u x B fragment b_tu; issue order t0u0, t1u0, t0u1, t1u1
    static_assert(PREC != 1, "two tiles per wave: single-product precisions only");
    if constexpr (PREC == 2)
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %4, %6, %0\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %4, %8, %2\n\t"
                     "v_mfma_f32_32x32x16_f16 %1, %5, %7, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %3, %5, %9, %3"
                     : "+v"(t0c0), "+v"(t0c1), "+v"(t1c0), "+v"(t1c1) : "v"(a0), "v"(a1), "v"(b00), "v"(b01), "v"(b10), "v"(b11) : "memory");
    else
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %4, %6, %0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %2, %4, %8, %2\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %5, %7, %1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %3, %5, %9, %3"
                     : "+v"(t0c0), "+v"(t0c1), "+v"(t1c0), "+v"(t1c1) : "v"(a0), "v"(a1), "v"(b00), "v"(b01), "v"(b10), "v"(b11) : "memory");
}
__device__ __forceinline__ void mfma_settle4(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
    asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// as mma_chains, both tiles at once: acc[t][u]; PAIR: chain (t, u) takes b_t[i]; split-K: chain (t, u) takes b_t[2 i + u]
template <int PREC, int NB, bool PAIR, bool MORE>
__device__ __forceinline__ void mma_chains2(const char* s, int u0, const BFrag<PREC>* b0, const BFrag<PREC>* b1, f32x16 (&acc)[2][2], AFrag<PREC>& cur) {
    constexpr int UNIT = Ctx<PREC>::UNIT;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        AFrag<PREC> nxt;
        const bool pre = i + 1 < NB || MORE;
        if (pre) nxt = load_units<PREC>(s + (u0 + 2 * (i + 1)) * UNIT);
        const int k0 = PAIR ? i : 2 * i, k1 = PAIR ? i : 2 * i + 1;
        mfma_block4<PREC>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], cur.h0, cur.h1, b0[k0].hi, b0[k1].hi, b1[k0].hi, b1[k1].hi);
        if (pre) cur = nxt;
    }
}
// four finished accumulator tiles (a pair of chunks x two tiles) -> four K-blocks of the next layer per tile
//   SETTLE = false: the caller vouches for >= 12 issued instructions between the last MFMA and this point (a SHERF_NEXT_STEP: waits, barrier,
//   fragment reads, DMA issue) -- tools/mfma_hazard_check.py (tests/test_isa_hazards.py) checks exactly that on the final ISA
template <int PREC, bool RELU, bool SETTLE = true>
__device__ __forceinline__ void finish_quad(f32x16 (&acc)[2][2], BFrag<PREC>* out0, BFrag<PREC>* out1) {
    if constexpr (SETTLE) mfma_settle4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
    else __builtin_amdgcn_sched_barrier(0);
    constexpr bool PK = RELU && PREC == 2 && SHERF_MLP_PK_RELU;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        BFrag<PREC>* out = t ? out1 : out0;
        if constexpr (RELU && !PK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[t][0][r] = relu(acc[t][0][r]); acc[t][1][r] = relu(acc[t][1][r]); }
        }
        split_tile<PREC>(acc[t][0], out[0], out[1]);
        split_tile<PREC>(acc[t][1], out[2], out[3]);
        if constexpr (PK) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) out[f].hi[e] = relu2_f16(out[f].hi[e]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// NeRF decoder of TWO tiles (steps 2..42 through the ring); zst = this wave's LDS park of the fused tokens (+ this lane's 16 bytes):
// fragment q of tile t at zst + (4 t + q) KiB, q = 0, 1: z_0's K-blocks, 2, 3: z_1's.  Entry state as decoder_tile.
template <int PREC>
__device__ __forceinline__ void decoder_tile2(Ctx<PREC>& cx, const int32_t* __restrict__ counters, const char* zst, const float (&xc)[2][3],
                                              const float (&vc)[2][3], const int64_t (&tile)[2], const bool (&live)[2], int64_t nv, float4* __restrict__ out) {
    const int j = cx.lane & 31, h = cx.h;
    if constexpr (SHERF_MLP_DECODER_PRIO > 0) __builtin_amdgcn_s_setprio(SHERF_MLP_DECODER_PRIO);
    int step = 2;
    BFrag<PREC> ha[2][8], hb[2][8];
    AFrag<PREC> cur = load_units<PREC>(cx.slot(step));
    auto zfrag = [&](int t, int q) { BFrag<PREC> f; f.hi = *reinterpret_cast<const u32x4*>(zst + (4 * t + q) * 1024); return f; };
#define SHERF_NEXT_STEP() do { step_wait(cx, step); cur = load_units<PREC>(cx.slot(step + 1)); dma_issue(cx, step + NSLOT); ++step; } while (0)
    // (tile 1's copy of the bias tables through a laundered offset: otherwise the compiler reads each table once and COPIES 16 registers)
    Ctx<PREC> cx1 = cx;
    { uint32_t boff = 0; asm volatile("" : "+v"(boff)); cx1.wbias = cx.wbias + boff; }
#define SHERF_BIAS4(ACC, C0) do { ACC[0][0] = bias_tile(cx, (C0)); ACC[0][1] = bias_tile(cx, (C0) + 1); ACC[1][0] = bias_tile(cx1, (C0)); ACC[1][1] = bias_tile(cx1, (C0) + 1); } while (0)
#define SHERF_LAYER128_2(C0, IN, OUT, RELU)                                                           \
    _Pragma("unroll") for (int P = 0; P < 2; ++P) {                                                   \
        f32x16 acc[2][2];                                                                             \
        SHERF_BIAS4(acc, (C0) + 2 * P);                                                               \
        mma_chains2<PREC, 4, true, false>(cx.slot(step), 0, IN[0], IN[1], acc, cur);                  \
        SHERF_NEXT_STEP();                                                                            \
        mma_chains2<PREC, 4, true, false>(cx.slot(step), 0, IN[0] + 4, IN[1] + 4, acc, cur);          \
        SHERF_NEXT_STEP();                                                                            \
        finish_quad<PREC, RELU, false>(acc, OUT[0] + 4 * P, OUT[1] + 4 * P);                          \
    }
    {   // pts_linears.0 : [PE6(x_c) (3 kb) | z_0 (2 kb)]
        BFrag<PREC> pe[2][3];
        pe_frags<PREC, 6, 3>(h, xc[0][0], xc[0][1], xc[0][2], pe[0]);
        pe_frags<PREC, 6, 3>(h, xc[1][0], xc[1][1], xc[1][2], pe[1]);
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            f32x16 acc[2][2];
            SHERF_BIAS4(acc, 9 + 2 * P);
            const char* s = cx.slot(step);
            mma_chains2<PREC, 3, true, true>(s, 0, pe[0], pe[1], acc, cur);
            const BFrag<PREC> z0[2][2] = {{zfrag(0, 0), zfrag(0, 1)}, {zfrag(1, 0), zfrag(1, 1)}};
            mma_chains2<PREC, 2, true, false>(s, 6, z0[0], z0[1], acc, cur);
            SHERF_NEXT_STEP();
            finish_quad<PREC, true, false>(acc, ha[0] + 4 * P, ha[1] + 4 * P);
        }
    }
    SHERF_LAYER128_2(13, ha, hb, true)     // pts_linears.1-4
    SHERF_LAYER128_2(17, hb, ha, true)
    SHERF_LAYER128_2(21, ha, hb, true)
    SHERF_LAYER128_2(25, hb, ha, true)
    {   // pts_linears.5 : [PE6 | z_0 | h(128)]
        float x0[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            x0[t][0] = xc[t][0]; x0[t][1] = xc[t][1]; x0[t][2] = xc[t][2];
            asm volatile("" : "+v"(x0[t][0]), "+v"(x0[t][1]), "+v"(x0[t][2]));
        }
        BFrag<PREC> pe[2][3];
        pe_frags<PREC, 6, 3>(h, x0[0][0], x0[0][1], x0[0][2], pe[0]);
        pe_frags<PREC, 6, 3>(h, x0[1][0], x0[1][1], x0[1][2], pe[1]);
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            f32x16 acc[2][2];
            SHERF_BIAS4(acc, 29 + 2 * P);
            const char* s = cx.slot(step);
            mma_chains2<PREC, 3, true, true>(s, 0, pe[0], pe[1], acc, cur);
            {
                const BFrag<PREC> z0[2][2] = {{zfrag(0, 0), zfrag(0, 1)}, {zfrag(1, 0), zfrag(1, 1)}};
                mma_chains2<PREC, 2, true, false>(s, 6, z0[0], z0[1], acc, cur);
            }
            SHERF_NEXT_STEP();
            mma_chains2<PREC, 4, true, false>(cx.slot(step), 0, ha[0], ha[1], acc, cur);
            SHERF_NEXT_STEP();
            mma_chains2<PREC, 4, true, false>(cx.slot(step), 0, ha[0] + 4, ha[1] + 4, acc, cur);
            SHERF_NEXT_STEP();
            finish_quad<PREC, true, false>(acc, hb[0] + 4 * P, hb[1] + 4 * P);
        }
    }
    SHERF_LAYER128_2(33, hb, ha, true)     // pts_linears.6
    SHERF_LAYER128_2(37, ha, hb, true)     // pts_linears.7
    SHERF_LAYER128_2(41, hb, ha, false)    // feature_linear -> ha
    float sigma[2];
    {   // alpha_linear (chunk 45, split-K over hb)
        f32x16 acc[2][2];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[0][0] = bias_tile(cx, 45); acc[1][0] = bias_tile(cx1, 45); acc[0][1] = zero; acc[1][1] = zero;
        mma_chains2<PREC, 4, false, false>(cx.slot(step), 0, hb[0], hb[1], acc, cur);
        SHERF_NEXT_STEP();
        mfma_settle4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        sigma[0] = acc[0][0][0] + acc[0][1][0];
        sigma[1] = acc[1][0][0] + acc[1][1][0];
        __builtin_amdgcn_sched_barrier(0);
    }
    BFrag<PREC> gb[2][4];
    {   // views_linear : [feature (8 kb) | PE4(v_c) (2 kb) | z_1 (2 kb)] -> 64, ReLU
        BFrag<PREC> pv[2][2];
        pe_frags<PREC, 4, 2>(h, vc[0][0], vc[0][1], vc[0][2], pv[0]);
        pe_frags<PREC, 4, 2>(h, vc[1][0], vc[1][1], vc[1][2], pv[1]);
        f32x16 acc[2][2];
        SHERF_BIAS4(acc, 46);
        mma_chains2<PREC, 4, true, false>(cx.slot(step), 0, ha[0], ha[1], acc, cur);
        SHERF_NEXT_STEP();
        mma_chains2<PREC, 4, true, false>(cx.slot(step), 0, ha[0] + 4, ha[1] + 4, acc, cur);
        SHERF_NEXT_STEP();
        const char* s = cx.slot(step);
        mma_chains2<PREC, 2, true, true>(s, 0, pv[0], pv[1], acc, cur);
        const BFrag<PREC> z1[2][2] = {{zfrag(0, 2), zfrag(0, 3)}, {zfrag(1, 2), zfrag(1, 3)}};
        mma_chains2<PREC, 2, true, false>(s, 4, z1[0], z1[1], acc, cur);
        SHERF_NEXT_STEP();
        finish_quad<PREC, true>(acc, gb[0], gb[1]);          // (no DMA issue behind the last steps: the settle stays)
    }
    {   // rgb_linear (chunk 48, split-K over gb) -> sigmoid
        f32x16 acc[2][2];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[0][0] = bias_tile(cx, 48); acc[1][0] = bias_tile(cx1, 48); acc[0][1] = zero; acc[1][1] = zero;
        mma_chains2<PREC, 2, false, false>(cx.slot(step), 0, gb[0], gb[1], acc, cur);
        mfma_settle4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        SHERF_TRACE_STAMP(cx, step, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (live[t] && h == 0) {
                const int64_t c = tile[t] * 32 + j;
                if (c < nv) {
                    float r = rcp_(1.0f + exp_(-(acc[t][0][0] + acc[t][1][0]))), g = rcp_(1.0f + exp_(-(acc[t][0][1] + acc[t][1][1]))),
                          b = rcp_(1.0f + exp_(-(acc[t][0][2] + acc[t][1][2])));
                    out[c] = make_float4(r * 1.002f - 0.001f, g * 1.002f - 0.001f, b * 1.002f - 0.001f, sigma[t]);   // triplane.py:314
                    if (!(fabsf(sigma[t]) <= 3.0e38f) || !(r + g + b <= 4.0f)) atomicOr(reinterpret_cast<unsigned*>(const_cast<int32_t*>(counters)) + 3, 1u);
                }
            }
        }
    }
#undef SHERF_NEXT_STEP
#undef SHERF_BIAS4
#undef SHERF_LAYER128_2
}

#ifndef SHERF_MLP_LB
#define SHERF_MLP_LB 2            // minimum waves / SIMD the single-product instances are compiled for (register cap 512 / LB)
#endif
// ring prologue shared by the fused kernel (S0 = 0) and the decoder kernel (S0 = 2): steps S0, S0+1 issued, S0 landed, S0+2 issued
template <int PREC, int S0>
__device__ __forceinline__ void ring_prologue(Ctx<PREC>& cx) {
    dma_issue(cx, S0);
    dma_issue(cx, S0 + 1);
    wait_vm((SHERF_MLP_ABLATE & 32) ? 0 : step_pieces<PREC>(S0 + 1) / NW);   // step S0 (this wave's pieces) landed
    __syncthreads();
    dma_issue(cx, S0 + 2);
}
template <int PREC>
__device__ __forceinline__ void ring_ctx(Ctx<PREC>& cx, char* lds, const char* ws, const float* wbias, bool lds_trace = true) {
    using CX = Ctx<PREC>;
    constexpr int NT = NW * 64;
    float* lbias = reinterpret_cast<float*>(lds + NSLOT * CX::SLOT);
    for (int i = threadIdx.x; i < (N_CHUNKS + 4) * 32; i += NT) lbias[i] = wbias[i];   // visible after the prologue barrier
    cx.lane = threadIdx.x & 63; cx.h = cx.lane >> 5; cx.notrans = 0;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cx.ws = ws + cx.wave * 1024 + cx.lane * 16; cx.wbias = lbias; cx.lds = lds + cx.lane * 16;
    cx.lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)lds) + cx.wave * 1024;
#if SHERF_MLP_TRACE
    cx.trace = reinterpret_cast<uint32_t*>(lds + NSLOT * CX::SLOT + (N_CHUNKS + 4) * 32 * 4) + cx.wave * 256;
    cx.treg = !lds_trace; cx.tv[0] = cx.tv[1] = cx.tv[2] = 0;
    if (lds_trace && cx.lane == 0) {
        cx.trace[63 * 4] = (uint32_t)__builtin_amdgcn_s_memtime();
        cx.trace[63 * 4 + 1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_ID: wave slot, SIMD, CU, SH, SE
        cx.trace[63 * 4 + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // XCC_ID
    }
#endif
}
#if SHERF_MLP_TRACE
#define SHERF_TRACE_FLUSH(cx) do {                                                                           \
    if (g_mlp_trace && g_mlp_trace_every > 0 && blockIdx.x % g_mlp_trace_every == 0) {                       \
        if ((cx).lane == 0) (cx).trace[63 * 4 + 2] = (uint32_t)__builtin_amdgcn_s_memtime();                 \
        const size_t slot = blockIdx.x / g_mlp_trace_every;                                                  \
        uint32_t* dst = g_mlp_trace + (slot * 8 + (cx).wave) * 256;                                          \
        for (int i = (cx).lane; i < 256; i += 64) dst[i] = (cx).trace[i];                                    \
    } } while (0)
#else
#define SHERF_TRACE_FLUSH(cx) do { } while (0)
#endif

template <int PREC>
__global__ void __launch_bounds__(NW * 64, PREC == 1 ? 2 : SHERF_MLP_LB)
nerf_mlp_kernel(const int32_t* __restrict__ counters, const float4* __restrict__ tokens, const float* __restrict__ extras,
                const char* __restrict__ ws, const float* __restrict__ wbias, int64_t capacity, float4* __restrict__ out, int part, int nparts, int notrans) {
    using CX = Ctx<PREC>;
    __shared__ __attribute__((aligned(16))) char lds[NSLOT * CX::SLOT + (N_CHUNKS + 4) * 32 * 4 + (SHERF_MLP_TRACE ? NW * 64 * 16 : 0)];
    const int64_t nv = min((int64_t)counters[0], capacity);
    int64_t t_lo, n_tiles;                                           // this launch's part of the tiles (sherf_nerf_mlp_part; whole: 0, n)
    sherf_part_range((nv + 31) / 32, part, nparts, t_lo, n_tiles);
    if (t_lo + (int64_t)blockIdx.x * NW >= n_tiles) return;          // whole workgroup beyond the data
    CX cx;
    ring_ctx<PREC>(cx, lds, ws, wbias);
    cx.notrans = notrans;
    int64_t tile = t_lo + (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    const bool live = tile < n_tiles;
    if (!live) tile = n_tiles - 1;                                   // dead tiles still take part in every barrier
    ring_prologue<PREC, 0>(cx);

    BFrag<PREC> z0b[2], z1b[2];                                      // fused tokens z_0, z_1 as K-blocks
    float xc[3], vc[3];
    const float* ex = extras + tile * 12 * 32 + (cx.lane & 31);
    xc[0] = ex[0]; xc[1] = ex[32]; xc[2] = ex[64]; vc[0] = ex[96]; vc[1] = ex[128]; vc[2] = ex[160];
    {
        f32x16 tok[3];
        transformer_tile<PREC, true>(cx, tokens, extras, tile, z0b, z1b, tok);
    }
    decoder_tile<PREC>(cx, counters, z0b, z1b, xc, vc, tile, live, nv, out);
    SHERF_TRACE_FLUSH(cx);
}

// One launch, two tiles per wave (round 5; single-product precisions).  A 4-wave workgroup owns EIGHT tiles; LDS = the 3-slot ring + the bias
// tables + 8 KiB per wave for the parked tokens (74.8 KiB: two workgroups per CU).  Schedule: ring prologue (steps 0, 1, 2 in slots 0, 1,
// 2) -> the transformer of tile 0, then of tile 1, reading its weights from slots 0 / 1 with no barrier and no DMA wait in between ->
// one barrier, slots 0 / 1 recycled for steps 3 / 4 -> decoder_tile2.
template <int PREC>
__global__ void __launch_bounds__(NW * 64, 2)
nerf_mlp2_kernel(const int32_t* __restrict__ counters, const float4* __restrict__ tokens, const float* __restrict__ extras,
                 const char* __restrict__ ws, const float* __restrict__ wbias, int64_t capacity, float4* __restrict__ out, int notrans) {
    using CX = Ctx<PREC>;
    static_assert(PREC != 1 && step_pieces<PREC>(0) * 1024 == CX::SLOT, "the transformer reads steps 0 / 1 from ring slots 0 / 1 back to back");
    constexpr int RING = NSLOT * CX::SLOT, TABLES = (N_CHUNKS + 4) * 32 * 4, PARK = 2 * 4 * 1024;
    __shared__ __attribute__((aligned(16))) char lds[RING + TABLES + NW * PARK];
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t n_tiles = (nv + 31) / 32;
    if ((int64_t)blockIdx.x * (2 * NW) >= n_tiles) return;           // whole workgroup beyond the data
    CX cx;
    ring_ctx<PREC>(cx, lds, ws, wbias, false);
    cx.notrans = notrans;
    SHERF_TRACE_REG(cx, 0);
    int64_t tile[2];
    bool live[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        tile[t] = (int64_t)blockIdx.x * (2 * NW) + 2 * cx.wave + t;
        live[t] = tile[t] < n_tiles;
        if (!live[t]) tile[t] = n_tiles - 1;                         // dead tiles still take part in every barrier
    }
    ring_prologue<PREC, 0>(cx);                                      // steps 0, 1 issued, step 0 landed, step 2 issued
    wait_vm((SHERF_MLP_ABLATE & 32) ? 0 : step_pieces<PREC>(2) / NW);   // step 1 (this wave's pieces) landed too
    wg_barrier();
    SHERF_TRACE_REG(cx, 1);
    char* park = lds + RING + TABLES + cx.wave * PARK + cx.lane * 16;
    float xc[2][3], vc[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* ex = extras + tile[t] * 12 * 32 + (cx.lane & 31);
        xc[t][0] = ex[0]; xc[t][1] = ex[32]; xc[t][2] = ex[64]; vc[t][0] = ex[96]; vc[t][1] = ex[128]; vc[t][2] = ex[160];
    }
    {
        const char* ring_lane = cx.lds;
        const float* tables = cx.wbias;
        f32x16 tnext[3];
        load_tokens(tokens, tile[0], cx.lane & 31, cx.h, tnext);
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            f32x16 tok[3] = {tnext[0], tnext[1], tnext[2]};
            if (t == 0) load_tokens(tokens, tile[1], cx.lane & 31, cx.h, tnext);      // tile 1's tokens arrive under tile 0's arithmetic
            // (the weights and tables in LDS do not change between the tiles: launder the offsets, as nerf_tokens_kernel does, or the
            //  compiler hoists their reads out of the loop)
            uint32_t woff = 0, boff = 0;
            asm volatile("" : "+v"(woff), "+v"(boff));
            cx.lds = ring_lane + woff;
            cx.wbias = tables + boff;
            BFrag<PREC> z0b[2], z1b[2];
            transformer_tile<PREC, false, true>(cx, tokens, extras, t ? tile[1] : tile[0], z0b, z1b, tok);
            char* pk = park + t * 4096;
            *reinterpret_cast<u32x4*>(pk) = z0b[0].hi; *reinterpret_cast<u32x4*>(pk + 1024) = z0b[1].hi;
            *reinterpret_cast<u32x4*>(pk + 2048) = z1b[0].hi; *reinterpret_cast<u32x4*>(pk + 3072) = z1b[1].hi;
            SHERF_TRACE_REG(cx, 2 + t);
        }
        cx.lds = ring_lane;
        cx.wbias = tables;
    }
    // every wave is done with the transformer's weights: slots 0 / 1 take steps 3 / 4 (step 2 landed long ago)
    wait_vm(0);
    wg_barrier();
    dma_issue(cx, 3);
    dma_issue(cx, 4);
    SHERF_TRACE_REG(cx, 4);
    decoder_tile2<PREC>(cx, counters, park, xc, vc, tile, live, nv, out);
#if SHERF_MLP_TRACE
    if (g_mlp_trace && g_mlp_trace_every > 0 && blockIdx.x % g_mlp_trace_every == 0) {      // [slot][wave 0..3][192]: stamps; 189 XCC_ID, 190 HW_ID, 191 end
        SHERF_TRACE_REG(cx, 191);
        trace_put(cx, 190, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
        trace_put(cx, 189, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));
        uint32_t* dst = g_mlp_trace + ((size_t)(blockIdx.x / g_mlp_trace_every) * 4 + cx.wave) * 192 + cx.lane;
        dst[0] = cx.tv[0]; dst[64] = cx.tv[1]; dst[128] = cx.tv[2];
    }
#endif
}

// One launch, one tile per wave, the decoder's epilogues inside its MFMA stream (decoder_tile_p; single-product precisions)
#ifndef SHERF_MLP3_LB
#define SHERF_MLP3_LB 2
#endif
template <int PREC, bool PEM = false>
__global__ void __launch_bounds__(NW * 64, SHERF_MLP3_LB)
nerf_mlp3_kernel(const int32_t* __restrict__ counters, const float4* __restrict__ tokens, const float* __restrict__ extras,
                 const char* __restrict__ ws, const float* __restrict__ wbias, int64_t capacity, float4* __restrict__ out, int part, int nparts, int notrans,
                 const u32x4* __restrict__ pefrag) {
    using CX = Ctx<PREC>;
    __shared__ __attribute__((aligned(16))) char lds[NSLOT * CX::SLOT + (N_CHUNKS + 4) * 32 * 4 + (SHERF_MLP_TRACE ? NW * 64 * 16 : 0)];
    const int64_t nv = min((int64_t)counters[0], capacity);
    int64_t t_lo, n_tiles;                                           // this launch's part of the tiles (sherf_nerf_mlp3_part; whole: 0, n)
    sherf_part_range((nv + 31) / 32, part, nparts, t_lo, n_tiles);
    if (t_lo + (int64_t)blockIdx.x * NW >= n_tiles) return;          // whole workgroup beyond the data
    CX cx;
    ring_ctx<PREC>(cx, lds, ws, wbias);
    cx.notrans = notrans;
    int64_t tile = t_lo + (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    const bool live = tile < n_tiles;
    if (!live) tile = n_tiles - 1;                                   // dead tiles still take part in every barrier
    ring_prologue<PREC, 0>(cx);
    BFrag<PREC> z0b[2], z1b[2];
    float xc[3] = {0.f, 0.f, 0.f}, vc[3] = {0.f, 0.f, 0.f};
    if constexpr (!PEM) {
        const float* ex = extras + tile * 12 * 32 + (cx.lane & 31);
        xc[0] = ex[0]; xc[1] = ex[32]; xc[2] = ex[64]; vc[0] = ex[96]; vc[1] = ex[128]; vc[2] = ex[160];
    }
    {
        f32x16 tok[3];
        transformer_tile<PREC, true, false, PEM>(cx, tokens, extras, tile, z0b, z1b, tok, pefrag);
    }
    decoder_tile_p<PREC, PEM>(cx, counters, z0b, z1b, xc, vc, tile, live, nv, out, pefrag + cx.lane);
    SHERF_TRACE_FLUSH(cx);
}

// ---- the two-launch form -------------------------------------------------------------------------------------------------------
// zfrag[tile][q][64 lanes] u32x4: the fused tokens as ready-made B-operand fragments, q = 2 * (0: z_0, 1: z_1) + kb for the single-product
// precisions (4 KiB per tile), q = 4 * (z) + 2 * kb + (0: hi, 1: lo) for prec 1 (8 KiB per tile).
template <int PREC> constexpr int ZFRAGS = PREC == 1 ? 8 : 4;

// Launch 1 of 2: the slot-fusion remainder + the 3-token transformer.  VALU / latency bound; its 24-48 KiB of weights stay resident in
// LDS (no ring, no per-step barrier) and every wave walks its own tiles (wave w of the grid: tiles w, w + W, ...).  (Round 4's first
// form pulled tiles from ONE atomic ticket word: 40 K tickets at the ~88 dequeues / us a single word sustains = 0.45 ms, measured
// 0.54 ms -- profiles/r04_call_a.txt.  The static split leaves at most one tile per wave of imbalance.)
#ifndef SHERF_MLP_TOKENS_WAVES
#define SHERF_MLP_TOKENS_WAVES 4
#endif
template <int PREC>
__global__ void __launch_bounds__(NW * 64, PREC == 1 ? 2 : SHERF_MLP_TOKENS_WAVES)
nerf_tokens_kernel(const int32_t* __restrict__ counters, const float4* __restrict__ tokens, const float* __restrict__ extras,
                   const char* __restrict__ ws, const float* __restrict__ wbias, int64_t capacity, u32x4* __restrict__ zfrag, int notrans) {
    using CX = Ctx<PREC>;
    constexpr int NT = NW * 64;
    constexpr int WBYTES = (step_pieces<PREC>(0) + step_pieces<PREC>(1)) * 1024;
    __shared__ __attribute__((aligned(16))) char lds[WBYTES + (N_CHUNKS + 4) * 32 * 4];
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t n_tiles = (nv + 31) / 32;
    float* lbias = reinterpret_cast<float*>(lds + WBYTES);
    for (int i = threadIdx.x; i < (N_CHUNKS + 4) * 32; i += NT) lbias[i] = wbias[i];
    for (int i = threadIdx.x; i < WBYTES / 16; i += NT) reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(ws)[i];
    __syncthreads();
    CX cx;
    cx.lane = threadIdx.x & 63; cx.h = cx.lane >> 5; cx.wave = threadIdx.x >> 6; cx.ws = ws; cx.lds_addr = 0; cx.notrans = notrans;
    for (int64_t tile = (int64_t)blockIdx.x * NW + cx.wave; tile < n_tiles; tile += (int64_t)gridDim.x * NW) {
        // the weights and tables in LDS do not change between tiles: without the launder the compiler hoists their reads out of the tile
        // loop (hundreds of live registers).  The OFFSETS are laundered, not the pointers: a laundered pointer loses its address space and
        // every weight read becomes a flat load.
        uint32_t woff = cx.lane * 16, boff = 0;
        asm volatile("" : "+v"(woff), "+v"(boff));
        cx.lds = lds + woff;
        cx.wbias = lbias + boff;
        BFrag<PREC> z0b[2], z1b[2];
        {
            f32x16 tok[3];
            transformer_tile<PREC, false>(cx, tokens, extras, tile, z0b, z1b, tok);
        }
        u32x4* zp = zfrag + tile * (ZFRAGS<PREC> * 64) + cx.lane;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (PREC == 1) {
                zp[(2 * kb) * 64] = z0b[kb].hi; zp[(2 * kb + 1) * 64] = z0b[kb].lo;
                zp[(4 + 2 * kb) * 64] = z1b[kb].hi; zp[(4 + 2 * kb + 1) * 64] = z1b[kb].lo;
            } else {
                zp[kb * 64] = z0b[kb].hi; zp[(2 + kb) * 64] = z1b[kb].hi;
            }
        }
    }
}

// Launch 2 of 2: the NeRF decoder, steps 2..42 of the weight stream through the 3-slot ring: every wave of the chip in the MFMA-bound phase.
template <int PREC>
__global__ void __launch_bounds__(NW * 64, PREC == 1 ? 2 : 3)
nerf_decoder_kernel(const int32_t* __restrict__ counters, const u32x4* __restrict__ zfrag, const float* __restrict__ extras,
                    const char* __restrict__ ws, const float* __restrict__ wbias, int64_t capacity, float4* __restrict__ out) {
    using CX = Ctx<PREC>;
    __shared__ __attribute__((aligned(16))) char lds[NSLOT * CX::SLOT + (N_CHUNKS + 4) * 32 * 4 + (SHERF_MLP_TRACE ? NW * 64 * 16 : 0)];
    const int64_t nv = min((int64_t)counters[0], capacity);
    const int64_t n_tiles = (nv + 31) / 32;
    if ((int64_t)blockIdx.x * NW >= n_tiles) return;                 // whole workgroup beyond the data
    CX cx;
    ring_ctx<PREC>(cx, lds, ws, wbias);
    int64_t tile = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    const bool live = tile < n_tiles;
    if (!live) tile = n_tiles - 1;                                   // dead tiles still take part in every barrier
    ring_prologue<PREC, 2>(cx);

    BFrag<PREC> z0b[2], z1b[2];
    const u32x4* zp = zfrag + tile * (ZFRAGS<PREC> * 64) + cx.lane;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if constexpr (PREC == 1) {
            z0b[kb].hi = zp[(2 * kb) * 64]; z0b[kb].lo = zp[(2 * kb + 1) * 64];
            z1b[kb].hi = zp[(4 + 2 * kb) * 64]; z1b[kb].lo = zp[(4 + 2 * kb + 1) * 64];
        } else {
            z0b[kb].hi = zp[kb * 64]; z1b[kb].hi = zp[(2 + kb) * 64];
        }
    }
    float xc[3], vc[3];
    const float* ex = extras + tile * 12 * 32 + (cx.lane & 31);
    xc[0] = ex[0]; xc[1] = ex[32]; xc[2] = ex[64]; vc[0] = ex[96]; vc[1] = ex[128]; vc[2] = ex[160];
    decoder_tile<PREC>(cx, counters, z0b, z1b, xc, vc, tile, live, nv, out);
    SHERF_TRACE_FLUSH(cx);
}

}  // namespace

#if SHERF_MLP_TRACE
extern "C" int sherf_mlp_set_trace(void* buf, int every) {
    SHERF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_trace), &buf, sizeof(buf)));
    SHERF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_trace_every), &every, sizeof(every)));
    return SHERF_OK;
}
#endif

extern "C" int sherf_mlp_stream_layout(int prec, int32_t* n_steps, int32_t* step_pieces_host, int32_t* units, int32_t max_steps) {
    SHERF_CHECK_ARG(prec >= 0 && prec <= 2 && n_steps && step_pieces_host && units && max_steps >= N_STEPS);
    *n_steps = N_STEPS;
    for (int s = 0; s < N_STEPS; ++s) {
        step_pieces_host[s] = prec == 1 ? step_pieces<1>(s) : step_pieces<0>(s);
        for (int u = 0; u < 10; ++u) units[s * 10 + u] = u < step_units(s) ? step_unit(s, u) : -1;
    }
    return SHERF_OK;
}

// ---- the weight stream packed on the device (sherf_amd/mlp_pack.py: stream_index) ----------------------------------------------------
// slot i (2 bytes) of the stream = piece (src[i] & 1: 0 hi, 1 lo) of element flat[src[i] >> 1], zero where src[i] < 0; bias table likewise
// in fp32.  flag |= 1: a packed value is not finite; |= 2: beyond the fp16 range in an fp16 mode (prec 1, 2) -- the caller raises.
namespace {
__global__ void __launch_bounds__(256) mlp_pack_stream_kernel(const float* __restrict__ flat, const int32_t* __restrict__ src, int64_t n16, int prec,
                                                              uint16_t* __restrict__ out, const int32_t* __restrict__ bias_src, int n_bias,
                                                              float* __restrict__ bias_out, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int bad = 0;
    if (i < n16) {
        const int32_t sidx = src[i];
        uint16_t bits = 0;
        if (sidx >= 0) {
            const float v = flat[sidx >> 1];
            if (!(fabsf(v) <= 3.0e38f)) bad |= 1;
            if (prec == 0) {
                const uint32_t b = __float_as_uint(v);
                bits = (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);               // round to nearest even (as torch / numpy)
            } else {
                if (fabsf(v) > 65504.0f) bad |= 2;
                const _Float16 hi = (_Float16)v;
                const _Float16 pc = (sidx & 1) ? (_Float16)(v - (float)hi) : hi;
                bits = __builtin_bit_cast(uint16_t, pc);
            }
        }
        out[i] = bits;
    }
    if (i < n_bias) {
        const int32_t b = bias_src[i];
        const float v = b >= 0 ? flat[b] : 0.f;
        if (!(fabsf(v) <= 3.0e38f)) bad |= 1;
        bias_out[i] = v;
    }
    if (bad) atomicOr(reinterpret_cast<unsigned*>(flag), (unsigned)bad);
}
}  // namespace

extern "C" int sherf_mlp_pack_stream(const float* flat, const int32_t* src, int64_t n_slots, int prec, void* stream_out,
                                     const int32_t* bias_src, int n_bias, float* bias_out, int32_t* flag, sherf_stream_t stream) {
    SHERF_CHECK_ARG(flat && src && stream_out && bias_src && bias_out && flag && n_slots > 0 && n_bias > 0 && n_bias <= n_slots);
    SHERF_CHECK_ARG(prec >= 0 && prec <= 2);
    SHERF_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int32_t), as_stream(stream)));
    hipLaunchKernelGGL(mlp_pack_stream_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, as_stream(stream), flat, src, n_slots, prec,
                       reinterpret_cast<uint16_t*>(stream_out), bias_src, n_bias, bias_out, flag);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_nerf_mlp_part(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                                   const float* wbias, int prec, int64_t capacity, float* out, int part, int nparts, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && tokens && extras && wstream && wbias && out);
    const int notrans = (prec & SHERF_MLP_NO_TRANSFORMER) ? 1 : 0;       // use_trans = False (include/sherf_hip.h)
    prec &= ~SHERF_MLP_NO_TRANSFORMER;
    SHERF_CHECK_ARG(prec >= 0 && prec <= 2 && capacity > 0 && nparts >= 0 && nparts < 256 && (nparts <= 1 || (part >= 0 && part < nparts)));
    const int64_t tiles = nparts > 1 ? ((capacity + 255) / 256 + nparts - 1) / nparts * 8 + 8 : (capacity + 31) / 32;    // most a part can hold
    const dim3 grid((unsigned)((tiles + NW - 1) / NW)), block(NW * 64);
    if (prec == 1)
        hipLaunchKernelGGL((nerf_mlp_kernel<1>), grid, block, 0, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                           reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), part, nparts, notrans);
    else if (prec == 2)
        hipLaunchKernelGGL((nerf_mlp_kernel<2>), grid, block, 0, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                           reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), part, nparts, notrans);
    else
        hipLaunchKernelGGL((nerf_mlp_kernel<0>), grid, block, 0, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                           reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), part, nparts, notrans);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_nerf_mlp(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                              const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream) {
    return sherf_nerf_mlp_part(counters, tokens, extras, wstream, wbias, prec, capacity, out, 0, 1, stream);
}

// Two tiles per wave (nerf_mlp2_kernel): the single-product precisions (prec 0, 2); same inputs, same outputs bit for bit as sherf_nerf_mlp.
extern "C" int sherf_nerf_mlp2(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                               const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && tokens && extras && wstream && wbias && out);
    const int notrans = (prec & SHERF_MLP_NO_TRANSFORMER) ? 1 : 0;
    prec &= ~SHERF_MLP_NO_TRANSFORMER;
    SHERF_CHECK_ARG((prec == 0 || prec == 2) && capacity > 0);
    const int64_t tiles = (capacity + 31) / 32;
    const dim3 grid((unsigned)((tiles + 2 * NW - 1) / (2 * NW))), block(NW * 64);
    if (prec == 2)
        hipLaunchKernelGGL((nerf_mlp2_kernel<2>), grid, block, 0, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                           reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), notrans);
    else
        hipLaunchKernelGGL((nerf_mlp2_kernel<0>), grid, block, 0, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                           reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), notrans);
    SHERF_LAUNCH_CHECK();
}

// One tile per wave with the decoder's epilogues inside its MFMA stream (nerf_mlp3_kernel): the single-product precisions (prec 0, 2); same
// inputs, same outputs bit for bit as sherf_nerf_mlp.
// `wgs_per_cu` (2 or 3; 0 = 3): the launch's residency.  3 = what the kernel's registers and LDS allow (three waves per SIMD); 2 = the launch declares
// 12 KiB of dynamic LDS it never touches, so that only two workgroups fit a CU and a third of every SIMD's registers (and 50 KiB of LDS) stay free for
// ANOTHER kernel's waves -- the gather of the next part of the frame on a second stream (csrc/frame.hip: mlp_parts).  The kernel is power-bound (DESIGN
// section 5.1): its own duration barely depends on the residency.
extern "C" int sherf_nerf_mlp3_part(const int32_t* counters, const float* tokens, const float* extras, const void* wstream, const float* wbias, int prec,
                                    int64_t capacity, float* out, int part, int nparts, int wgs_per_cu, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && tokens && extras && wstream && wbias && out);
    const int notrans = (prec & SHERF_MLP_NO_TRANSFORMER) ? 1 : 0;
    prec &= ~SHERF_MLP_NO_TRANSFORMER;
    SHERF_CHECK_ARG((prec == 0 || prec == 2) && capacity > 0 && nparts >= 0 && nparts < 256 && (nparts <= 1 || (part >= 0 && part < nparts)));
    SHERF_CHECK_ARG(wgs_per_cu >= 0 && wgs_per_cu <= 3);
    const int64_t tiles = nparts > 1 ? ((capacity + 255) / 256 + nparts - 1) / nparts * 8 + 8 : (capacity + 31) / 32;    // most a part can hold
    const dim3 grid((unsigned)((tiles + NW - 1) / NW)), block(NW * 64);
    const unsigned pad = wgs_per_cu == 2 ? 12 * 1024 : wgs_per_cu == 1 ? 44 * 1024 : 0;
    if (wgs_per_cu == 1) {      // static 43.6 KiB + 44 KiB of padding is beyond the 64 KiB a launch may use without saying so
        if (prec == 2) SHERF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_mlp3_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
        else SHERF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_mlp3_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    }
    if (prec == 2)
        hipLaunchKernelGGL((nerf_mlp3_kernel<2>), grid, block, pad, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                           reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), part, nparts, notrans,
                           static_cast<const u32x4*>(nullptr));
    else
        hipLaunchKernelGGL((nerf_mlp3_kernel<0>), grid, block, pad, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                           reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), part, nparts, notrans,
                           static_cast<const u32x4*>(nullptr));
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_nerf_mlp3(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                               const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream) {
    return sherf_nerf_mlp3_part(counters, tokens, extras, wstream, wbias, prec, capacity, out, 0, 1, 0, stream);
}

// sherf_nerf_mlp3 (prec 2 only) with the positional encodings READ as fp16 operand fragments written by sherf_gather_tokens_pe (pefrag:
// [tile][7][64 lanes] x 16 bytes) instead of evaluated in the kernel: same operand bits -> the same outputs bit for bit as sherf_nerf_mlp3 on the
// tokens / extras of the same gather; `extras` is still read by the use_trans = False path only (and may not be NULL).
extern "C" int sherf_nerf_mlp3_pe(const int32_t* counters, const float* tokens, const float* extras, const void* pefrag, const void* wstream,
                                  const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && tokens && extras && pefrag && wstream && wbias && out);
    const int notrans = (prec & SHERF_MLP_NO_TRANSFORMER) ? 1 : 0;
    prec &= ~SHERF_MLP_NO_TRANSFORMER;
    SHERF_CHECK_ARG(prec == 2 && capacity > 0);
    const int64_t tiles = (capacity + 31) / 32;
    const dim3 grid((unsigned)((tiles + NW - 1) / NW)), block(NW * 64);
    hipLaunchKernelGGL((nerf_mlp3_kernel<2, true>), grid, block, 0, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), extras,
                       reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out), 0, 1, notrans,
                       reinterpret_cast<const u32x4*>(pefrag));
    SHERF_LAUNCH_CHECK();
}

// The two-launch form (see nerf_tokens_kernel / nerf_decoder_kernel): same inputs, same outputs bit for bit; zfrag = scratch for the fused
// tokens, (capacity + 31) / 32 tiles x 4 KiB (prec 0, 2) or 8 KiB (prec 1).
extern "C" int sherf_nerf_mlp_split(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                                    const float* wbias, int prec, int64_t capacity, void* zfrag, float* out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && tokens && extras && wstream && wbias && out && zfrag);
    const int notrans = (prec & SHERF_MLP_NO_TRANSFORMER) ? 1 : 0;
    prec &= ~SHERF_MLP_NO_TRANSFORMER;
    SHERF_CHECK_ARG(prec >= 0 && prec <= 2 && capacity > 0);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, v = 0;
        SHERF_HIP_CHECK(hipGetDevice(&dev));
        SHERF_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu = v > 0 ? v : 256;
    }
    const int64_t tiles = (capacity + 31) / 32;
    const int wgs_per_cu = prec == 1 ? 2 : SHERF_MLP_TOKENS_WAVES;           // (a 4-wave workgroup = one wave per SIMD)
    const dim3 tgrid((unsigned)std::min<int64_t>((tiles + NW - 1) / NW, (int64_t)n_cu * wgs_per_cu)), block(NW * 64);
    const dim3 dgrid((unsigned)((tiles + NW - 1) / NW));
#define SHERF_SPLIT(P)                                                                                                                  \
    do {                                                                                                                                 \
        hipLaunchKernelGGL((nerf_tokens_kernel<P>), tgrid, block, 0, as_stream(stream), counters, reinterpret_cast<const float4*>(tokens), \
                           extras, reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<u32x4*>(zfrag), notrans);   \
        hipLaunchKernelGGL((nerf_decoder_kernel<P>), dgrid, block, 0, as_stream(stream), counters, reinterpret_cast<const u32x4*>(zfrag), \
                           extras, reinterpret_cast<const char*>(wstream), wbias, capacity, reinterpret_cast<float4*>(out));             \
    } while (0)
    if (prec == 1) SHERF_SPLIT(1); else if (prec == 2) SHERF_SPLIT(2); else SHERF_SPLIT(0);
#undef SHERF_SPLIT
    SHERF_LAUNCH_CHECK();
}

