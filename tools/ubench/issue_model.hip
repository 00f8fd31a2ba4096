// Microbenchmark (round 5): what hides under a v_mfma_f32_32x32x16_f16 on gfx950, and from WHICH wave?
// Round 4's counters say the network kernel runs at the SUM of its MFMA time (32 cycles each) and its VALU issue time (~4.4 cycles each) although
// two or three waves share every SIMD -- i.e. one wave's VALU work does not run under another wave's MFMAs.  Hypothesis: an MFMA that reaches
// the issue stage while the matrix pipe is busy waits THERE and holds the VALU issue port of the SIMD, so the other waves' VALU instructions
// queue behind it; a wave that spaces its own MFMAs (s_nop, or its own fillers) never parks an MFMA at the port.
// Geometry: 512-thread workgroups, one per CU: waves w and w + 4 share a SIMD.  Role A = waves 0-3, role B = waves 4-7.
//   stream M<k>: 8 MFMAs per iteration on two alternating accumulators, each followed by k wait states of s_nop (k = 0: back to back)
//   stream F<k>: the same MFMAs, each followed by k independent v_fma_f32 (in-wave fillers)
//   stream V   : 64 independent v_fma_f32 per iteration (8 chains), no MFMA
//   stream I   : idle (the wave exits)
// Output per case: kernel wall time, and each role's own duration in shader cycles (s_memtime, lane 0 of waves 0 and 4 of block 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF(ACC) "v_mfma_f32_32x32x16_f16 %" #ACC ", %2, %3, %" #ACC "\n\t"
#define VF(R) "v_fma_f32 %" #R ", %" #R ", %12, %13\n\t"

enum { S_IDLE = 0, S_M0, S_M4, S_M6, S_M7, S_F3, S_F5, S_F7, S_V, S_M2 };

template <int S>
__device__ __forceinline__ void body(f32x16& c0, f32x16& c1, const f16x8& a, const f16x8& b, float (&v)[8], float p, float q) {
#define OPS : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(p), "v"(q)
    // (the v[] are inputs only in the M streams; F / V streams declare them read-write below)
    if constexpr (S == S_M0) { asm volatile(MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) OPS); }
    else if constexpr (S == S_M2) { asm volatile(MF(0) "s_nop 1\n\t" MF(1) "s_nop 1\n\t" MF(0) "s_nop 1\n\t" MF(1) "s_nop 1\n\t" MF(0) "s_nop 1\n\t" MF(1) "s_nop 1\n\t" MF(0) "s_nop 1\n\t" MF(1) "s_nop 1\n\t" OPS); }
    else if constexpr (S == S_M4) { asm volatile(MF(0) "s_nop 3\n\t" MF(1) "s_nop 3\n\t" MF(0) "s_nop 3\n\t" MF(1) "s_nop 3\n\t" MF(0) "s_nop 3\n\t" MF(1) "s_nop 3\n\t" MF(0) "s_nop 3\n\t" MF(1) "s_nop 3\n\t" OPS); }
    else if constexpr (S == S_M6) { asm volatile(MF(0) "s_nop 5\n\t" MF(1) "s_nop 5\n\t" MF(0) "s_nop 5\n\t" MF(1) "s_nop 5\n\t" MF(0) "s_nop 5\n\t" MF(1) "s_nop 5\n\t" MF(0) "s_nop 5\n\t" MF(1) "s_nop 5\n\t" OPS); }
    else if constexpr (S == S_M7) { asm volatile(MF(0) "s_nop 6\n\t" MF(1) "s_nop 6\n\t" MF(0) "s_nop 6\n\t" MF(1) "s_nop 6\n\t" MF(0) "s_nop 6\n\t" MF(1) "s_nop 6\n\t" MF(0) "s_nop 6\n\t" MF(1) "s_nop 6\n\t" OPS); }
#undef OPS
#define OPSW : "+v"(c0), "+v"(c1), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a), "v"(b), "v"(p), "v"(q)
#undef MF
#undef VF
#define MF(ACC) "v_mfma_f32_32x32x16_f16 %" #ACC ", %10, %11, %" #ACC "\n\t"
#define VF(R) "v_fma_f32 %" #R ", %" #R ", %12, %13\n\t"
    else if constexpr (S == S_F3) { asm volatile(MF(0) VF(2) VF(3) VF(4) MF(1) VF(5) VF(6) VF(7) MF(0) VF(8) VF(9) VF(2) MF(1) VF(3) VF(4) VF(5) MF(0) VF(6) VF(7) VF(8) MF(1) VF(9) VF(2) VF(3) MF(0) VF(4) VF(5) VF(6) MF(1) VF(7) VF(8) VF(9) OPSW); }
    else if constexpr (S == S_F5) { asm volatile(MF(0) VF(2) VF(3) VF(4) VF(5) VF(6) MF(1) VF(7) VF(8) VF(9) VF(2) VF(3) MF(0) VF(4) VF(5) VF(6) VF(7) VF(8) MF(1) VF(9) VF(2) VF(3) VF(4) VF(5)
                                                 MF(0) VF(6) VF(7) VF(8) VF(9) VF(2) MF(1) VF(3) VF(4) VF(5) VF(6) VF(7) MF(0) VF(8) VF(9) VF(2) VF(3) VF(4) MF(1) VF(5) VF(6) VF(7) VF(8) VF(9) OPSW); }
    else if constexpr (S == S_F7) { asm volatile(MF(0) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) MF(1) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) MF(0) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) MF(1) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5)
                                                 MF(0) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) MF(1) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) MF(0) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) MF(1) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) OPSW); }
    else if constexpr (S == S_V) { asm volatile(VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9)
                                                VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) OPSW); }
#undef OPSW
#undef MF
#undef VF
}

template <int SA, int SB>
__global__ void __launch_bounds__(512, 2) k(int itA, int itB, int prioA, float* out, unsigned long long* cyc) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    f32x16 c0 = {0}, c1 = {0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * ((threadIdx.x + i) % 37)); b[i] = (_Float16)(0.002f * ((threadIdx.x * 7 + i) % 29)); }
    const float p = 1.0001f, q = 1e-6f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (prioA) __builtin_amdgcn_s_setprio(2);
        if constexpr (SA != S_IDLE) for (int it = 0; it < itA; ++it) body<SA>(c0, c1, a, b, v, p, q);
    } else {
        if constexpr (SB != S_IDLE) for (int it = 0; it < itB; ++it) body<SB>(c0, c1, a, b, v, p, q);
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

struct Res { float ms; unsigned long long ca, cb; };
template <int SA, int SB>
Res run(int itA, int itB, int prioA, float* d, unsigned long long* dc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SA, SB>), dim3(256), dim3(512), 0, 0, itA, itB, prioA, d, dc);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<SA, SB>), dim3(256), dim3(512), 0, 0, itA, itB, prioA, d, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
    return {ms / 5, h[0], h[1]};
}

int main() {
    float* d; hipMalloc(&d, 4096);
    unsigned long long* dc; hipMalloc(&dc, 16); hipMemset(dc, 0, 16);
    const int N = 4000;          // iterations: M / F streams 8 MFMAs each (256 pipe cycles), V 64 VALU each
    printf("N = %d iterations; per iteration: M / F = 8 MFMA (256 cycles of pipe) [+ 8k fillers], V = 64 v_fma_f32.  cyc/it = role duration / N (s_memtime ticks)\n", N);
#define CASE(SA, SB, itB, prio, label) { Res r = run<SA, SB>(N, itB, prio, d, dc); printf("%-52s %7.3f ms   A %7.1f cyc/it   B %7.1f cyc/it\n", label, r.ms, (double)r.ca / N, (itB) ? (double)r.cb / (itB) : 0.0); }
    CASE(S_M0, S_IDLE, 0, 0, "M0 alone (back-to-back MFMAs)");
    CASE(S_M2, S_IDLE, 0, 0, "M2 alone (s_nop 1 after each)");
    CASE(S_M4, S_IDLE, 0, 0, "M4 alone (s_nop 3)");
    CASE(S_M6, S_IDLE, 0, 0, "M6 alone (s_nop 5)");
    CASE(S_M7, S_IDLE, 0, 0, "M7 alone (s_nop 6)");
    CASE(S_F3, S_IDLE, 0, 0, "F3 alone (3 fillers per MFMA)");
    CASE(S_F5, S_IDLE, 0, 0, "F5 alone (5 fillers per MFMA)");
    CASE(S_F7, S_IDLE, 0, 0, "F7 alone (7 fillers per MFMA)");
    CASE(S_V, S_IDLE, 0, 0, "V alone (64 VALU / it)");
    CASE(S_M0, S_V, N, 0, "M0 + V");
    CASE(S_M0, S_V, N, 2, "M0 (prio 2) + V");
    CASE(S_M2, S_V, N, 0, "M2 + V");
    CASE(S_M4, S_V, N, 0, "M4 + V");
    CASE(S_M6, S_V, N, 0, "M6 + V");
    CASE(S_M7, S_V, N, 0, "M7 + V");
    CASE(S_M7, S_V, N, 2, "M7 (prio 2) + V");
    CASE(S_M0, S_V, 2 * N, 0, "M0 + V x2 (128 VALU per M iteration)");
    CASE(S_M6, S_V, 2 * N, 0, "M6 + V x2");
    CASE(S_M7, S_V, 2 * N, 0, "M7 + V x2");
    CASE(S_M0, S_M0, N, 0, "M0 + M0");
    CASE(S_M6, S_M6, N, 0, "M6 + M6");
    CASE(S_F5, S_F5, N, 0, "F5 + F5");
    CASE(S_F5, S_V, N, 0, "F5 + V");
    CASE(S_F3, S_V, N, 0, "F3 + V");
    CASE(S_V, S_V, N, 0, "V + V");
    return 0;
}
