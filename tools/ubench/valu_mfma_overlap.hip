// Microbenchmark (round 4): do VALU instructions of one wave overlap with MFMAs of ANOTHER wave on the same SIMD of gfx950?
// A workgroup = 12 waves = 3 per SIMD (one workgroup per CU, 256 workgroups).  Every wave runs ITER iterations of a body chosen by its role:
//   M: two independent chains of v_mfma_f32_32x32x16_f16 (8 MFMAs per iteration = 256 cycles of matrix pipe)
//   V: NV independent v_fma_f32 (VALU only)
// Cases: all M (3 waves x 256 = pipe-bound), all V, and mixes (1 V + 2 M, 2 V + 1 M), with / without s_setprio on the M waves.
// If the mixes take ~ max(M time, V time) the units overlap across waves; if ~ sum, they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__device__ __forceinline__ void body_v(float (&v)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < NV / 8; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], a, b);
    }
}
__device__ __forceinline__ void body_m(f32x16& c0, f32x16& c1, f16x8 a, f16x8 b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    }
}
// roles: bit w (0..2) of `vmask` set -> the w-th wave of every SIMD is a V wave, else an M wave.  mix: V waves ALSO run the M body every
// `mix`-th iteration (0 = never), M waves also run NVM VALU per iteration (the decoder's own epilogue VALU)
template <int NV, int NVM>
__global__ void __launch_bounds__(768, 3) k(int iters, int vmask, int prio, float* out) {
    const int wave = threadIdx.x >> 6;            // 12 waves; waves are dealt to SIMDs round-robin: wave w -> SIMD w & 3, slot w >> 2
    const int slot = wave >> 2;
    const bool is_v = (vmask >> slot) & 1;
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3f + k;
    f32x16 c0 = {0}, c1 = {0};
    f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * k); }
    if (!is_v && prio) __builtin_amdgcn_s_setprio(2);
    if (is_v) {
        for (int it = 0; it < iters; ++it) body_v<NV>(v, 1.0001f, 1e-6f);
    } else {
        for (int it = 0; it < iters; ++it) { body_m(c0, c1, a, b); if (NVM) body_v<NVM ? NVM : 8>(v, 1.0001f, 1e-6f); }
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += v[k];
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NV, int NVM>
float run(int iters, int vmask, int prio, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NV, NVM>), dim3(256), dim3(768), 0, 0, iters, vmask, prio, d);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NV, NVM>), dim3(256), dim3(768), 0, 0, iters, vmask, prio, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const int iters = 2000;
    // per iteration: M body = 8 MFMAs (256 pipe cycles); V body = NV VALU
    printf("iters %d; M body = 8 MFMA (256 cycles of pipe); times in ms, cycles per iteration per SIMD at 2.4 GHz in brackets\n", iters);
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / iters; };
#define CASE(NV, NVM, vmask, prio, label) { float ms = run<NV, NVM>(iters, vmask, prio, d); printf("%-58s %.3f ms [%.0f]\n", label, ms, cyc(ms)); }
    CASE(64, 0, 0, 0, "MMM  (3 M waves / SIMD)");
    CASE(64, 0, 7, 0, "VVV  NV=64 (3 V waves)");
    CASE(64, 0, 1, 0, "VMM  NV=64");
    CASE(64, 0, 1, 2, "VMM  NV=64, M waves at prio 2");
    CASE(64, 0, 3, 0, "VVM  NV=64");
    CASE(128, 0, 1, 0, "VMM  NV=128");
    CASE(128, 0, 1, 2, "VMM  NV=128, M waves at prio 2");
    CASE(128, 0, 3, 0, "VVM  NV=128");
    CASE(128, 0, 7, 0, "VVV  NV=128");
    CASE(64, 16, 0, 0, "MMM  each M iteration + 16 VALU (the decoder's epilogue share)");
    CASE(64, 32, 0, 0, "MMM  each M iteration + 32 VALU");
    CASE(64, 64, 0, 0, "MMM  each M iteration + 64 VALU");
    CASE(128, 16, 1, 0, "VMM  NV=128, M iterations + 16 VALU");
    return 0;
}
