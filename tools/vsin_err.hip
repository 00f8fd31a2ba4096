// Accuracy of gfx950's v_sin_f32 / v_cos_f32 (input in revolutions) against fp64, and of the MLP kernel's phase reduction
// (sincos_exact_phase in sherf_amd/csrc/mlp.hip) on the arguments the positional encodings see:  hipcc --offload-arch=gfx950 tools/vsin_err.hip -o tools/vsin_err
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* a, float* s, float* c, float* s2, float* c2, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    const float kHi = 0.15915494f, kLo = 6.4206383e-09f;
    const float p = x * kHi;
    const float e = __builtin_fmaf(x, kHi, -p);
    const float r = __builtin_amdgcn_fractf(p) + __builtin_fmaf(x, kLo, e);
    s[i] = __builtin_amdgcn_sinf(r); c[i] = __builtin_amdgcn_cosf(r);
    s2[i] = __sinf(x); c2[i] = __cosf(x);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * (i % 6 == 0 ? 1.f : (float)(1 << (i % 6))) * 1.5f;   // |x| <= 1.5 * 2^q
    float *a, *s, *c, *s2, *c2;
    hipMalloc(&a, n * 4); hipMalloc(&s, n * 4); hipMalloc(&c, n * 4); hipMalloc(&s2, n * 4); hipMalloc(&c2, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, s, c, s2, c2, n);
    std::vector<float> hs(n), hc(n), hs2(n), hc2(n);
    hipMemcpy(hs.data(), s, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs2.data(), s2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc2.data(), c2, n * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0, e4 = 0;
    for (int i = 0; i < n; ++i) {
        e1 = fmax(e1, fabs(hs[i] - sin((double)h[i]))); e2 = fmax(e2, fabs(hc[i] - cos((double)h[i])));
        e3 = fmax(e3, fabs(hs2[i] - sin((double)h[i]))); e4 = fmax(e4, fabs(hc2[i] - cos((double)h[i])));
    }
    printf("[vsin] |x| <= 48: exact-phase reduction + v_sin/v_cos: max abs err sin %.3e cos %.3e ;  __sinf/__cosf: sin %.3e cos %.3e\n", e1, e2, e3, e4);
    return 0;
}
