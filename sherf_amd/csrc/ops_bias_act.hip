// bias + activation + gain + clamp and its first / second derivative (gfx950).  Element-wise and HBM-bound: one pass, 16-byte
// accesses, grid-stride over float4 / half8 groups so a launch has >= 8 groups in flight per lane's CU slot.  The formulas are the
// reference kernel's (torch_utils/ops/bias_act.cu:27-151), evaluated in float32 for both storage types.
#include <hip/hip_fp16.h>

#include "ops_common.h"

namespace {

struct BiasActParams {
    int64_t n, step_b, size_b;
    float alpha, gain, clamp;
    int grad;
};

template <int A>
__device__ __forceinline__ float bias_act_one(float x, float b, float xref, float yref, float dy, const BiasActParams& p) {
    const int G = p.grad;
    const float gain = p.gain, kExpRange = 80.f, kHalfExpRange = 40.f;
    const float kSeluScale = 1.0507009873554804934193349852946f, kSeluAlpha = 1.6732632423543772848170429916717f;
    const float yy = gain != 0.f ? yref / gain : 0.f;
    if (G == 0) x += b; else xref += b;
    float y = 0.f;
    if (A == 1) y = G == 2 ? 0.f : x;                                          // linear
    if (A == 2) y = G == 0 ? (x > 0.f ? x : 0.f) : G == 1 ? (yy > 0.f ? x : 0.f) : 0.f;
    if (A == 3) y = G == 0 ? (x > 0.f ? x : x * p.alpha) : G == 1 ? (yy > 0.f ? x : x * p.alpha) : 0.f;
    if (A == 4) {
        if (G == 0) { const float c = expf(x), d = 1.f / c; y = x < -kExpRange ? -1.f : x > kExpRange ? 1.f : (c - d) / (c + d); }
        if (G == 1) y = x * (1.f - yy * yy);
        if (G == 2) y = x * (1.f - yy * yy) * (-2.f * yy);
    }
    if (A == 5) {
        if (G == 0) y = x < -kExpRange ? 0.f : 1.f / (expf(-x) + 1.f);
        if (G == 1) y = x * yy * (1.f - yy);
        if (G == 2) y = x * yy * (1.f - yy) * (1.f - 2.f * yy);
    }
    if (A == 6) {
        if (G == 0) y = x >= 0.f ? x : expf(x) - 1.f;
        if (G == 1) y = yy >= 0.f ? x : x * (yy + 1.f);
        if (G == 2) y = yy >= 0.f ? 0.f : x * (yy + 1.f);
    }
    if (A == 7) {
        if (G == 0) y = x >= 0.f ? kSeluScale * x : (kSeluScale * kSeluAlpha) * (expf(x) - 1.f);
        if (G == 1) y = yy >= 0.f ? x * kSeluScale : x * (yy + kSeluScale * kSeluAlpha);
        if (G == 2) y = yy >= 0.f ? 0.f : x * (yy + kSeluScale * kSeluAlpha);
    }
    if (A == 8) {
        if (G == 0) y = x > kExpRange ? x : logf(expf(x) + 1.f);
        if (G == 1) y = x * (1.f - expf(-yy));
        if (G == 2) { const float c = expf(-yy); y = x * c * (1.f - c); }
    }
    if (A == 9) {
        if (G == 0) y = x < -kExpRange ? 0.f : x / (expf(-x) + 1.f);
        else {
            const float c = expf(xref), d = c + 1.f;
            if (G == 1) y = xref > kHalfExpRange ? x : x * c * (xref + d) / (d * d);
            else y = xref > kHalfExpRange ? 0.f : x * c * (xref * (2.f - d) + 2.f * d) / (d * d * d);
            yref = xref < -kExpRange ? 0.f : xref / (expf(-xref) + 1.f) * gain;
        }
    }
    y *= gain * dy;
    if (p.clamp >= 0.f) {
        if (G == 0) y = (y > -p.clamp && y < p.clamp) ? y : (y >= 0.f ? p.clamp : -p.clamp);
        else y = (yref > -p.clamp && yref < p.clamp) ? y : 0.f;
    }
    return y;
}

template <class T> __device__ __forceinline__ float ld(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <class T> __device__ __forceinline__ void st(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

// V elements (16 bytes) per thread and iteration; the tail (n % V) is handled element-wise by the last group.
template <class T, int A, int V>
__global__ void __launch_bounds__(256) bias_act_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ xref,
                                                       const T* __restrict__ yref, const T* __restrict__ dy, T* __restrict__ y,
                                                       BiasActParams p) {
    const int64_t groups = (p.n + V - 1) / V;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        const int64_t i0 = g * V;
        float vx[V], vr[V], vy[V], vd[V];
        const bool full = i0 + V <= p.n;
        if (full) {
            struct alignas(16) Pack { T e[V]; };
            const Pack px = *reinterpret_cast<const Pack*>(x + i0);
#pragma unroll
            for (int k = 0; k < V; ++k) vx[k] = ld<T>(px.e, k);
            if (xref) { const Pack q = *reinterpret_cast<const Pack*>(xref + i0); for (int k = 0; k < V; ++k) vr[k] = ld<T>(q.e, k); }
            if (yref) { const Pack q = *reinterpret_cast<const Pack*>(yref + i0); for (int k = 0; k < V; ++k) vy[k] = ld<T>(q.e, k); }
            if (dy) { const Pack q = *reinterpret_cast<const Pack*>(dy + i0); for (int k = 0; k < V; ++k) vd[k] = ld<T>(q.e, k); }
        }
        struct alignas(16) PackO { T e[V]; } out;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int64_t i = i0 + k;
            if (i >= p.n) break;
            const float xv = full ? vx[k] : ld<T>(x, i);
            const float rv = xref ? (full ? vr[k] : ld<T>(xref, i)) : 0.f;
            const float yv = yref ? (full ? vy[k] : ld<T>(yref, i)) : 0.f;
            const float dv = dy ? (full ? vd[k] : ld<T>(dy, i)) : 1.f;
            const float bv = b ? ld<T>(b, (i / p.step_b) % p.size_b) : 0.f;
            const float r = bias_act_one<A>(xv, bv, rv, yv, dv, p);
            if (full) st<T>(out.e, k, r); else st<T>(y, i, r);
        }
        if (full) *reinterpret_cast<PackO*>(y + i0) = out;
    }
}

template <class T, int V>
int launch(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, const BiasActParams& p, int act,
           sherf_stream_t stream) {
    const int64_t groups = (p.n + V - 1) / V;
    const unsigned grid = (unsigned)((groups + 255) / 256 < 8192 ? (groups + 255) / 256 : 8192);   // >= 32 workgroups per CU when large
#define SHERF_BA(A)                                                                                                          \
    case A:                                                                                                                  \
        hipLaunchKernelGGL((bias_act_kernel<T, A, V>), dim3(grid), dim3(256), 0, as_stream(stream), (const T*)x, (const T*)b,  \
                           (const T*)xref, (const T*)yref, (const T*)dy, (T*)y, p);                                          \
        break;
    switch (act) { SHERF_BA(1) SHERF_BA(2) SHERF_BA(3) SHERF_BA(4) SHERF_BA(5) SHERF_BA(6) SHERF_BA(7) SHERF_BA(8) SHERF_BA(9) }
#undef SHERF_BA
    SHERF_LAUNCH_CHECK();
}

}  // namespace

extern "C" int sherf_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n,
                              int64_t step_b, int64_t size_b, int grad, int act, float alpha, float gain, float clamp, int dtype,
                              sherf_stream_t stream) {
    SHERF_CHECK_ARG(x && y && n >= 0 && grad >= 0 && grad <= 2 && act >= 1 && act <= 9 && (dtype == 0 || dtype == 1));
    SHERF_CHECK_ARG(!b || (step_b >= 1 && size_b >= 1));
    SHERF_CHECK_ARG(grad == 0 || yref || act == 1 || act == 9);          // the derivative formulas read the forward's output
    SHERF_CHECK_ARG(act != 9 || grad == 0 || xref);                      // ... swish its input
    SHERF_CHECK_ARG(((uintptr_t)x | (uintptr_t)y | (uintptr_t)xref | (uintptr_t)yref | (uintptr_t)dy) % 16 == 0);
    if (n == 0) return SHERF_OK;
    BiasActParams p{n, b ? step_b : 1, b ? size_b : 1, alpha, gain, clamp, grad};
    return dtype == 0 ? launch<float, 4>(x, b, xref, yref, dy, y, p, act, stream) : launch<__half, 8>(x, b, xref, yref, dy, y, p, act, stream);
}
