// upfirdn2d (gfx950): zero-upsample -> pad / crop -> FIR -> downsample as ONE gather per output sample: only the filter taps that
// land on a real input sample are visited (every upy-th row, every upx-th column), so the zero-stuffed image never exists.
// HBM-bound: each thread produces 4 horizontally adjacent outputs (their input footprints overlap and stay in registers / L1), the
// filter lives in LDS, a workgroup covers a 64 x 16 output tile of one (n, c) plane.  Formulas: torch_utils/ops/upfirdn2d.py:139-193
// (`_upfirdn2d_ref`), parameter conventions of upfirdn2d.cpp:22-104.
#include <hip/hip_fp16.h>

#include "ops_common.h"

namespace {

struct UpfirdnParams {
    int N, C, H, W, OH, OW, fh, fw, upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
};

template <class T> __device__ __forceinline__ float ldg(const T* p, int64_t i);
template <> __device__ __forceinline__ float ldg<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldg<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <class T> __device__ __forceinline__ void stg(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void stg<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stg<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

constexpr int kMaxTaps = 32 * 32;
constexpr int kOutPerThread = 4;

template <class T>
__global__ void __launch_bounds__(256) upfirdn2d_kernel(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y, UpfirdnParams p) {
    __shared__ float s_f[kMaxTaps];
    // s_f[ky][kx] = the weight applied to up-sampled position (Y0 + ky, X0 + kx): the flipped filter unless flip_filter (:173-175)
    for (int i = threadIdx.x; i < p.fh * p.fw; i += 256) {
        const int ky = i / p.fw, kx = i % p.fw;
        s_f[i] = p.flip ? f[i] : f[(p.fh - 1 - ky) * p.fw + (p.fw - 1 - kx)];
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;                       // 16 x 16 threads, 4 outputs each along x
    const int ox0 = (blockIdx.x * 16 + tx) * kOutPerThread, oy = blockIdx.y * 16 + ty;
    const int64_t plane = blockIdx.z;                                             // n * C + c
    if (oy >= p.OH || ox0 >= p.OW) return;
    const T* xp = x + plane * p.H * p.W;
    float acc[kOutPerThread] = {0.f, 0.f, 0.f, 0.f};
    const int Y0 = oy * p.downy - p.pady0;                                        // up-sampled row of tap ky = 0
    // first tap row whose up-sampled position is a multiple of upy (a real input row), then every upy-th
    int ky = ((-Y0) % p.upy + p.upy) % p.upy;
    for (; ky < p.fh; ky += p.upy) {
        const int iy = (Y0 + ky) / p.upy;                                         // exact: Y0 + ky is a multiple of upy
        if (Y0 + ky < 0 || iy >= p.H) continue;
#pragma unroll
        for (int q = 0; q < kOutPerThread; ++q) {
            const int ox = ox0 + q;
            if (ox >= p.OW) break;
            const int X0 = ox * p.downx - p.padx0;
            int kx = ((-X0) % p.upx + p.upx) % p.upx;
            for (; kx < p.fw; kx += p.upx) {
                const int ix = (X0 + kx) / p.upx;
                if (X0 + kx < 0 || ix >= p.W) continue;
                acc[q] += s_f[ky * p.fw + kx] * ldg<T>(xp, (int64_t)iy * p.W + ix);
            }
        }
    }
    T* yp = y + (plane * p.OH + oy) * p.OW;
#pragma unroll
    for (int q = 0; q < kOutPerThread; ++q)
        if (ox0 + q < p.OW) stg<T>(yp, ox0 + q, acc[q] * p.gain);
}

}  // namespace

extern "C" int sherf_upfirdn2d(const void* x, const float* f, void* y, int N, int C, int H, int W, int fh, int fw, int upx, int upy,
                               int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip_filter, float gain,
                               int dtype, sherf_stream_t stream) {
    SHERF_CHECK_ARG(x && f && y && N > 0 && C > 0 && H > 0 && W > 0 && fh >= 1 && fw >= 1 && fh * fw <= kMaxTaps);
    SHERF_CHECK_ARG(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1 && (dtype == 0 || dtype == 1));
    const int64_t upW = (int64_t)W * upx + padx0 + padx1, upH = (int64_t)H * upy + pady0 + pady1;
    SHERF_CHECK_ARG(upW >= fw && upH >= fh);                                      // upfirdn2d.py:158-160
    UpfirdnParams p;
    p.N = N; p.C = C; p.H = H; p.W = W; p.fh = fh; p.fw = fw; p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy;
    p.padx0 = padx0; p.pady0 = pady0; p.flip = flip_filter ? 1 : 0; p.gain = gain;
    p.OW = (int)((upW - fw + downx) / downx); p.OH = (int)((upH - fh + downy) / downy);     // upfirdn2d.cpp:60-63
    SHERF_CHECK_ARG(p.OW >= 1 && p.OH >= 1 && (int64_t)N * C <= 65535);
    const dim3 grid((unsigned)((p.OW + 16 * kOutPerThread - 1) / (16 * kOutPerThread)), (unsigned)((p.OH + 15) / 16), (unsigned)(N * C));
    if (dtype == 0) hipLaunchKernelGGL(upfirdn2d_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, f, (float*)y, p);
    else hipLaunchKernelGGL(upfirdn2d_kernel<__half>, grid, dim3(256), 0, as_stream(stream), (const __half*)x, f, (__half*)y, p);
    SHERF_LAUNCH_CHECK();
}
