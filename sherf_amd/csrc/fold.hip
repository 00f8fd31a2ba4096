// Per-frame table folding (gfx950): channel-last re-layout of the tri-planes and of the 2-D feature map with the
// slot projection W (32x32 block of conv1d_reprojection, renderer.py:273,423-424) applied per texel, so the gather
// kernel's taps land directly in token space.  Memory bound: reads NCHW once (coalesced over pixels), writes
// channel-last once.   out[group_base*g + pix*pix_stride + o] = sum_c W[o][c] * in[(g*32 + c)*HW + pix]
#include "common.h"

namespace {

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// HALF: the folded table is written as fp16 (round to nearest even), 8 bytes per channel quad -- the gather then moves half the bytes
// through L2 (it is bound there: ~7.8 KB requested per valid sample in fp32).  Used with the single-product MLP precisions, whose
// first MFMA rounds the tokens to fp16 / bf16 anyway (sherf_amd/renderer.py: table_precision).
template <bool HALF>
__global__ void __launch_bounds__(256) fold32_kernel(const float* __restrict__ in, const float* __restrict__ Wt, float* __restrict__ out,
                                                     int HW, int pix_stride, int64_t group_base) {
    __shared__ __attribute__((aligned(16))) float s_w[32 * 32];          // [c][o]
    for (int i = threadIdx.x; i < 1024; i += 256) s_w[i] = Wt[i];
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (pix >= HW) return;
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = in[((size_t)g * 32 + c) * HW + pix];
    const size_t off = (size_t)g * group_base + (size_t)pix * pix_stride;          // in elements
    float4* o = reinterpret_cast<float4*>(out + off);
    h16x4* oh = reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(out) + off);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float4 w = *reinterpret_cast<const float4*>(s_w + c * 32 + 4 * q);
            a.x += x[c] * w.x; a.y += x[c] * w.y; a.z += x[c] * w.z; a.w += x[c] * w.w;
        }
        if constexpr (HALF) oh[q] = h16x4{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w};
        else o[q] = a;
    }
}

__global__ void __launch_bounds__(256) img4_kernel(const float* __restrict__ img, float4* __restrict__ out, int HW) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < HW) out[p] = make_float4(img[p], img[HW + p], img[2 * (size_t)HW + p], 0.f);
}

}  // namespace

extern "C" int sherf_fold_tables(const float* in, const float* Wt, float* out, int HW, int groups, int pix_stride,
                                 int64_t group_base, int out_half, sherf_stream_t stream) {
    SHERF_CHECK_ARG(in && Wt && out && HW > 0 && groups > 0 && pix_stride >= 32 && pix_stride % 4 == 0);
    if (out_half)
        hipLaunchKernelGGL(fold32_kernel<true>, dim3(cdiv(HW, 256), groups), dim3(256), 0, as_stream(stream), in, Wt, out, HW, pix_stride, group_base);
    else
        hipLaunchKernelGGL(fold32_kernel<false>, dim3(cdiv(HW, 256), groups), dim3(256), 0, as_stream(stream), in, Wt, out, HW, pix_stride, group_base);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_img_to_hwc4(const float* img, float* out, int HW, sherf_stream_t stream) {
    SHERF_CHECK_ARG(img && out && HW > 0);
    hipLaunchKernelGGL(img4_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, as_stream(stream), img, reinterpret_cast<float4*>(out), HW);
    SHERF_LAUNCH_CHECK();
}
