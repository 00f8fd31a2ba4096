// Alpha compositing (gfx950): rows a15 + a16 of SURVEY.md section 8.
//
// MipRayMarcher2.run_forward (ray_marcher.py:25-64, clamp_mode 'relu'):
//   delta_k = (t_{k+1}-t_k)*|d|, last = 1e10*|d|;  alpha = 1-exp(-relu(sigma)*delta);
//   T_k = prod_{i<k} (1-alpha_i+1e-10);  w = alpha*T;  rgb = sum w c;  depth = sum w t / sum w -> nan->inf
//   -> clamp to the GLOBAL [min t, max t];  (+1-acc if white_back);  rgb*2-1.
// Samples the shell mask rejected carry (rgb 0, sigma -80) in the reference (renderer.py:364-368): relu makes
// alpha exactly 0, the transmittance factor is fl(1+1e-10) == 1 and the weight is 0, so skipping them is exact.
// The compact kernel therefore walks only a ray's valid samples (ascending k): HBM-bound, 20 B/ray out.
#include "common.h"

namespace {

__device__ __forceinline__ float depth_at(float near, float range, int k, int S) {
    float step = __fdiv_rn((float)k, (float)(S - 1));
    return __fadd_rn(near, __fmul_rn(step, range));
}

template <int BATCH>
__global__ void __launch_bounds__(256) composite_compact_kernel(const int32_t* __restrict__ counters,
                                                                const int32_t* __restrict__ ray_base,
                                                                const int32_t* __restrict__ ray_cnt,
                                                                const int32_t* __restrict__ cs_idx,
                                                                const float4* __restrict__ sample_out,
                                                                const float* __restrict__ ray_d, const float* __restrict__ near,
                                                                const float* __restrict__ far, int R, int S, int white_back,
                                                                int64_t tok_cap, int32_t* __restrict__ flags,
                                                                float* __restrict__ rgb, float* __restrict__ depth,
                                                                float* __restrict__ acc) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    if ((int64_t)ray_base[r] + ray_cnt[r] > tok_cap) {
        // the frame holds more valid samples than the token-side buffers (sherf_frame.tok_capacity): this ray's samples were not
        // evaluated.  Loud, not wrong: NaN pixels + counters[3] bit 1; the caller re-sizes from counters[0] and renders again.
        const float qnan = __int_as_float(0x7fc00000);
        rgb[r * 3] = rgb[r * 3 + 1] = rgb[r * 3 + 2] = qnan; depth[r] = qnan; acc[r] = qnan;
        if (flags) atomicOr(flags, 2);
        return;
    }
    const float dmin = ord2f(counters[1]), dmax = ord2f(counters[2]);
    const float d0 = ray_d[r * 3], d1 = ray_d[r * 3 + 1], d2 = ray_d[r * 3 + 2];
    const float dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float nr = near[r], range = __fsub_rn(far[r], nr);
    const int base = ray_base[r], cnt = ray_cnt[r];
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, wsum = 0.f, dsum = 0.f;
    // A ray's samples in batches of BATCH = 4 (1: round 5's loop, SHERF_EXPERIMENT bit 20): the eight loads of a batch are requested together, then its samples enter the running products in order (the same arithmetic in the
    // same order).  One sample per trip made a wave wait for memory once per sample of its longest ray -- up to 64 dependent round trips at the tail of every frame.
    for (int i0 = 0; i0 < cnt; i0 += BATCH) {
        int kk[BATCH];
        float4 oo[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int i = min(i0 + j, cnt - 1);
            kk[j] = cs_idx[base + i] - r * S;
            oo[j] = sample_out[base + i];
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (i0 + j >= cnt) break;
            const int k = kk[j];
            const float4 o = oo[j];
            const float t = depth_at(nr, range, k, S);
            const float delta = (k == S - 1 ? 1e10f : depth_at(nr, range, k + 1, S) - t) * dn;
            const float alpha = 1.f - expf(-(fmaxf(o.w, 0.f) * delta));
            const float w = alpha * T;
            T = T * (1.f - alpha + 1e-10f);
            cr += w * o.x; cg += w * o.y; cb += w * o.z; wsum += w; dsum += w * t;
        }
    }
    float dep = dsum / wsum;                                   // 0/0 -> NaN for empty rays
    if (dep != dep) dep = __int_as_float(0x7f800000);          // nan_to_num(.., inf)
    dep = fminf(fmaxf(dep, dmin), dmax);
    if (white_back) { cr += 1.f - wsum; cg += 1.f - wsum; cb += 1.f - wsum; }
    rgb[r * 3] = cr * 2.f - 1.f; rgb[r * 3 + 1] = cg * 2.f - 1.f; rgb[r * 3 + 2] = cb * 2.f - 1.f;
    depth[r] = dep;
    acc[r] = wsum;
}

__global__ void __launch_bounds__(256) composite_dense_kernel(const float* __restrict__ colors, const float* __restrict__ sigma,
                                                              const float* __restrict__ depths, const float* __restrict__ rays_d,
                                                              int R, int S, int white_back, const float* __restrict__ dminmax,
                                                              float* __restrict__ rgb, float* __restrict__ depth,
                                                              float* __restrict__ weights) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float dmin = dminmax[0], dmax = dminmax[1];
    const float d0 = rays_d[r * 3], d1 = rays_d[r * 3 + 1], d2 = rays_d[r * 3 + 2];
    const float dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float* c = colors + (size_t)r * S * 3;
    const float* sg = sigma + (size_t)r * S;
    const float* tt = depths + (size_t)r * S;
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, wsum = 0.f, dsum = 0.f;
    for (int k = 0; k < S; ++k) {
        const float t = tt[k];
        const float delta = (k == S - 1 ? 1e10f : tt[k + 1] - t) * dn;
        const float alpha = 1.f - expf(-(fmaxf(sg[k], 0.f) * delta));
        const float w = alpha * T;
        T = T * (1.f - alpha + 1e-10f);
        weights[(size_t)r * S + k] = w;
        cr += w * c[k * 3]; cg += w * c[k * 3 + 1]; cb += w * c[k * 3 + 2]; wsum += w; dsum += w * t;
    }
    float dep = dsum / wsum;
    if (dep != dep) dep = __int_as_float(0x7f800000);
    dep = fminf(fmaxf(dep, dmin), dmax);
    if (white_back) { cr += 1.f - wsum; cg += 1.f - wsum; cb += 1.f - wsum; }
    rgb[r * 3] = cr * 2.f - 1.f; rgb[r * 3 + 1] = cg * 2.f - 1.f; rgb[r * 3 + 2] = cb * 2.f - 1.f;
    depth[r] = dep;
}

// Backward of composite_compact_kernel w.r.t. the per-sample (rgb, sigma), for a loss that reads rgb_final and acc
// (the reference's losses never read the depth map: loss.py:103-176).  With g_c = 2 dL/drgb_final and
//   g_w[k] = g_c . c_k + dL/dacc - white_back * sum(g_c):
//   dL/dc_k     = g_c w_k
//   dL/dalpha_k = g_w[k] T_k - (sum_{m>k} g_w[m] w_m) / (1 - alpha_k + 1e-10)        (T_m carries the factor of sample k)
//   dL/dsigma_k = dL/dalpha_k * delta_k * exp(-sigma_k delta_k) * [sigma_k > 0]
// Two forward sweeps over the ray's compact samples: the first takes the total of g_w w, the second rebuilds T, alpha, w
// and turns the running prefix into the suffix sum.  Same thread-per-ray shape and traffic class as the forward.
__global__ void __launch_bounds__(256) composite_compact_bwd_kernel(const int32_t* __restrict__ ray_base, const int32_t* __restrict__ ray_cnt,
                                                                    const int32_t* __restrict__ cs_idx, const float4* __restrict__ sample_out,
                                                                    const float* __restrict__ ray_d, const float* __restrict__ near,
                                                                    const float* __restrict__ far, int R, int S, int white_back,
                                                                    const float* __restrict__ d_rgb, const float* __restrict__ d_acc,
                                                                    float4* __restrict__ d_sample_out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float d0 = ray_d[r * 3], d1 = ray_d[r * 3 + 1], d2 = ray_d[r * 3 + 2];
    const float dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float nr = near[r], range = __fsub_rn(far[r], nr);
    const int base = ray_base[r], cnt = ray_cnt[r];
    const float gx = 2.f * d_rgb[r * 3], gy = 2.f * d_rgb[r * 3 + 1], gz = 2.f * d_rgb[r * 3 + 2];
    const float gconst = d_acc[r] - (white_back ? gx + gy + gz : 0.f);
    float T = 1.f, total = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const int k = cs_idx[base + i] - r * S;
        const float4 o = sample_out[base + i];
        const float t = depth_at(nr, range, k, S);
        const float delta = (k == S - 1 ? 1e10f : depth_at(nr, range, k + 1, S) - t) * dn;
        const float alpha = 1.f - expf(-(fmaxf(o.w, 0.f) * delta));
        total += (gx * o.x + gy * o.y + gz * o.z + gconst) * (alpha * T);
        T = T * (1.f - alpha + 1e-10f);
    }
    T = 1.f;
    float prefix = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const int k = cs_idx[base + i] - r * S;
        const float4 o = sample_out[base + i];
        const float t = depth_at(nr, range, k, S);
        const float delta = (k == S - 1 ? 1e10f : depth_at(nr, range, k + 1, S) - t) * dn;
        const float e = expf(-(fmaxf(o.w, 0.f) * delta));
        const float alpha = 1.f - e, w = alpha * T;
        const float gw = gx * o.x + gy * o.y + gz * o.z + gconst;
        prefix += gw * w;
        const float dalpha = gw * T - (total - prefix) / (1.f - alpha + 1e-10f);
        const float dsigma = o.w > 0.f ? dalpha * delta * e : 0.f;
        d_sample_out[base + i] = make_float4(gx * w, gy * w, gz * w, dsigma);
        T = T * (1.f - alpha + 1e-10f);
    }
}

}  // namespace

extern "C" int sherf_composite_compact_cap(int32_t* counters, const int32_t* ray_base, const int32_t* ray_cnt,
                                           const int32_t* cs_idx, const float* sample_out, const float* ray_d,
                                           const float* near, const float* far, int R, int S, int white_back, int64_t tok_cap,
                                           float* rgb, float* depth, float* acc, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && ray_base && ray_cnt && cs_idx && sample_out && ray_d && near && far && rgb && depth && acc);
    SHERF_CHECK_ARG(R > 0 && S >= 2 && tok_cap > 0);
    if (sherf_experiment() & (1 << 20))
        hipLaunchKernelGGL(composite_compact_kernel<1>, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), counters, ray_base,
                           ray_cnt, cs_idx, reinterpret_cast<const float4*>(sample_out), ray_d, near, far, R, S, white_back, tok_cap,
                           counters + 3, rgb, depth, acc);
    else
    hipLaunchKernelGGL(composite_compact_kernel<4>, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), counters, ray_base,
                       ray_cnt, cs_idx, reinterpret_cast<const float4*>(sample_out), ray_d, near, far, R, S, white_back, tok_cap,
                       counters + 3, rgb, depth, acc);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_composite_compact(const int32_t* counters, const int32_t* ray_base, const int32_t* ray_cnt,
                                       const int32_t* cs_idx, const float* sample_out, const float* ray_d,
                                       const float* near, const float* far, int R, int S, int white_back, float* rgb,
                                       float* depth, float* acc, sherf_stream_t stream) {
    SHERF_CHECK_ARG(counters && ray_base && ray_cnt && cs_idx && sample_out && ray_d && near && far && rgb && depth && acc);
    SHERF_CHECK_ARG(R > 0 && S >= 2);
    hipLaunchKernelGGL(composite_compact_kernel<4>, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), counters, ray_base,
                       ray_cnt, cs_idx, reinterpret_cast<const float4*>(sample_out), ray_d, near, far, R, S, white_back,
                       (int64_t)R * S, static_cast<int32_t*>(nullptr), rgb, depth, acc);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_composite_dense(const float* colors, const float* sigma, const float* depths, const float* rays_d,
                                     int R, int S, int white_back, const float* dminmax, float* rgb, float* depth,
                                     float* weights, sherf_stream_t stream) {
    SHERF_CHECK_ARG(colors && sigma && depths && rays_d && dminmax && rgb && depth && weights);
    SHERF_CHECK_ARG(R > 0 && S >= 1);
    hipLaunchKernelGGL(composite_dense_kernel, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), colors, sigma, depths,
                       rays_d, R, S, white_back, dminmax, rgb, depth, weights);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_composite_compact_bwd(const int32_t* ray_base, const int32_t* ray_cnt, const int32_t* cs_idx,
                                           const float* sample_out, const float* ray_d, const float* near, const float* far,
                                           int R, int S, int white_back, const float* d_rgb, const float* d_acc,
                                           float* d_sample_out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(ray_base && ray_cnt && cs_idx && sample_out && ray_d && near && far && d_rgb && d_acc && d_sample_out);
    SHERF_CHECK_ARG(R > 0 && S >= 2);
    hipLaunchKernelGGL(composite_compact_bwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), ray_base, ray_cnt, cs_idx,
                       reinterpret_cast<const float4*>(sample_out), ray_d, near, far, R, S, white_back, d_rgb, d_acc,
                       reinterpret_cast<float4*>(d_sample_out));
    SHERF_LAUNCH_CHECK();
}
