// Ray generation (gfx950): rows a1, a2, a3 of SURVEY.md section 8.
#include "common.h"

namespace {

// training/volumetric_rendering/ray_sampler.py:24-61: uv at pixel centres, x fastest, lifted with fx,fy,cx,cy,skew
__global__ void __launch_bounds__(256) ray_sampler_kernel(const float* __restrict__ c2w, const float* __restrict__ intr, int N,
                                                          int res, float* __restrict__ origins, float* __restrict__ dirs) {
    const int M = res * res;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * M) return;
    const int n = i / M, m = i % M;
    const float* C = c2w + n * 16;
    const float* I = intr + n * 9;
    const float fx = I[0], fy = I[4], cx = I[2], cy = I[5], sk = I[1];
    const float xc = (float)(m % res) * (1.f / res) + (0.5f / res);
    const float yc = (float)(m / res) * (1.f / res) + (0.5f / res);
    const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;
    const float yl = (yc - cy) / fy;
    float w[3];
    for (int r = 0; r < 3; ++r) w[r] = C[r * 4] * xl + C[r * 4 + 1] * yl + C[r * 4 + 2] + C[r * 4 + 3];
    float d[3] = {w[0] - C[3], w[1] - C[7], w[2] - C[11]};
    float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);     // F.normalize eps
    for (int r = 0; r < 3; ++r) { dirs[(size_t)i * 3 + r] = d[r] / nrm; origins[(size_t)i * 3 + r] = C[r * 4 + 3]; }
}

// training/RenderPeople_dataset.py:14-27 (get_rays), 68-101 (get_near_far), 129-134 (packing), evaluated in fp32.
// Pixel (i=column, j=row, no +0.5): d = ([i,j,1] K^-T - T^T) R - o,  o = -R^T T.
__global__ void __launch_bounds__(256) dataset_rays_kernel(const float* __restrict__ Kinv, const float* __restrict__ Rc,
                                                           const float* __restrict__ Tc, const float* __restrict__ bounds,
                                                           int H, int W, float* __restrict__ ray_o, float* __restrict__ ray_d,
                                                           float* __restrict__ near, float* __restrict__ far,
                                                           uint8_t* __restrict__ mask) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const float i = (float)(p % W), j = (float)(p / W);
    float o[3], pc[3], pw[3], d[3];
    for (int c = 0; c < 3; ++c) o[c] = -(Rc[0 * 3 + c] * Tc[0] + Rc[1 * 3 + c] * Tc[1] + Rc[2 * 3 + c] * Tc[2]);
    for (int c = 0; c < 3; ++c) pc[c] = i * Kinv[c * 3 + 0] + j * Kinv[c * 3 + 1] + Kinv[c * 3 + 2] - Tc[c];
    for (int c = 0; c < 3; ++c) pw[c] = pc[0] * Rc[0 * 3 + c] + pc[1] * Rc[1 * 3 + c] + pc[2] * Rc[2 * 3 + c];
    for (int c = 0; c < 3; ++c) { d[c] = pw[c] - o[c]; ray_o[(size_t)p * 3 + c] = o[c]; ray_d[(size_t)p * 3 + c] = d[c]; }
    // slab test against bounds widened by 1 cm; a ray is "at box" when exactly two of its six plane hits lie on the box
    float lo[3], hi[3], dd[3];
    for (int c = 0; c < 3; ++c) { lo[c] = bounds[c] - 0.01f; hi[c] = bounds[3 + c] + 0.01f; dd[c] = d[c] == 0.f ? 1e-8f : d[c]; }
    const float eps = 1e-6f;
    int hits = 0;
    float th[2] = {0.f, 0.f};
    float ph[2][3];
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < 3; ++c) {
            float t = ((s ? hi[c] : lo[c]) - o[c]) / dd[c];
            float q[3] = {t * dd[0] + o[0], t * dd[1] + o[1], t * dd[2] + o[2]};
            bool in = q[0] >= lo[0] - eps && q[0] <= hi[0] + eps && q[1] >= lo[1] - eps && q[1] <= hi[1] + eps &&
                      q[2] >= lo[2] - eps && q[2] <= hi[2] + eps;
            if (in) { if (hits < 2) { th[hits] = t; ph[hits][0] = q[0]; ph[hits][1] = q[1]; ph[hits][2] = q[2]; } ++hits; }
        }
    float nr = 0.f, fr = 1.f;
    if (hits == 2) {
        float nd = sqrtf(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
        float a = sqrtf((ph[0][0] - o[0]) * (ph[0][0] - o[0]) + (ph[0][1] - o[1]) * (ph[0][1] - o[1]) + (ph[0][2] - o[2]) * (ph[0][2] - o[2])) / nd;
        float b = sqrtf((ph[1][0] - o[0]) * (ph[1][0] - o[0]) + (ph[1][1] - o[1]) * (ph[1][1] - o[1]) + (ph[1][2] - o[2]) * (ph[1][2] - o[2])) / nd;
        nr = fminf(a, b); fr = fmaxf(a, b);
    }
    (void)th;
    near[p] = nr; far[p] = fr; mask[p] = hits == 2;
}

}  // namespace

extern "C" int sherf_ray_sampler(const float* cam2world, const float* intrinsics, int N, int res, float* origins,
                                 float* dirs, sherf_stream_t stream) {
    SHERF_CHECK_ARG(cam2world && intrinsics && origins && dirs && N > 0 && res > 0);
    hipLaunchKernelGGL(ray_sampler_kernel, dim3(cdiv((int64_t)N * res * res, 256)), dim3(256), 0, as_stream(stream),
                       cam2world, intrinsics, N, res, origins, dirs);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_dataset_rays(const float* K_inv, const float* Rc, const float* Tc, const float* bounds, int H, int W,
                                  float* ray_o, float* ray_d, float* near, float* far, uint8_t* mask_at_box,
                                  sherf_stream_t stream) {
    SHERF_CHECK_ARG(K_inv && Rc && Tc && bounds && ray_o && ray_d && near && far && mask_at_box && H > 0 && W > 0);
    hipLaunchKernelGGL(dataset_rays_kernel, dim3(cdiv((int64_t)H * W, 256)), dim3(256), 0, as_stream(stream), K_inv, Rc, Tc,
                       bounds, H, W, ray_o, ray_d, near, far, mask_at_box);
    SHERF_LAUNCH_CHECK();
}
