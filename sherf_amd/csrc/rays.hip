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

// training/RenderPeople_dataset.py:14-27 (get_rays), 68-101 (get_near_far), 121-134 (casts + packing).
// The reference evaluates get_rays in float64 (K, R, T are float64 numpy arrays), casts ray_o / ray_d to float32 (:123-124),
// and then evaluates the slab test in float64 AGAIN (float32 rays + `bounds + np.array([-0.01, 0.01])` promotes to float64)
// before casting near / far to float32 (:126-127).  The kernel follows exactly that precision ladder, so its fp32 outputs and
// the boolean mask equal numpy's (a float64 last-bit difference survives the float32 rounding with probability ~1e-9).
// Pixel (i=column, j=row, no +0.5): d = ([i,j,1] K^-T - T^T) R - o,  o = -R^T T.  Like the reference (:71, in place on the
// array it returns) a zero direction component becomes 1e-8.
__global__ void __launch_bounds__(256) dataset_rays_kernel(const double* __restrict__ Kinv, const double* __restrict__ Rc,
                                                           const double* __restrict__ Tc, const double* __restrict__ bounds,
                                                           int H, int W, float* __restrict__ ray_o, float* __restrict__ ray_d,
                                                           float* __restrict__ near, float* __restrict__ far,
                                                           uint8_t* __restrict__ mask) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const double i = (double)(p % W), j = (double)(p / W);
    double o64[3], pc[3];
    for (int c = 0; c < 3; ++c) o64[c] = -((Rc[0 * 3 + c] * Tc[0] + Rc[1 * 3 + c] * Tc[1]) + Rc[2 * 3 + c] * Tc[2]);
    for (int c = 0; c < 3; ++c) pc[c] = ((i * Kinv[c * 3 + 0] + j * Kinv[c * 3 + 1]) + Kinv[c * 3 + 2]) - Tc[c];
    double o[3], d[3];
    for (int c = 0; c < 3; ++c) {
        const double pw = (pc[0] * Rc[0 * 3 + c] + pc[1] * Rc[1 * 3 + c]) + pc[2] * Rc[2 * 3 + c];
        float df = (float)(pw - o64[c]);
        const float of = (float)o64[c];
        if (df == 0.0f) df = 1e-8f;
        ray_o[(size_t)p * 3 + c] = of; ray_d[(size_t)p * 3 + c] = df;
        o[c] = (double)of; d[c] = (double)df;
    }
    // slab test against bounds widened by 1 cm; a ray is "at box" when exactly two of its six plane hits lie on the box
    double lo[3], hi[3];
    for (int c = 0; c < 3; ++c) { lo[c] = bounds[c] + -0.01; hi[c] = bounds[3 + c] + 0.01; }
    const double eps = 1e-6;
    int hits = 0;
    double ph[2][3];
    for (int s = 0; s < 2; ++s)                              // reference order of the six planes: min x,y,z then max x,y,z
        for (int c = 0; c < 3; ++c) {
            const double t = ((s ? hi[c] : lo[c]) - o[c]) / d[c];
            const double q[3] = {__dadd_rn(__dmul_rn(t, d[0]), o[0]), __dadd_rn(__dmul_rn(t, d[1]), o[1]), __dadd_rn(__dmul_rn(t, d[2]), o[2])};
            const bool in = q[0] >= lo[0] - eps && q[0] <= hi[0] + eps && q[1] >= lo[1] - eps && q[1] <= hi[1] + eps &&
                            q[2] >= lo[2] - eps && q[2] <= hi[2] + eps;
            if (in) { if (hits < 2) { ph[hits][0] = q[0]; ph[hits][1] = q[1]; ph[hits][2] = q[2]; } ++hits; }
        }
    float nr = 0.f, fr = 1.f;
    if (hits == 2) {
        // np.linalg.norm(axis=1) on float32 ray_d returns float32; on the float64 interval points float64
        const float ndf = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn((float)d[0], (float)d[0]), __fmul_rn((float)d[1], (float)d[1])), __fmul_rn((float)d[2], (float)d[2])));
        const double nd = (double)ndf;
        double a = 0.0, b = 0.0;
        for (int c = 0; c < 3; ++c) { const double u = ph[0][c] - o[c], v = ph[1][c] - o[c]; a = __dadd_rn(a, __dmul_rn(u, u)); b = __dadd_rn(b, __dmul_rn(v, v)); }
        a = sqrt(a) / nd; b = sqrt(b) / nd;
        nr = (float)fmin(a, b); fr = (float)fmax(a, b);
    }
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

extern "C" int sherf_dataset_rays(const double* K_inv, const double* Rc, const double* Tc, const double* bounds, int H, int W,
                                  float* ray_o, float* ray_d, float* near, float* far, uint8_t* mask_at_box,
                                  sherf_stream_t stream) {
    SHERF_CHECK_ARG(K_inv && Rc && Tc && bounds && ray_o && ray_d && near && far && mask_at_box && H > 0 && W > 0);
    hipLaunchKernelGGL(dataset_rays_kernel, dim3(cdiv((int64_t)H * W, 256)), dim3(256), 0, as_stream(stream), K_inv, Rc, Tc,
                       bounds, H, W, ray_o, ray_d, near, far, mask_at_box);
    SHERF_LAUNCH_CHECK();
}
