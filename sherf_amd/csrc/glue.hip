// Per-frame glue in front of the path: what TriPlaneGenerator.synthesis does between the
// image encoders and the renderer call (triplane.py:105-137, 174-217) as two launches instead of ~40 small tensor ops --
//   vertex_features : back-face test (compute_normal + projection, renderer.py:50-63, 686-704), pixel-aligned taps of the feature
//                     map and the image (align_corners=True), PE5(rgb)[:32], conv1d_projection 96 -> 32, back-facing rows zeroed;
//   voxelize        : bounds of the canonical vertices +-5 cm, 5 mm voxel coordinates of the canonicalised observation vertices,
//                     sparse-tensor shape (ceil(extent / 0.005) | 31) + 1.
// Inference-time only (no gradient flows through these kernels; training keeps the tensor-op glue so autograd reaches the producers).
#include "common.h"

namespace {

constexpr float kVoxel = 0.005f, kPad = 0.05f;

__device__ __forceinline__ void unit(float& x, float& y, float& z) {          // v / max(|v|, 1e-8)  (compute_normal's clamp_min)
    const float n = fmaxf(sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z))), 1e-8f);
    x = __fdiv_rn(x, n); y = __fdiv_rn(y, n); z = __fdiv_rn(z, n);
}

// zeros-padding bilinear tap of one channel plane, align_corners=True (grid_sample, triplane.py:116-119)
__device__ __forceinline__ float tap(const float* __restrict__ plane, int Hh, int Ww, float gx, float gy) {
    const float ix = (gx + 1.f) * 0.5f * (float)(Ww - 1), iy = (gy + 1.f) * 0.5f * (float)(Hh - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), fx = ix - x0, fy = iy - y0;
    const int xi = (int)x0, yi = (int)y0;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xx = xi + (t & 1), yy = yi + (t >> 1);
        if (xx >= 0 && xx < Ww && yy >= 0 && yy < Hh)
            acc += ((t & 1) ? fx : 1.f - fx) * ((t >> 1) ? fy : 1.f - fy) * plane[(size_t)yy * Ww + xx];
    }
    return acc;
}

// one 64-lane workgroup per vertex: every lane repeats the (cheap) geometry, lane c taps feature channel c, lanes 0..31 finish
__global__ void __launch_bounds__(64) vertex_features_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tri,
                                                             const int32_t* __restrict__ last_face, int V, const float* __restrict__ Rc,
                                                             const float* __restrict__ Tc, const float* __restrict__ K,
                                                             const float* __restrict__ feat, int Hf, int Wf, const float* __restrict__ img,
                                                             int H, int W, const float* __restrict__ Wp, const float* __restrict__ bp,
                                                             float* __restrict__ f3d, uint8_t* __restrict__ front) {
    __shared__ float s_f[96];
    __shared__ float s_rgb[3];
    const int v = blockIdx.x, lane = threadIdx.x;
    // vertex normal: of the faces listing v in column c the one with the highest index contributes (sherf_amd.renderer.compute_normal)
    float nx = 0.f, ny = 0.f, nz = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int f = last_face[c * V + v];
        if (f >= 0) {
            const float* a = verts + 3 * tri[3 * f]; const float* b = verts + 3 * tri[3 * f + 1]; const float* d = verts + 3 * tri[3 * f + 2];
            const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2], e2x = d[0] - a[0], e2y = d[1] - a[1], e2z = d[2] - a[2];
            float cx = __fsub_rn(__fmul_rn(e1y, e2z), __fmul_rn(e1z, e2y)), cy = __fsub_rn(__fmul_rn(e1z, e2x), __fmul_rn(e1x, e2z)),
                  cz = __fsub_rn(__fmul_rn(e1x, e2y), __fmul_rn(e1y, e2x));
            unit(cx, cy, cz);
            nx += cx; ny += cy; nz += cz;
        }
    }
    unit(nx, ny, nz);
    const float px = verts[3 * v], py = verts[3 * v + 1], pz = verts[3 * v + 2];
    const float cx = Rc[0] * px + Rc[1] * py + Rc[2] * pz + Tc[0], cy = Rc[3] * px + Rc[4] * py + Rc[5] * pz + Tc[1],
                cz = Rc[6] * px + Rc[7] * py + Rc[8] * pz + Tc[2];
    const float mx = Rc[0] * nx + Rc[1] * ny + Rc[2] * nz, my = Rc[3] * nx + Rc[4] * ny + Rc[5] * nz, mz = Rc[6] * nx + Rc[7] * ny + Rc[8] * nz;
    const bool is_front = mx * cx + my * cy + mz * cz < 0.f;                               // renderer.py:693-695
    const float hx = K[0] * cx + K[1] * cy + K[2] * cz, hy = K[3] * cx + K[4] * cy + K[5] * cz, hz = K[6] * cx + K[7] * cy + K[8] * cz;
    const float u = hx / (hz + 1e-5f), w = hy / (hz + 1e-5f);
    const float gx = 2.f * u / (float)W - 1.f, gy = 2.f * w / (float)H - 1.f;               // normalised by the IMAGE size (triplane.py:115)
    s_f[lane] = tap(feat + (size_t)lane * Hf * Wf, Hf, Wf, gx, gy);
    if (lane < 3) s_rgb[lane] = tap(img + (size_t)lane * H * W, H, W, gx, gy);
    __syncthreads();
    if (lane < 32) {                 // PE5(rgb) = [rgb, sin(2^q rgb), sin(2^q rgb + pi/2), q = 0..4] (33 values), first 32 kept
        float val;
        if (lane < 3) val = s_rgb[lane];
        else {
            const int t = lane - 3, q = t / 6, r = t % 6;
            val = sinf(fmaf(s_rgb[r % 3], (float)(1 << q), r >= 3 ? 1.57079632679489661923f : 0.f));
        }
        s_f[64 + lane] = val;
    }
    __syncthreads();
    if (lane < 32) {
        float acc = bp[lane];
        const float* wr = Wp + lane * 96;
#pragma unroll 8
        for (int i = 0; i < 96; ++i) acc = fmaf(wr[i], s_f[i], acc);
        f3d[(size_t)v * 32 + lane] = is_front ? acc : 0.f;                                  // triplane.py:126
    }
    if (lane == 0) front[v] = is_front ? 1 : 0;
}

__global__ void __launch_bounds__(1024) voxelize_kernel(const float* __restrict__ t_verts, const float* __restrict__ can, int V,
                                                        float* __restrict__ bounds, int32_t* __restrict__ coord, int32_t* __restrict__ out_sh) {
    __shared__ float s_mn[3][1024 / 64], s_mx[3][1024 / 64];
    __shared__ float s_b[6];
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = threadIdx.x; i < V; i += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float x = t_verts[3 * i + a]; mn[a] = fminf(mn[a], x); mx[a] = fmaxf(mx[a], x); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off)); }
        if ((threadIdx.x & 63) == 0) { s_mn[a][threadIdx.x >> 6] = mn[a]; s_mx[a][threadIdx.x >> 6] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float lo = s_mn[threadIdx.x][0], hi = s_mx[threadIdx.x][0];
        for (int k = 1; k < 1024 / 64; ++k) { lo = fminf(lo, s_mn[threadIdx.x][k]); hi = fmaxf(hi, s_mx[threadIdx.x][k]); }
        lo = __fsub_rn(lo, kPad); hi = __fadd_rn(hi, kPad);                                 // triplane.py:183-185
        s_b[threadIdx.x] = lo; s_b[3 + threadIdx.x] = hi;
        bounds[threadIdx.x] = lo; bounds[3 + threadIdx.x] = hi;
        // out_sh is (z, y, x): (ceil(extent / 0.005) | 31) + 1   (triplane.py:205-207)
        out_sh[2 - threadIdx.x] = ((int)ceilf(__fdiv_rn(__fsub_rn(hi, lo), kVoxel)) | 31) + 1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < V; i += 1024) {
        coord[4 * i] = 0;                                                                    // batch index (per-GPU batch is 1)
#pragma unroll
        for (int a = 0; a < 3; ++a)                                                          // (z, y, x), round half to even like torch.round
            coord[4 * i + 1 + a] = (int)rintf(__fdiv_rn(__fsub_rn(can[3 * i + (2 - a)], s_b[2 - a]), kVoxel));
    }
}

}  // namespace

extern "C" int sherf_vertex_features(const float* verts, const int32_t* tri, const int32_t* last_face, int V, const float* cam_R,
                                     const float* cam_T, const float* cam_K, const float* feat, int Hf, int Wf, const float* img, int H,
                                     int W, const float* Wp, const float* bp, float* f3d, uint8_t* front, sherf_stream_t stream) {
    SHERF_CHECK_ARG(verts && tri && last_face && cam_R && cam_T && cam_K && feat && img && Wp && bp && f3d && front);
    SHERF_CHECK_ARG(V > 0 && Hf > 1 && Wf > 1 && H > 1 && W > 1);
    hipLaunchKernelGGL(vertex_features_kernel, dim3((unsigned)V), dim3(64), 0, as_stream(stream), verts, tri, last_face, V, cam_R, cam_T,
                       cam_K, feat, Hf, Wf, img, H, W, Wp, bp, f3d, front);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_voxelize(const float* t_verts, const float* can, int V, float* bounds, int32_t* coord, int32_t* out_sh,
                              sherf_stream_t stream) {
    SHERF_CHECK_ARG(t_verts && can && bounds && coord && out_sh && V > 0);
    hipLaunchKernelGGL(voxelize_kernel, dim3(1), dim3(1024), 0, as_stream(stream), t_verts, can, V, bounds, coord, out_sh);
    SHERF_LAUNCH_CHECK();
}
