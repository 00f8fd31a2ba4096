// SMPL per-frame tables (gfx950). Rows a7/a8/a9 of SURVEY.md section 8.
//
// The reference evaluates the LBS chain per query point (renderer.py:558-684): blend 24 bone matrices with
// the skinning weights of the nearest vertex, invert the blended 3x3, add/subtract blend-shape offsets,
// blend again.  Everything except the point itself depends only on the nearest vertex id, so the chain is
// an affine map per vertex; we compose it once per frame (6890 rows x 12 floats, L2 resident) in fp64 and
// the per-sample kernels apply one 3x4 affine instead.
#include "common.h"

char g_sherf_err[256] = {0};

int g_sherf_debug = 0;
int g_sherf_cap_trace = 0;
extern "C" int sherf_version(void) { return 100; }
// profiling aid only (ablation switches used by tools/gpu_ablate.sh); 0 in production
extern "C" int sherf_set_debug(int flags) { g_sherf_debug = flags; return SHERF_OK; }
extern "C" const char* sherf_last_error(void) { return g_sherf_err; }

namespace {

// renderer.py:76-94: R = I + sin(a) K + (1-cos a) K K,  a = ||theta + 1e-8||,  K = skew(theta / a)
__device__ void rodrigues(const float* th, float* Rm) {
    float ex = th[0] + 1e-8f, ey = th[1] + 1e-8f, ez = th[2] + 1e-8f;
    float a = sqrtf(ex * ex + ey * ey + ez * ez);
    float rx = th[0] / a, ry = th[1] / a, rz = th[2] / a;
    float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float s = sinf(a), c1 = 1.f - cosf(a);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float kk = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
            Rm[i * 3 + j] = (i == j ? 1.f : 0.f) + s * K[i * 3 + j] + c1 * kk;
        }
}

// one block (64 threads) per parameter set
__global__ void smpl_bones_kernel(const float* __restrict__ poses, const float* __restrict__ shapes,
                                  const float* __restrict__ Jt, const float* __restrict__ Js,
                                  const int32_t* __restrict__ parents, float* __restrict__ A,
                                  float* __restrict__ posefeat) {
    __shared__ float Rm[24][9];
    __shared__ float J[24][3];
    __shared__ float G[24][12];
    const int set = blockIdx.x, t = threadIdx.x;
    const float* th = poses + set * 72;
    const float* be = shapes + set * 10;
    if (t < 24) {
        rodrigues(th + 3 * t, Rm[t]);
        for (int c = 0; c < 3; ++c) {
            float v = Jt[t * 3 + c];
            for (int b = 0; b < 10; ++b) v += Js[(t * 3 + c) * 10 + b] * be[b];
            J[t][c] = v;
        }
    }
    __syncthreads();
    if (t == 0) {
        // renderer.py:96-118: kinematic chain G_i = G_parent [R_i | J_i - J_parent]
        for (int i = 0; i < 24; ++i) {
            float rel[3];
            int p = i == 0 ? -1 : parents[i];
            for (int c = 0; c < 3; ++c) rel[c] = J[i][c] - (i == 0 ? 0.f : J[p][c]);
            if (i == 0) {
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) G[0][r * 4 + c] = Rm[0][r * 3 + c];
                    G[0][r * 4 + 3] = rel[r];
                }
            } else {
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c)
                        G[i][r * 4 + c] = G[p][r * 4 + 0] * Rm[i][0 * 3 + c] + G[p][r * 4 + 1] * Rm[i][1 * 3 + c] +
                                          G[p][r * 4 + 2] * Rm[i][2 * 3 + c];
                    G[i][r * 4 + 3] = G[p][r * 4 + 0] * rel[0] + G[p][r * 4 + 1] * rel[1] + G[p][r * 4 + 2] * rel[2] +
                                      G[p][r * 4 + 3];
                }
            }
        }
    }
    __syncthreads();
    if (t < 24) {
        // renderer.py:121-124: remove the rest pose: t -= G_R * J
        float* out = A + (set * 24 + t) * 12;
        for (int r = 0; r < 3; ++r) {
            float tr = G[t][r * 4 + 3] - (G[t][r * 4 + 0] * J[t][0] + G[t][r * 4 + 1] * J[t][1] + G[t][r * 4 + 2] * J[t][2]);
            out[r * 4 + 0] = G[t][r * 4 + 0]; out[r * 4 + 1] = G[t][r * 4 + 1]; out[r * 4 + 2] = G[t][r * 4 + 2];
            out[r * 4 + 3] = tr;
        }
        if (t >= 1)
            for (int e = 0; e < 9; ++e) posefeat[set * 207 + (t - 1) * 9 + e] = Rm[t][e] - ((e % 4 == 0) ? 1.f : 0.f);
    }
}

constexpr int MAX_SETS = 4;

// one wave per vertex: 3 x 207 dot products per parameter set (posedirs row-triple is 621 contiguous floats)
__global__ void __launch_bounds__(256) smpl_offsets_kernel(const float* __restrict__ posedirs,
                                                           const float* __restrict__ shapedirs,
                                                           const float* __restrict__ posefeat,
                                                           const float* __restrict__ shapes, int n_sets,
                                                           float* __restrict__ PO, float* __restrict__ SO) {
    __shared__ float feat[MAX_SETS][208];
    for (int i = threadIdx.x; i < n_sets * 207; i += blockDim.x) feat[i / 207][i % 207] = posefeat[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = blockIdx.x * 4 + wave;
    if (v >= SHERF_V) return;
    const float* row = posedirs + (size_t)v * 621;
    float acc[MAX_SETS][3];
#pragma unroll
    for (int s = 0; s < MAX_SETS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int k = lane; k < 207; k += 64) {
            float p = row[c * 207 + k];
#pragma unroll
            for (int s = 0; s < MAX_SETS; ++s)
                if (s < n_sets) acc[s][c] += p * feat[s][k];
        }
#pragma unroll
    for (int s = 0; s < MAX_SETS; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x = acc[s][c];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            acc[s][c] = x;
        }
    if (lane == 0)
        for (int s = 0; s < n_sets; ++s)
            for (int c = 0; c < 3; ++c) PO[((size_t)s * SHERF_V + v) * 3 + c] = acc[s][c];
    if (lane < n_sets * 3) {
        int s = lane / 3, c = lane % 3;
        float x = 0.f;
        for (int b = 0; b < 10; ++b) x += shapedirs[((size_t)v * 3 + c) * 10 + b] * shapes[s * 10 + b];
        SO[((size_t)s * SHERF_V + v) * 3 + c] = x;
    }
}

struct Aff { double R[9]; double t[3]; };

__device__ void blend(const float* __restrict__ w, const float* __restrict__ A, double norm, Aff& M) {
    for (int e = 0; e < 9; ++e) M.R[e] = 0.0;
    M.t[0] = M.t[1] = M.t[2] = 0.0;
    for (int b = 0; b < 24; ++b) {
        double wb = (double)w[b] * norm;
        const float* a = A + b * 12;
        for (int r = 0; r < 3; ++r) {
            M.R[r * 3 + 0] += wb * a[r * 4 + 0]; M.R[r * 3 + 1] += wb * a[r * 4 + 1]; M.R[r * 3 + 2] += wb * a[r * 4 + 2];
            M.t[r] += wb * a[r * 4 + 3];
        }
    }
}

__device__ void inv3(const double* m, double* o) {
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__device__ void mat3mul(const double* a, const double* b, double* o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
__device__ void mat3vec(const double* a, const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}

// renderer.py:558-621 collapsed: x_c = Mb_R (inv(M_R)(x - M_t) - PO_tgt - SO_tgt + PO_big) + Mb_t
__global__ void t2c_table_kernel(const float* __restrict__ weights, const float* __restrict__ A_tgt,
                                 const float* __restrict__ A_big, const float* __restrict__ PO_tgt,
                                 const float* __restrict__ SO_tgt, const float* __restrict__ PO_big,
                                 float* __restrict__ T2C) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= SHERF_V) return;
    Aff M, Mb;
    blend(weights + v * 24, A_tgt, 1.0, M);
    blend(weights + v * 24, A_big, 1.0, Mb);
    double Ri[9], P[9], tmp[3], off[3], q[3];
    inv3(M.R, Ri);
    mat3mul(Mb.R, Ri, P);
    mat3vec(Ri, M.t, tmp);
    for (int c = 0; c < 3; ++c)
        off[c] = -tmp[c] - (double)PO_tgt[v * 3 + c] - (double)SO_tgt[v * 3 + c] + (double)PO_big[v * 3 + c];
    mat3vec(Mb.R, off, q);
    float* o = T2C + v * 12;
    for (int e = 0; e < 9; ++e) o[e] = (float)P[e];
    for (int c = 0; c < 3; ++c) o[9 + c] = (float)(q[c] + Mb.t[c]);
}

// renderer.py:623-704 collapsed into h = L x_c + l (homogeneous image coordinates of the observation camera)
__global__ void c2s_table_kernel(const float* __restrict__ weights, const float* __restrict__ A_big,
                                 const float* __restrict__ A_obs, const float* __restrict__ PO_big,
                                 const float* __restrict__ SO_obs, const float* __restrict__ PO_obs,
                                 const float* __restrict__ R_obs, const float* __restrict__ Th_obs,
                                 const float* __restrict__ cam_R, const float* __restrict__ cam_T,
                                 const float* __restrict__ cam_K, float* __restrict__ C2S) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= SHERF_V) return;
    const float* w = weights + v * 24;
    double sw = 0.0;
    for (int b = 0; b < 24; ++b) sw += (double)w[b];   // renderer.py:631-632 re-normalisation
    Aff Mb, Mo;
    blend(w, A_big, 1.0 / sw, Mb);
    blend(w, A_obs, 1.0 / sw, Mo);
    double B[9], E[9], tmp[3], off[3], e[3];
    inv3(Mb.R, B);
    mat3mul(Mo.R, B, E);                        // x_o = E x_c + e
    mat3vec(B, Mb.t, tmp);
    for (int c = 0; c < 3; ++c)
        off[c] = -tmp[c] - (double)PO_big[v * 3 + c] + (double)SO_obs[v * 3 + c] + (double)PO_obs[v * 3 + c];
    mat3vec(Mo.R, off, e);
    for (int c = 0; c < 3; ++c) e[c] += Mo.t[c];
    // x_w = x_o @ inv(Rg) + Th  ==  Gt x_o + Th with Gt = inv(Rg)^T   (renderer.py:681-682)
    double Rg[9], Rgi[9], Gt[9];
    for (int i = 0; i < 9; ++i) Rg[i] = R_obs[i];
    inv3(Rg, Rgi);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Gt[i * 3 + j] = Rgi[j * 3 + i];
    double Rc[9], K[9], RcGt[9], KRcGt[9], L[9], l0[3], l1[3], l[3];
    for (int i = 0; i < 9; ++i) { Rc[i] = cam_R[i]; K[i] = cam_K[i]; }
    mat3mul(Rc, Gt, RcGt);
    mat3mul(K, RcGt, KRcGt);
    mat3mul(KRcGt, E, L);
    mat3vec(Gt, e, l0);
    for (int c = 0; c < 3; ++c) l0[c] += (double)Th_obs[c];
    mat3vec(Rc, l0, l1);
    for (int c = 0; c < 3; ++c) l1[c] += (double)cam_T[c];
    mat3vec(K, l1, l);
    float* o = C2S + v * 12;
    for (int i = 0; i < 9; ++i) o[i] = (float)L[i];
    for (int c = 0; c < 3; ++c) o[9 + c] = (float)l[c];
}

}  // namespace

extern "C" int sherf_smpl_bones(const float* poses, const float* shapes, int n_sets, const float* J_template,
                                const float* J_shapedirs, const int32_t* parents, float* A, float* posefeat,
                                sherf_stream_t stream) {
    SHERF_CHECK_ARG(poses && shapes && J_template && J_shapedirs && parents && A && posefeat);
    SHERF_CHECK_ARG(n_sets >= 1 && n_sets <= MAX_SETS);
    hipLaunchKernelGGL(smpl_bones_kernel, dim3(n_sets), dim3(64), 0, as_stream(stream), poses, shapes, J_template,
                       J_shapedirs, parents, A, posefeat);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_smpl_offsets(const float* posedirs, const float* shapedirs, const float* posefeat,
                                  const float* shapes, int n_sets, float* PO, float* SO, sherf_stream_t stream) {
    SHERF_CHECK_ARG(posedirs && shapedirs && posefeat && shapes && PO && SO);
    SHERF_CHECK_ARG(n_sets >= 1 && n_sets <= MAX_SETS);
    hipLaunchKernelGGL(smpl_offsets_kernel, dim3(cdiv(SHERF_V, 4)), dim3(256), 0, as_stream(stream), posedirs,
                       shapedirs, posefeat, shapes, n_sets, PO, SO);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_smpl_t2c_table(const float* weights, const float* A_tgt, const float* A_big,
                                    const float* PO_tgt, const float* SO_tgt, const float* PO_big, float* T2C,
                                    sherf_stream_t stream) {
    SHERF_CHECK_ARG(weights && A_tgt && A_big && PO_tgt && SO_tgt && PO_big && T2C);
    hipLaunchKernelGGL(t2c_table_kernel, dim3(cdiv(SHERF_V, 128)), dim3(128), 0, as_stream(stream), weights, A_tgt,
                       A_big, PO_tgt, SO_tgt, PO_big, T2C);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_smpl_c2s_table(const float* weights, const float* A_big, const float* A_obs,
                                    const float* PO_big, const float* SO_obs, const float* PO_obs,
                                    const float* R_obs, const float* Th_obs, const float* cam_R, const float* cam_T,
                                    const float* cam_K, float* C2S, sherf_stream_t stream) {
    SHERF_CHECK_ARG(weights && A_big && A_obs && PO_big && SO_obs && PO_obs && R_obs && Th_obs && cam_R && cam_T &&
                    cam_K && C2S);
    hipLaunchKernelGGL(c2s_table_kernel, dim3(cdiv(SHERF_V, 128)), dim3(128), 0, as_stream(stream), weights, A_big,
                       A_obs, PO_big, SO_obs, PO_obs, R_obs, Th_obs, cam_R, cam_T, cam_K, C2S);
    SHERF_LAUNCH_CHECK();
}
