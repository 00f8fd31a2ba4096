// Sparse voxel encoder (gfx950): row a11 of SURVEY.md section 8 -- SparseConvNet (renderer.py:708-871) over the
// <= 6890 voxels that hold an SMPL vertex, without spconv and without ever materialising `.dense()` volumes.
//
// A level is {bitmap (1 bit / voxel), popcount prefix per 32-bit word, keys[row]}: row id == rank of the voxel's
// bit, so rows are sorted by linear voxel index, neighbour lookup is one bit test + popcount, and the level
// order is deterministic (no hash insertion order).  Semantics follow the published spconv-2.x indice-pair
// algorithm as restated in oracle/sherf_oracle.py (rows sharing a voxel sum at that voxel; `mult` carries the
// multiplicity so BatchNorm statistics are taken over the reference's ROW set).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) mark_rows_kernel(const int32_t* __restrict__ coord, int n, int D, int H, int W,
                                                        uint32_t* __restrict__ bitmap) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int z = coord[i * 4 + 1], y = coord[i * 4 + 2], x = coord[i * 4 + 3];
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return;     // outside the T-pose box: dropped
    int key = (z * H + y) * W + x;
    atomicOr(&bitmap[key >> 5], 1u << (key & 31));
}

// SparseConv3d(k=3,s=2,p=1) output sites: o = (q + 1 - k)/2 for every tap k with q+1-k even and o in range
__global__ void __launch_bounds__(256) mark_down_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ n_rows,
                                                        int D, int H, int W, uint32_t* __restrict__ bitmap_out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *n_rows) return;
    const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int key = keys[i];
    int z = key / (H * W), y = (key / W) % H, x = key % W;
    for (int kz = 0; kz < 3; ++kz) {
        int nz = z + 1 - kz;
        if (nz < 0 || (nz & 1) || (nz >> 1) >= Do) continue;
        for (int ky = 0; ky < 3; ++ky) {
            int ny = y + 1 - ky;
            if (ny < 0 || (ny & 1) || (ny >> 1) >= Ho) continue;
            for (int kx = 0; kx < 3; ++kx) {
                int nx = x + 1 - kx;
                if (nx < 0 || (nx & 1) || (nx >> 1) >= Wo) continue;
                int ok = ((nz >> 1) * Ho + (ny >> 1)) * Wo + (nx >> 1);
                atomicOr(&bitmap_out[ok >> 5], 1u << (ok & 31));
            }
        }
    }
}

__global__ void __launch_bounds__(1024) scan_kernel(const uint32_t* __restrict__ bitmap, int n_words,
                                                    int32_t* __restrict__ prefix, int32_t* __restrict__ n_rows) {
    __shared__ int s[1024];
    const int tid = threadIdx.x;
    const int seg = (n_words + 1023) / 1024;
    const int s0 = tid * seg, s1 = min(n_words, s0 + seg);
    int sum = 0;
    for (int i = s0; i < s1; ++i) sum += __popc(bitmap[i]);
    s[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? s[tid - off] : 0;
        __syncthreads();
        s[tid] += v;
        __syncthreads();
    }
    int run = s[tid] - sum;
    for (int i = s0; i < s1; ++i) { prefix[i] = run; run += __popc(bitmap[i]); }
    if (tid == 1023) *n_rows = s[1023];
}

__global__ void __launch_bounds__(256) keys_kernel(const uint32_t* __restrict__ bitmap, const int32_t* __restrict__ prefix,
                                                   int n_words, int32_t* __restrict__ keys) {
    int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    uint32_t bits = bitmap[w];
    int r = prefix[w];
    while (bits) {
        int b = __ffs(bits) - 1;
        keys[r++] = w * 32 + b;
        bits &= bits - 1;
    }
}

__global__ void __launch_bounds__(256) scatter_rows_kernel(const int32_t* __restrict__ coord, const float* __restrict__ feat,
                                                           int n, int C, int D, int H, int W, const uint32_t* __restrict__ bitmap,
                                                           const int32_t* __restrict__ prefix, float* __restrict__ g,
                                                           int32_t* __restrict__ mult) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    int i = idx / C, c = idx % C;
    if (i >= n) return;
    int z = coord[i * 4 + 1], y = coord[i * 4 + 2], x = coord[i * 4 + 3];
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return;
    int key = (z * H + y) * W + x;
    uint32_t word = bitmap[key >> 5], bit = 1u << (key & 31);
    int row = prefix[key >> 5] + __popc(word & (bit - 1u));
    atomicAdd(&g[(size_t)row * C + c], feat[(size_t)i * C + c]);
    if (c == 0) atomicAdd(&mult[row], 1);
}

// out[row][co] = sum_{tap, ci} in[nbr(row, tap)][ci] * wt[tap][ci][co]; blockDim = (TPR, RPB)
constexpr int kMaxCin = 96;
__global__ void sconv_kernel(const int32_t* __restrict__ keys_out, const int32_t* __restrict__ n_rows_out, int Do, int Ho, int Wo,
                             const uint32_t* __restrict__ bitmap_in, const int32_t* __restrict__ prefix_in, int Di, int Hi, int Wi,
                             const float* __restrict__ in, int Cin, const float* __restrict__ wt, int Cout, int down,
                             float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int RPB = blockDim.y, TPR = blockDim.x;
    int* s_nb = reinterpret_cast<int*>(smem);                                   // [RPB][28]
    float* s_in = reinterpret_cast<float*>(smem + RPB * 28 * sizeof(int));      // [RPB][27][Cin]
    const int n_rows = *n_rows_out;
    const int row = blockIdx.x * RPB + threadIdx.y;
    if (blockIdx.x * RPB >= n_rows) return;
    const bool live = row < n_rows;
    const int cx = threadIdx.x;
    if (cx < 27) {
        int nb = -1;
        if (live) {
            int key = keys_out[row];
            int z = key / (Ho * Wo), y = (key / Wo) % Ho, x = key % Wo;
            int kz = cx / 9, ky = (cx / 3) % 3, kx = cx % 3;
            int qz = down ? 2 * z + kz - 1 : z + kz - 1, qy = down ? 2 * y + ky - 1 : y + ky - 1,
                qx = down ? 2 * x + kx - 1 : x + kx - 1;
            if (qz >= 0 && qz < Di && qy >= 0 && qy < Hi && qx >= 0 && qx < Wi) {
                int qk = (qz * Hi + qy) * Wi + qx;
                uint32_t word = bitmap_in[qk >> 5], bit = 1u << (qk & 31);
                if (word & bit) nb = prefix_in[qk >> 5] + __popc(word & (bit - 1u));
            }
        }
        s_nb[threadIdx.y * 28 + cx] = nb;
    }
    __syncthreads();
    for (int tap = 0; tap < 27; ++tap) {
        int nb = s_nb[threadIdx.y * 28 + tap];
        if (nb >= 0)
            for (int ci = cx; ci < Cin; ci += TPR) s_in[(threadIdx.y * 27 + tap) * Cin + ci] = in[(size_t)nb * Cin + ci];
    }
    __syncthreads();
    if (!live || cx >= Cout) return;
    float acc = 0.f;
    for (int tap = 0; tap < 27; ++tap) {
        if (s_nb[threadIdx.y * 28 + tap] < 0) continue;
        const float* si = s_in + (threadIdx.y * 27 + tap) * Cin;
        const float* w = wt + (size_t)tap * Cin * Cout + cx;
        for (int ci = 0; ci < Cin; ++ci) acc += si[ci] * w[(size_t)ci * Cout];
    }
    out[(size_t)row * Cout + cx] = acc;
}

// BatchNorm1d(eps=1e-3) + ReLU over the reference's row set; one block (rows <= ~60k, C <= 96)
__global__ void __launch_bounds__(1024) bn_relu_kernel(float* __restrict__ x, const int32_t* __restrict__ n_rows_p,
                                                       const int32_t* __restrict__ mult, const int32_t* __restrict__ n_total_p,
                                                       int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ stats, int training) {
    __shared__ float s_red[1024];
    __shared__ float s_mean[kMaxCin], s_scale[kMaxCin], s_shift[kMaxCin], s_v0[kMaxCin];
    const int n_rows = *n_rows_p;
    const float n_total = (float)(*n_total_p);
    const int G = 1024 / C;
    const int tid = threadIdx.x;
    const int grp = tid / C, c = tid % C;
    const bool act = grp < G;
    if (training) {
        float sum = 0.f;
        if (act) for (int r = grp; r < n_rows; r += G) sum += x[(size_t)r * C + c];
        s_red[tid] = sum;
        __syncthreads();
        if (tid < C) {
            float t = 0.f;
            for (int g = 0; g < G; ++g) t += s_red[g * C + tid];
            s_mean[tid] = t / n_total;
        }
        __syncthreads();
        float m = act ? s_mean[c] : 0.f, sq = 0.f;
        if (act) for (int r = grp; r < n_rows; r += G) { float d = x[(size_t)r * C + c] - m; sq += d * d; }
        s_red[tid] = sq;
        __syncthreads();
        if (tid < C) {
            float t = 0.f;
            for (int g = 0; g < G; ++g) t += s_red[g * C + tid];
            float mean = s_mean[tid];
            float var = (t + (n_total - (float)n_rows) * mean * mean) / n_total;      // zero rows of the reference
            stats[tid] = mean; stats[C + tid] = var;
        }
        __syncthreads();
    }
    if (tid < C) {
        float mean = stats[tid], var = stats[C + tid];
        float inv = 1.0f / sqrtf(var + 1e-3f);
        s_mean[tid] = mean; s_scale[tid] = inv * gamma[tid]; s_shift[tid] = beta[tid];
        s_v0[tid] = fmaxf((0.f - mean) * inv * gamma[tid] + beta[tid], 0.f);
    }
    __syncthreads();
    if (act)
        for (int r = grp; r < n_rows; r += G) {
            float v = fmaxf((x[(size_t)r * C + c] - s_mean[c]) * s_scale[c] + s_shift[c], 0.f);
            if (mult) v += (float)(mult[r] - 1) * s_v0[c];
            x[(size_t)r * C + c] = v;
        }
}

}  // namespace

extern "C" int sherf_svox_mark_rows(const int32_t* coord, int n, int D, int H, int W, uint32_t* bitmap, sherf_stream_t stream) {
    SHERF_CHECK_ARG(coord && bitmap && n > 0 && D > 0 && H > 0 && W > 0 && (int64_t)D * H * W < 2147483647LL);
    hipLaunchKernelGGL(mark_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), coord, n, D, H, W, bitmap);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_mark_down(const int32_t* keys, const int32_t* n_rows, int D, int H, int W, uint32_t* bitmap_out,
                                    int max_rows, sherf_stream_t stream) {
    SHERF_CHECK_ARG(keys && n_rows && bitmap_out && D > 0 && H > 0 && W > 0 && max_rows > 0);
    hipLaunchKernelGGL(mark_down_kernel, dim3(cdiv(max_rows, 256)), dim3(256), 0, as_stream(stream), keys, n_rows, D, H, W,
                       bitmap_out);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_scan(const uint32_t* bitmap, int n_words, int32_t* prefix, int32_t* n_rows, sherf_stream_t stream) {
    SHERF_CHECK_ARG(bitmap && prefix && n_rows && n_words > 0);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), bitmap, n_words, prefix, n_rows);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_keys(const uint32_t* bitmap, const int32_t* prefix, int n_words, int32_t* keys, sherf_stream_t stream) {
    SHERF_CHECK_ARG(bitmap && prefix && keys && n_words > 0);
    hipLaunchKernelGGL(keys_kernel, dim3(cdiv(n_words, 256)), dim3(256), 0, as_stream(stream), bitmap, prefix, n_words, keys);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_scatter_rows(const int32_t* coord, const float* feat, int n, int C, int D, int H, int W,
                                       const uint32_t* bitmap, const int32_t* prefix, float* g, int32_t* mult,
                                       sherf_stream_t stream) {
    SHERF_CHECK_ARG(coord && feat && bitmap && prefix && g && mult && n > 0 && C > 0);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((int64_t)n * C, 256)), dim3(256), 0, as_stream(stream), coord, feat, n, C,
                       D, H, W, bitmap, prefix, g, mult);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_conv(const int32_t* keys_out, const int32_t* n_rows_out, int Do, int Ho, int Wo,
                               const uint32_t* bitmap_in, const int32_t* prefix_in, int Di, int Hi, int Wi, const float* in,
                               int Cin, const float* wt, int Cout, int down, int max_rows, float* out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(keys_out && n_rows_out && bitmap_in && prefix_in && in && wt && out);
    SHERF_CHECK_ARG(Cin > 0 && Cin <= kMaxCin && Cout > 0 && Cout <= 96 && max_rows > 0);
    const int TPR = ((Cout + 31) / 32) * 32;
    const int RPB = TPR == 32 ? 8 : (TPR == 64 ? 4 : 2);
    const size_t smem = (size_t)RPB * 28 * sizeof(int) + (size_t)RPB * 27 * Cin * sizeof(float);
    hipLaunchKernelGGL(sconv_kernel, dim3(cdiv(max_rows, RPB)), dim3(TPR, RPB), smem, as_stream(stream), keys_out, n_rows_out,
                       Do, Ho, Wo, bitmap_in, prefix_in, Di, Hi, Wi, in, Cin, wt, Cout, down, out);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_bn_relu(float* x, const int32_t* n_rows, const int32_t* mult, const int32_t* n_total_rows, int C,
                                  const float* gamma, const float* beta, float* stats, int training, sherf_stream_t stream) {
    SHERF_CHECK_ARG(x && n_rows && n_total_rows && gamma && beta && stats && C > 0 && C <= kMaxCin);
    hipLaunchKernelGGL(bn_relu_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, n_rows, mult, n_total_rows, C, gamma, beta,
                       stats, training);
    SHERF_LAUNCH_CHECK();
}
