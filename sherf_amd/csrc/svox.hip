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

// exclusive popcount scan, 3 launches: per-1024-word chunk scan, scan of chunk sums, add chunk offsets
__global__ void __launch_bounds__(256) scan_local_kernel(const uint32_t* __restrict__ bitmap, int n_words,
                                                        int32_t* __restrict__ prefix, int32_t* __restrict__ chunk_sum) {
    __shared__ int s[256];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * 1024 + tid * 4;
    int c[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[i] = (w0 + i < n_words) ? __popc(bitmap[w0 + i]) : 0; sum += c[i]; }
    s[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int v = tid >= off ? s[tid - off] : 0;
        __syncthreads();
        s[tid] += v;
        __syncthreads();
    }
    int run = s[tid] - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (w0 + i < n_words) prefix[w0 + i] = run; run += c[i]; }
    if (tid == 255) chunk_sum[blockIdx.x] = s[255];
}

__global__ void __launch_bounds__(1024) scan_chunks_kernel(int32_t* __restrict__ chunk_sum, int n_chunks, int32_t* __restrict__ n_rows) {
    __shared__ int s[1024];
    int carry = 0;
    for (int b0 = 0; b0 < n_chunks; b0 += 1024) {
        int i = b0 + threadIdx.x;
        int v = i < n_chunks ? chunk_sum[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int a = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_chunks) chunk_sum[i] = carry + s[threadIdx.x] - v;
        int tot = s[1023];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) *n_rows = carry;
}

__global__ void __launch_bounds__(256) scan_add_kernel(const uint32_t* __restrict__ bitmap, int32_t* __restrict__ prefix, int n_words,
                                                       const int32_t* __restrict__ chunk_off, uint2* __restrict__ wp) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w < n_words) {
        const int pf = prefix[w] + chunk_off[w >> 10];
        prefix[w] = pf;
        wp[w] = make_uint2(bitmap[w], (uint32_t)pf);          // one 8-byte record per word: a lookup is a single load
    }
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

// Rows that share a voxel SUM there (spconv indice pairs).  Floating-point atomics would make the sum depend on the
// arrival order when three or more rows collide, so the sum is taken in 2^-30 fixed point (integer adds commute:
// bitwise reproducible), then converted once.  |feature| must stay below 2^33.
constexpr double kFix = 1073741824.0;   // 2^30
__global__ void __launch_bounds__(256) scatter_rows_kernel(const int32_t* __restrict__ coord, const float* __restrict__ feat,
                                                           int n, int C, int D, int H, int W, const uint32_t* __restrict__ bitmap,
                                                           const int32_t* __restrict__ prefix, unsigned long long* __restrict__ acc,
                                                           int32_t* __restrict__ mult) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    int i = idx / C, c = idx % C;
    if (i >= n) return;
    int z = coord[i * 4 + 1], y = coord[i * 4 + 2], x = coord[i * 4 + 3];
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return;
    int key = (z * H + y) * W + x;
    uint32_t word = bitmap[key >> 5], bit = 1u << (key & 31);
    int row = prefix[key >> 5] + __popc(word & (bit - 1u));
    long long q = __double2ll_rn((double)feat[(size_t)i * C + c] * kFix);
    atomicAdd(&acc[(size_t)row * C + c], (unsigned long long)q);
    if (c == 0) atomicAdd(&mult[row], 1);
}

__global__ void __launch_bounds__(256) fix_to_float_kernel(const long long* __restrict__ acc, const int32_t* __restrict__ n_rows, int C,
                                                           float* __restrict__ g) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < *n_rows * C) g[idx] = (float)((double)acc[idx] * (1.0 / kFix));
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


// ---------------------------------------------------------------------------------------------
// v2: tiled sparse convolution.  A block owns TM output rows; per kernel tap it stages W_tap [Cin][COUT] and the
// gathered (BatchNorm+ReLU applied on the fly) input rows in LDS and does a 4x4 register-tiled fp32 product.
// mode 0 submanifold, 1 stride-2 (k3 p1), 2 pointwise (1 tap, row -> same row; used to fold the 1x1 projections).
// BatchNorm statistics of the OUTPUT are produced as per-block fp64 partial sums (deterministic two-stage reduce).
// ---------------------------------------------------------------------------------------------
template <int COUT, int NTHR>
__global__ void __launch_bounds__(NTHR) sconv2_kernel(const int32_t* __restrict__ keys_out, const int32_t* __restrict__ n_rows_out,
                                                     int Do, int Ho, int Wo, const uint2* __restrict__ wp_in, int Di, int Hi, int Wi,
                                                     const float* __restrict__ in_raw, int Cin, const float* __restrict__ in_bn,
                                                     const int32_t* __restrict__ in_mult, const float* __restrict__ wt, int mode,
                                                     float* __restrict__ out_raw, double* __restrict__ partials) {
    constexpr int CQ = COUT / 4, RQ = NTHR / CQ, TM = RQ * 4, TMP = TM + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_w = reinterpret_cast<float*>(smem);                          // [Cin][COUT]
    float* s_in = s_w + Cin * COUT;                                       // [Cin][TMP]
    int* s_nb = reinterpret_cast<int*>(s_in + Cin * TMP);                 // [ntaps][TM]
    float* s_red = s_in;                                                  // [2][RQ][COUT], reused after the tap loop
    const int n_rows = *n_rows_out;
    const int row0 = blockIdx.x * TM;
    if (row0 >= n_rows) return;
    const int tid = threadIdx.x;
    const int ntaps = mode == 2 ? 1 : 27;
    for (int i = tid; i < ntaps * TM; i += NTHR) {
        const int tap = i / TM, r = i % TM, row = row0 + r;
        int nb = -1;
        if (row < n_rows) {
            if (mode == 2) nb = row;
            else {
                const int key = keys_out[row];
                const int z = key / (Ho * Wo), y = (key / Wo) % Ho, x = key % Wo;
                const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
                const int qz = mode ? 2 * z + kz - 1 : z + kz - 1, qy = mode ? 2 * y + ky - 1 : y + ky - 1,
                          qx = mode ? 2 * x + kx - 1 : x + kx - 1;
                if (qz >= 0 && qz < Di && qy >= 0 && qy < Hi && qx >= 0 && qx < Wi) {
                    const int qk = (qz * Hi + qy) * Wi + qx;
                    const uint2 rec = wp_in[qk >> 5];
                    const uint32_t bit = 1u << (qk & 31);
                    if (rec.x & bit) nb = (int)rec.y + __popc(rec.x & (bit - 1u));
                }
            }
        }
        s_nb[tap * TM + r] = nb;
    }
    __syncthreads();
    const int cq = tid % CQ, rq = tid / CQ;
    const bool worker = rq < RQ;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int tap = 0; tap < ntaps; ++tap) {
        bool any = false;
        for (int r = tid; r < TM; r += NTHR) any |= s_nb[tap * TM + r] >= 0;
        if (!__syncthreads_or(any)) continue;                             // no row of this tile has that neighbour
        const float4* wsrc = reinterpret_cast<const float4*>(wt + (size_t)tap * Cin * COUT);
        for (int i = tid; i < Cin * COUT / 4; i += NTHR) reinterpret_cast<float4*>(s_w)[i] = wsrc[i];
        for (int i = tid; i < TM * Cin; i += NTHR) {
            const int r = i / Cin, ci = i % Cin;
            const int nb = s_nb[tap * TM + r];
            float v = 0.f;
            if (nb >= 0) {
                v = in_raw[(size_t)nb * Cin + ci];
                if (in_bn) {
                    v = fmaxf(v * in_bn[ci] + in_bn[Cin + ci], 0.f);
                    if (in_mult) v += (float)(in_mult[nb] - 1) * in_bn[2 * Cin + ci];
                }
            }
            s_in[ci * TMP + r] = v;
        }
        __syncthreads();
        if (worker) {
            for (int ci = 0; ci < Cin; ++ci) {
                const float4 a = *reinterpret_cast<const float4*>(s_in + ci * TMP + 4 * rq);
                const float4 w = *reinterpret_cast<const float4*>(s_w + ci * COUT + 4 * cq);
                acc[0][0] += a.x * w.x; acc[0][1] += a.x * w.y; acc[0][2] += a.x * w.z; acc[0][3] += a.x * w.w;
                acc[1][0] += a.y * w.x; acc[1][1] += a.y * w.y; acc[1][2] += a.y * w.z; acc[1][3] += a.y * w.w;
                acc[2][0] += a.z * w.x; acc[2][1] += a.z * w.y; acc[2][2] += a.z * w.z; acc[2][3] += a.z * w.w;
                acc[3][0] += a.w * w.x; acc[3][1] += a.w * w.y; acc[3][2] += a.w * w.z; acc[3][3] += a.w * w.w;
            }
        }
        __syncthreads();
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (worker) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * rq + i;
            if (row < n_rows) {
                *reinterpret_cast<float4*>(out_raw + (size_t)row * COUT + 4 * cq) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1[j] += acc[i][j]; s2[j] += acc[i][j] * acc[i][j]; }
            }
        }
    }
    if (partials) {
        if (worker)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s_red[rq * COUT + 4 * cq + j] = s1[j]; s_red[(RQ + rq) * COUT + 4 * cq + j] = s2[j]; }
        __syncthreads();
        for (int u = tid; u < 2 * COUT; u += NTHR) {
            const int which = u / COUT, co = u % COUT;
            double t = 0.0;
            for (int q = 0; q < RQ; ++q) t += (double)s_red[(which * RQ + q) * COUT + co];
            partials[((size_t)blockIdx.x * 2 + which) * COUT + co] = t;
        }
    }
}

// mean/var over the reference's ROW set from the per-block partials -> bnparam[3][C] = (scale, shift, relu(shift)), stats[2][C]
__global__ void __launch_bounds__(1024) bn_finalize_kernel(const double* __restrict__ partials, const int32_t* __restrict__ n_rows_p,
                                                           const int32_t* __restrict__ n_total_p, int C, int rows_per_block,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ stats, int training, float* __restrict__ bnparam) {
    __shared__ double s[1024];
    const int tid = threadIdx.x;
    if (training) {
        const int series = 2 * C, nparts = 1024 / series;
        const int sidx = tid % series, part = tid / series;
        const int nblk = (*n_rows_p + rows_per_block - 1) / rows_per_block;
        double t = 0.0;
        if (part < nparts)
            for (int b = part; b < nblk; b += nparts) t += partials[(size_t)b * series + sidx];    // fixed order: deterministic
        s[tid] = t;
        __syncthreads();
        if (tid < series) {
            double a = 0.0;
            for (int q = 0; q < nparts; ++q) a += s[q * series + tid];
            s[tid] = a;
        }
        __syncthreads();
        if (tid < C) {
            const double n = (double)(*n_total_p);
            const double mean = s[tid] / n, var = fmax(s[C + tid] / n - mean * mean, 0.0);
            stats[tid] = (float)mean; stats[C + tid] = (float)var;
        }
        __syncthreads();
    }
    if (tid < C) {
        const float mean = stats[tid], var = stats[C + tid];
        const float scale = gamma[tid] / sqrtf(var + 1e-3f);
        const float shift = beta[tid] - mean * scale;
        bnparam[tid] = scale; bnparam[C + tid] = shift; bnparam[2 * C + tid] = fmaxf(shift, 0.f);
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

extern "C" int sherf_svox_scan(const uint32_t* bitmap, int n_words, int32_t* prefix, int32_t* n_rows, int32_t* chunk_ws,
                               uint32_t* wp, sherf_stream_t stream) {
    SHERF_CHECK_ARG(bitmap && prefix && n_rows && chunk_ws && wp && n_words > 0);
    const int n_chunks = cdiv(n_words, 1024);
    hipLaunchKernelGGL(scan_local_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), bitmap, n_words, prefix, chunk_ws);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(1), dim3(1024), 0, as_stream(stream), chunk_ws, n_chunks, n_rows);
    hipLaunchKernelGGL(scan_add_kernel, dim3(cdiv(n_words, 256)), dim3(256), 0, as_stream(stream), bitmap, prefix, n_words, chunk_ws,
                       reinterpret_cast<uint2*>(wp));
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_keys(const uint32_t* bitmap, const int32_t* prefix, int n_words, int32_t* keys, sherf_stream_t stream) {
    SHERF_CHECK_ARG(bitmap && prefix && keys && n_words > 0);
    hipLaunchKernelGGL(keys_kernel, dim3(cdiv(n_words, 256)), dim3(256), 0, as_stream(stream), bitmap, prefix, n_words, keys);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_scatter_rows(const int32_t* coord, const float* feat, int n, int C, int D, int H, int W,
                                       const uint32_t* bitmap, const int32_t* prefix, const int32_t* n_rows, int64_t* acc_fix,
                                       float* g, int32_t* mult, sherf_stream_t stream) {
    SHERF_CHECK_ARG(coord && feat && bitmap && prefix && n_rows && acc_fix && g && mult && n > 0 && C > 0);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((int64_t)n * C, 256)), dim3(256), 0, as_stream(stream), coord, feat, n, C,
                       D, H, W, bitmap, prefix, reinterpret_cast<unsigned long long*>(acc_fix), mult);
    hipLaunchKernelGGL(fix_to_float_kernel, dim3(cdiv((int64_t)n * C, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const long long*>(acc_fix), n_rows, C, g);
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

extern "C" int sherf_svox_conv2(const int32_t* keys_out, const int32_t* n_rows_out, int Do, int Ho, int Wo,
                                const uint32_t* wp_in, int Di, int Hi, int Wi, const float* in_raw,
                                int Cin, const float* in_bn, const int32_t* in_mult, const float* wt, int Cout, int mode,
                                int max_rows, float* out_raw, double* partials, sherf_stream_t stream) {
    SHERF_CHECK_ARG(n_rows_out && in_raw && wt && out_raw && (mode == 2 || (keys_out && wp_in)));
    SHERF_CHECK_ARG(Cin > 0 && Cin <= 96 && Cin % 4 == 0 && (Cout == 32 || Cout == 64 || Cout == 96) && max_rows > 0 && mode >= 0 && mode <= 2);
    constexpr int NTHR = 64;                    // one wave per block: 32 / 16 / 8 rows per block -> hundreds of blocks even at 2.7k rows
    const int CQ = Cout / 4, RQ = NTHR / CQ, TM = RQ * 4, TMP = TM + 4;
    const size_t smem = (size_t)Cin * Cout * 4 + (size_t)Cin * TMP * 4 + (size_t)27 * TM * 4;
    SHERF_CHECK_ARG(partials == nullptr || Cin * TMP >= 2 * RQ * Cout);
    const dim3 grid(cdiv(max_rows, TM)), block(NTHR);
#define SHERF_CONV2(C)                                                                                                        \
    hipLaunchKernelGGL((sconv2_kernel<C, NTHR>), grid, block, smem, as_stream(stream), keys_out, n_rows_out, Do, Ho, Wo, \
                       reinterpret_cast<const uint2*>(wp_in), Di, Hi, Wi, in_raw, Cin, in_bn, in_mult, wt, mode, out_raw, partials)
    if (Cout == 32) SHERF_CONV2(32); else if (Cout == 64) SHERF_CONV2(64); else SHERF_CONV2(96);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_svox_conv2_rows_per_block(int Cout) { return (64 / (Cout / 4)) * 4; }

extern "C" int sherf_svox_bn_finalize(const double* partials, const int32_t* n_rows, const int32_t* n_total_rows, int C,
                                      int rows_per_block, const float* gamma, const float* beta, float* stats, int training,
                                      float* bnparam, sherf_stream_t stream) {
    SHERF_CHECK_ARG(partials && n_rows && n_total_rows && gamma && beta && stats && bnparam && C > 0 && C <= 96 && rows_per_block > 0);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partials, n_rows, n_total_rows, C,
                       rows_per_block, gamma, beta, stats, training, bnparam);
    SHERF_LAUNCH_CHECK();
}
