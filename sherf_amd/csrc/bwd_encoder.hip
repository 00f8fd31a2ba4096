// Backward of the sparse voxel encoder (a11): BatchNorm + ReLU backward, the weight gradient and the row scatter of the aggregation (fp32
// VALU); the input gradient runs on the forward's MFMA convolution (csrc/svox.hip: sherf_svox_conv3_dgrad), its fp32 form here is kept
// as the check.  Mirrors oracle/backward_explicit.py (_bn_relu_bwd, encoder_bwd), which is verified on the CPU against autograd and
// against the unmodified reference's gradients.  Part of libsherf_hip_bwd.so (include/sherf_hip_bwd.h).
#include "common.h"

#include "../../include/sherf_hip_bwd.h"

namespace {

// voxel (z,y,x) of `key` in a level of dims (H,W)
__device__ __forceinline__ void unkey(int key, int H, int W, int& z, int& y, int& x) { z = key / (H * W); y = (key / W) % H; x = key % W; }

// row id of voxel (z,y,x) in a level given by its (bits, prefix) records, -1 if absent / out of range
__device__ __forceinline__ int lookup(const uint2* __restrict__ wp, int D, int H, int W, int z, int y, int x) {
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return -1;
    const int k = (z * H + y) * W + x;
    const uint2 rec = wp[k >> 5];
    const uint32_t bit = 1u << (k & 31);
    return (rec.x & bit) ? (int)rec.y + __popc(rec.x & (bit - 1u)) : -1;
}

// ---- BatchNorm (batch statistics over the reference's ROW set) + ReLU backward, oracle: _bn_relu_bwd -------------------
// y = raw*scale + shift (bnparam), xh = (raw - mean) * inv with (mean, var) = stats, inv = 1/sqrt(var + 1e-3).
// sums[0][c] = sum_r d_y, sums[1][c] = sum_r d_y * xh, sums[2][c] = sum_r (mult_r - 1) * d_out    (d_y = d_out * [y > 0])
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float* __restrict__ d_out, const float* __restrict__ raw,
                                                            const float* __restrict__ bnparam, const float* __restrict__ stats,
                                                            const int32_t* __restrict__ mult, const int32_t* __restrict__ n_rows_p, int C,
                                                            float* __restrict__ sums) {
    // C <= 128: the 256 / C row groups of a workgroup walk its 256 rows interleaved (with one thread per channel, 32-channel layers kept
    // 32 of 256 threads busy), their partial sums meet in LDS, three atomics per channel per workgroup
    __shared__ float s_part[3][256];
    const int n_rows = *n_rows_p;
    const int r0 = blockIdx.x * 256, r1 = min(r0 + 256, n_rows);
    const int G = 256 / C, g = threadIdx.x / C, c = threadIdx.x % C;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (g < G) {
        const float scale = bnparam[c], shift = bnparam[C + c], mean = stats[c], inv = 1.f / sqrtf(stats[C + c] + 1e-3f);
        for (int r = r0 + g; r < r1; r += G) {
            const float x = raw[(size_t)r * C + c], d = d_out[(size_t)r * C + c];
            const float dy = (x * scale + shift > 0.f) ? d : 0.f;
            s1 += dy; s2 += dy * (x - mean) * inv;
            if (mult) s3 += (float)(mult[r] - 1) * d;
        }
    }
    s_part[0][threadIdx.x] = s1; s_part[1][threadIdx.x] = s2; s_part[2][threadIdx.x] = s3;
    __syncthreads();
    if (g == 0 && r1 > r0) {
        for (int q = 1; q < G; ++q) { s1 += s_part[0][q * C + c]; s2 += s_part[1][q * C + c]; s3 += s_part[2][q * C + c]; }
        unsafeAtomicAdd(sums + c, s1); unsafeAtomicAdd(sums + C + c, s2); unsafeAtomicAdd(sums + 2 * C + c, s3);
    }
}

// d_raw = (gamma*inv/N) * (N*d_y - s1 - xh*s2),  s1 = S1 + d_y0,  s2 = S2 + d_y0*xh0,  d_y0 = S3 * [shift > 0], xh0 = -mean*inv;
// dgamma = s2, dbeta = s1 (written by block 0).  Rows >= n_rows of d_raw are zeroed.
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ d_out, const float* __restrict__ raw,
                                                           const float* __restrict__ bnparam, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           const int32_t* __restrict__ n_total_p, const int32_t* __restrict__ n_rows_p,
                                                           int64_t cap, int C, float* __restrict__ d_raw, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, uint32_t* __restrict__ amax) {
    // 32 channel lanes x 8 row lanes, 8 rows per thread (C is a multiple of 32 here; other widths leave lanes idle): the channel's
    // constants are formed once per thread instead of once per element, and no element needs a 64-bit division.
    // grid = (row groups of 64, channel groups of 32).
    const int c = blockIdx.y * 32 + (threadIdx.x & 31);
    const bool on = c < C;
    const int cc = on ? c : 0;
    const float scale = bnparam[cc], shift = bnparam[C + cc], mean = stats[cc], inv = 1.f / sqrtf(stats[C + cc] + 1e-3f);
    const float dy0 = shift > 0.f ? sums[2 * C + cc] : 0.f, xh0 = -mean * inv;
    const float s1 = sums[cc] + dy0, s2 = sums[C + cc] + dy0 * xh0;
    if (on && blockIdx.x == 0 && threadIdx.x < 32) { dgamma[c] = s2; dbeta[c] = s1; }
    const int n_rows = *n_rows_p;
    const float N = (float)(*n_total_p);
    const float k = gamma[cc] * inv / N;
    float m = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 5);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t r = r0 + 8 * q;
        if (!on || r >= cap) continue;
        float out = 0.f;
        if (r < n_rows) {
            const float x = raw[r * C + c];
            const float dy = (x * scale + shift > 0.f) ? d_out[r * C + c] : 0.f;
            out = k * (N * dy - s1 - (x - mean) * inv * s2);
        }
        d_raw[r * C + c] = out;
        const float a = fabsf(out);
        if (a <= 3.0e38f) m = fmaxf(m, a);           // (a non-finite gradient must not pick the scale; it still propagates as itself)
    }
    // max |d_raw| of the layer (bit pattern; non-negative floats order like unsigned integers): the scale of the MFMA input-gradient
    // convolution that reads d_raw next (sherf_svox_conv3_dgrad).  One atomic per wave that holds a new candidate.
    if (amax) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0 && m > 0.f && __float_as_uint(m) > *reinterpret_cast<volatile uint32_t*>(amax)) atomicMax(amax, __float_as_uint(m));
    }
}

// ---- sparse conv backward w.r.t. the INPUT: d_in[i][ci] = sum_k sum_co d_raw[o(i,k)][co] * W[co][k][ci] ------------------
// mode 0 (submanifold): o(i,k) = the row at offset (1-k) from i in the same level;
// mode 1 (stride 2):    i is a fine voxel q, o = (q + 1 - k) / 2 in the coarse level when every component is even and >= 0.
// W is the reference weight viewed [Cout][27][Cin].  One workgroup = 32 receiving rows; thread t owns columns ci = t % Cin of
// rows t / Cin + j * (256 / Cin)  (Cin in {32, 64, 96} -> 8, 4, 2 rows at a time; 96: 64 threads idle).
__global__ void __launch_bounds__(256) conv_dgrad_kernel(const int32_t* __restrict__ keys_i, const int32_t* __restrict__ n_rows_i, int Di, int Hi,
                                                         int Wi, const uint2* __restrict__ wp_o, int Do, int Ho, int Wo,
                                                         const float* __restrict__ d_raw, int Cout, const float* __restrict__ W, int Cin,
                                                         int mode, float* __restrict__ d_in) {
    extern __shared__ float smem[];
    int* s_nb = reinterpret_cast<int*>(smem);             // [27][32]
    float* s_d = smem + 27 * 32;                          // [32][Cout]
    const int n_rows = *n_rows_i;
    const int row0 = blockIdx.x * 32;
    if (row0 >= n_rows) return;
    for (int i = threadIdx.x; i < 27 * 32; i += 256) {
        const int tap = i >> 5, r = i & 31, row = row0 + r;
        int nb = -1;
        if (row < n_rows) {
            int z, y, x;
            unkey(keys_i[row], Hi, Wi, z, y, x);
            const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
            const int nz = z + 1 - kz, ny = y + 1 - ky, nx = x + 1 - kx;
            if (mode == 0) nb = lookup(wp_o, Do, Ho, Wo, nz, ny, nx);
            else if (nz >= 0 && ny >= 0 && nx >= 0 && !((nz | ny | nx) & 1)) nb = lookup(wp_o, Do, Ho, Wo, nz >> 1, ny >> 1, nx >> 1);
        }
        s_nb[i] = nb;
    }
    const int rpp = 256 / Cin;                            // rows per pass
    const int ci = threadIdx.x % Cin, rsub = threadIdx.x / Cin;
    const bool active = rsub < rpp;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int tap = 0; tap < 27; ++tap) {
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * Cout; i += 256) {
            const int r = i / Cout, co = i % Cout;
            const int nb = s_nb[tap * 32 + r];
            s_d[i] = nb >= 0 ? d_raw[(size_t)nb * Cout + co] : 0.f;
        }
        __syncthreads();
        if (!active) continue;
        for (int co = 0; co < Cout; ++co) {
            const float w = W[((size_t)co * 27 + tap) * Cin + ci];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int r = rsub + j * rpp;
                if (r < 32) acc[j] += s_d[r * Cout + co] * w;
            }
        }
    }
    if (active)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = rsub + j * rpp;
            if (r < 32 && row0 + r < n_rows) d_in[(size_t)(row0 + r) * Cin + ci] = acc[j];
        }
}

// ---- sparse conv backward w.r.t. the WEIGHT: dW[co][k][ci] += sum_o d_raw[o][co] * act(in[nb(o,k)][ci]) --------------------
// forward orientation (same neighbour rule as sconv3_kernel): mode 0 nb = row at offset (k-1) in the same level, mode 1
// nb = fine voxel 2*o + k - 1.  act = BatchNorm+ReLU of the producer applied on the fly (+ (mult-1)*relu(shift) at level 0),
// identity when in_bn == nullptr.  grid = (27 taps, row splits); thread t owns entries e = t, t+256, ... of [Cout][Cin].
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const int32_t* __restrict__ keys_o, const int32_t* __restrict__ n_rows_o, int Do, int Ho,
                                                         int Wo, const uint2* __restrict__ wp_i, int Di, int Hi, int Wi,
                                                         const float* __restrict__ in_raw, int Cin, const float* __restrict__ in_bn,
                                                         const int32_t* __restrict__ in_mult, const float* __restrict__ d_raw, int Cout, int mode,
                                                         float* __restrict__ dW) {
    extern __shared__ float smem[];
    float* s_x = smem;                                    // [32][Cin]
    float* s_d = smem + 32 * Cin;                         // [32][Cout]
    int* s_nb = reinterpret_cast<int*>(s_d + 32 * Cout);  // [32]
    const int tap = blockIdx.x;
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    const int n_rows = *n_rows_o;
    const int n_chunks = (n_rows + 31) / 32;
    float acc[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) acc[j] = 0.f;
    const int n_ent = Cin * Cout;
    for (int chunk = blockIdx.y; chunk < n_chunks; chunk += gridDim.y) {
        __syncthreads();
        if (threadIdx.x < 32) {
            const int row = chunk * 32 + threadIdx.x;
            int nb = -1;
            if (row < n_rows) {
                int z, y, x;
                unkey(keys_o[row], Ho, Wo, z, y, x);
                nb = mode ? lookup(wp_i, Di, Hi, Wi, 2 * z + kz - 1, 2 * y + ky - 1, 2 * x + kx - 1)
                          : lookup(wp_i, Di, Hi, Wi, z + kz - 1, y + ky - 1, x + kx - 1);
            }
            s_nb[threadIdx.x] = nb;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * Cin; i += 256) {
            const int r = i / Cin, c = i % Cin;
            const int nb = s_nb[r];
            float v = 0.f;
            if (nb >= 0) {
                v = in_raw[(size_t)nb * Cin + c];
                if (in_bn) v = fmaxf(v * in_bn[c] + in_bn[Cin + c], 0.f) + (in_mult ? (float)(in_mult[nb] - 1) * in_bn[2 * Cin + c] : 0.f);
            }
            s_x[i] = v;
        }
        for (int i = threadIdx.x; i < 32 * Cout; i += 256) {
            const int r = i / Cout, row = chunk * 32 + r;
            s_d[i] = (row < n_rows && s_nb[r] >= 0) ? d_raw[(size_t)row * Cout + i % Cout] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 36; ++j) {
            const int e = threadIdx.x + 256 * j;
            if (e < n_ent) {
                const int co = e / Cin, c = e % Cin;
                float a = 0.f;
                for (int r = 0; r < 32; ++r) a += s_d[r * Cout + co] * s_x[r * Cin + c];
                acc[j] += a;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 36; ++j) {
        const int e = threadIdx.x + 256 * j;
        if (e < n_ent) unsafeAtomicAdd(dW + ((size_t)(e / Cin) * 27 + tap) * Cin + e % Cin, acc[j]);
    }
}

// The same for the channel pairs whose Cin divides 256 (compile-time CIN, COUT): entry e = t + 256 j of thread t has the SAME input
// channel c = t % CIN for every j (co = t / CIN + (256 / CIN) j), so a row's x is read from LDS once and reused for all NJ = CIN COUT / 256
// entries of the thread -- one LDS read per product instead of two (the generic kernel, rows innermost, is LDS-bandwidth bound) -- and a
// thread holds NJ accumulators instead of 36 (4 at 32 -> 32: more workgroups per CU).
template <int CIN, int COUT>
__global__ void __launch_bounds__(256) conv_wgrad_fast_kernel(const int32_t* __restrict__ keys_o, const int32_t* __restrict__ n_rows_o, int Do, int Ho,
                                                              int Wo, const uint2* __restrict__ wp_i, int Di, int Hi, int Wi,
                                                              const float* __restrict__ in_raw, const float* __restrict__ in_bn,
                                                              const int32_t* __restrict__ in_mult, const float* __restrict__ d_raw, int mode,
                                                              float* __restrict__ dW) {
    static_assert(256 % CIN == 0 && (CIN * COUT) % 256 == 0, "channel pair not covered by the fast kernel");
    constexpr int NJ = CIN * COUT / 256, CSTEP = 256 / CIN;
    __shared__ float s_x[32 * CIN], s_d[32 * COUT];
    __shared__ int s_nb[32];
    const int tap = blockIdx.x;
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    const int n_rows = *n_rows_o;
    const int n_chunks = (n_rows + 31) / 32;
    const int c = threadIdx.x % CIN, co0 = threadIdx.x / CIN;
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    for (int chunk = blockIdx.y; chunk < n_chunks; chunk += gridDim.y) {
        __syncthreads();
        if (threadIdx.x < 32) {
            const int row = chunk * 32 + threadIdx.x;
            int nb = -1;
            if (row < n_rows) {
                int z, y, x;
                unkey(keys_o[row], Ho, Wo, z, y, x);
                nb = mode ? lookup(wp_i, Di, Hi, Wi, 2 * z + kz - 1, 2 * y + ky - 1, 2 * x + kx - 1)
                          : lookup(wp_i, Di, Hi, Wi, z + kz - 1, y + ky - 1, x + kx - 1);
            }
            s_nb[threadIdx.x] = nb;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * CIN; i += 256) {
            const int r = i / CIN, ci = i % CIN;
            const int nb = s_nb[r];
            float v = 0.f;
            if (nb >= 0) {
                v = in_raw[(size_t)nb * CIN + ci];
                if (in_bn) v = fmaxf(v * in_bn[ci] + in_bn[CIN + ci], 0.f) + (in_mult ? (float)(in_mult[nb] - 1) * in_bn[2 * CIN + ci] : 0.f);
            }
            s_x[i] = v;
        }
        for (int i = threadIdx.x; i < 32 * COUT; i += 256) {
            const int r = i / COUT, row = chunk * 32 + r;
            s_d[i] = (row < n_rows && s_nb[r] >= 0) ? d_raw[(size_t)row * COUT + i % COUT] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
            const float x = s_x[r * CIN + c];
            const float* dr = s_d + r * COUT + co0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] += dr[CSTEP * j] * x;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) unsafeAtomicAdd(dW + ((size_t)(co0 + CSTEP * j) * 27 + tap) * CIN + c, acc[j]);
}

// level-0 aggregation backward: every input row gets the gradient of the voxel it was summed into
__global__ void __launch_bounds__(256) gather_rows_kernel(const int32_t* __restrict__ coord, int n, int D, int H, int W,
                                                          const uint2* __restrict__ wp, const float* __restrict__ d_g, int C,
                                                          float* __restrict__ d_feat) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    const int row = lookup(wp, D, H, W, coord[r * 4 + 1], coord[r * 4 + 2], coord[r * 4 + 3]);
    d_feat[i] = row >= 0 ? d_g[(size_t)row * C + c] : 0.f;
}

}  // namespace

extern "C" int sherf_bwd_bn_relu(const float* d_out, const float* raw, const float* bnparam, const float* stats, const float* gamma,
                                 const int32_t* mult, const int32_t* n_total, const int32_t* n_rows, int64_t cap, int C, float* sums,
                                 float* d_raw, float* dgamma, float* dbeta, uint32_t* amax, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d_out && raw && bnparam && stats && gamma && n_total && n_rows && sums && d_raw && dgamma && dbeta && cap > 0 && C > 0 && C <= 256);
    SHERF_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)3 * C * sizeof(float), as_stream(stream)));
    if (amax) SHERF_HIP_CHECK(hipMemsetAsync(amax, 0, sizeof(uint32_t), as_stream(stream)));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, as_stream(stream), d_out, raw, bnparam, stats,
                       mult, n_rows, C, sums);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((cap + 63) / 64), (unsigned)((C + 31) / 32)), dim3(256), 0, as_stream(stream), d_out, raw,
                       bnparam, stats, gamma, sums, n_total, n_rows, cap, C, d_raw, dgamma, dbeta, amax);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_conv_dgrad(const int32_t* keys_i, const int32_t* n_rows_i, int Di, int Hi, int Wi, const uint32_t* wp_o, int Do,
                                    int Ho, int Wo, const float* d_raw, int Cout, const float* W, int Cin, int mode, int max_rows,
                                    float* d_in, sherf_stream_t stream) {
    SHERF_CHECK_ARG(keys_i && n_rows_i && wp_o && d_raw && W && d_in && max_rows > 0 && (mode == 0 || mode == 1));
    SHERF_CHECK_ARG((Cin == 32 || Cin == 64 || Cin == 96) && Cout > 0 && Cout <= 96);
    const size_t smem = (size_t)27 * 32 * 4 + (size_t)32 * Cout * 4;
    hipLaunchKernelGGL(conv_dgrad_kernel, dim3((max_rows + 31) / 32), dim3(256), smem, as_stream(stream), keys_i, n_rows_i, Di, Hi, Wi,
                       reinterpret_cast<const uint2*>(wp_o), Do, Ho, Wo, d_raw, Cout, W, Cin, mode, d_in);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_conv_wgrad(const int32_t* keys_o, const int32_t* n_rows_o, int Do, int Ho, int Wo, const uint32_t* wp_i, int Di,
                                    int Hi, int Wi, const float* in_raw, int Cin, const float* in_bn, const int32_t* in_mult,
                                    const float* d_raw, int Cout, int mode, int max_rows, float* dW, sherf_stream_t stream) {
    SHERF_CHECK_ARG(keys_o && n_rows_o && wp_i && in_raw && d_raw && dW && max_rows > 0 && (mode == 0 || mode == 1));
    SHERF_CHECK_ARG(Cin > 0 && Cout > 0 && Cin * Cout <= 36 * 256);
    const size_t smem = (size_t)32 * (Cin + Cout) * 4 + 32 * 4;
    const int splits = max_rows / 32 / 8 > 0 ? (max_rows / 32 / 8 < 64 ? max_rows / 32 / 8 : 64) : 1;
#define SHERF_WGRAD_FAST(CI, CO)                                                                                                       \
    if (Cin == CI && Cout == CO) {                                                                                                     \
        hipLaunchKernelGGL((conv_wgrad_fast_kernel<CI, CO>), dim3(27, splits), dim3(256), 0, as_stream(stream), keys_o, n_rows_o, Do, Ho, Wo, \
                           reinterpret_cast<const uint2*>(wp_i), Di, Hi, Wi, in_raw, in_bn, in_mult, d_raw, mode, dW);                   \
        SHERF_LAUNCH_CHECK();                                                                                                          \
    }
    SHERF_WGRAD_FAST(32, 32) SHERF_WGRAD_FAST(32, 64) SHERF_WGRAD_FAST(64, 64) SHERF_WGRAD_FAST(64, 96)
#undef SHERF_WGRAD_FAST
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(27, splits), dim3(256), smem, as_stream(stream), keys_o, n_rows_o, Do, Ho, Wo,
                       reinterpret_cast<const uint2*>(wp_i), Di, Hi, Wi, in_raw, Cin, in_bn, in_mult, d_raw, Cout, mode, dW);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_gather_rows(const int32_t* coord, int n, int D, int H, int W, const uint32_t* wp, const float* d_g, int C,
                                     float* d_feat, sherf_stream_t stream) {
    SHERF_CHECK_ARG(coord && wp && d_g && d_feat && n > 0 && C > 0);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(((int64_t)n * C + 255) / 256)), dim3(256), 0, as_stream(stream), coord, n, D, H, W,
                       reinterpret_cast<const uint2*>(wp), d_g, C, d_feat);
    SHERF_LAUNCH_CHECK();
}
