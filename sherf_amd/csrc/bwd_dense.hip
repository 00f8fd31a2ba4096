// Dense building blocks of the backward path (gfx950) -- see include/sherf_hip_bwd.h; mirrors oracle/backward_explicit.py.  fp32
// element-wise / reduction kernels; the GEMMs are the MFMA kernels of csrc/bwd_gemm.hip.  First run on an MI355X in round 2
// (tests/test_gpu_backward.py, green).
#include "common.h"

#include "../../include/sherf_hip_bwd.h"

char g_sherf_err[256] = "";        // this library's own error buffer (common.h macros)
int g_sherf_debug = 0;

namespace {


#define SHERF_GRID(count) dim3((unsigned)((count + 255) / 256)), dim3(256)

__global__ void untile_kernel(const float4* __restrict__ tokens, const float* __restrict__ extras, int64_t n, float* __restrict__ tok,
                              float* __restrict__ ext) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;           // one thread per (sample, slot, quad) = n * 24
    if (i >= n * 24) return;
    const int64_t c = i / 24;
    const int q = (int)(i % 24), s = q / 8, l = q % 8;
    const int64_t tile = c >> 5;
    const int j = (int)(c & 31);
    const float4 v = tokens[((tile * 3 + s) * 8 + l) * 32 + j];
    *reinterpret_cast<float4*>(tok + c * 96 + 32 * s + 4 * l) = v;
    if (q < 12) ext[c * 12 + q] = extras[(tile * 12 + q) * 32 + j];
}

__global__ void tile_kernel(const float* __restrict__ d_tok, int64_t n, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;           // over padded samples * 24
    const int64_t npad = ((n + 31) / 32) * 32;
    if (i >= npad * 24) return;
    const int64_t c = i / 24;
    const int q = (int)(i % 24), s = q / 8, l = q % 8;
    const float4 v = c < n ? *reinterpret_cast<const float4*>(d_tok + c * 96 + 32 * s + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
    out[(((c >> 5) * 3 + s) * 8 + l) * 32 + (c & 31)] = v;
}

__global__ void bias_act_kernel(float* __restrict__ y, int ldy, const float* __restrict__ bias, int64_t n, int C, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * C) return;
    const int64_t r = i / C;
    const int c = (int)(i % C);
    float v = y[r * ldy + c] + (bias ? bias[c] : 0.f);
    y[r * ldy + c] = act == 1 ? fmaxf(v, 0.f) : v;
}

__global__ void relu_mask_kernel(float* __restrict__ d, int ldd, const float* __restrict__ h, int ldh, int64_t n, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * C) return;
    const int64_t r = i / C;
    const int c = (int)(i % C);
    if (!(h[r * ldh + c] > 0.f)) d[r * ldd + c] = 0.f;
}

// each block sums 512 rows; thread t owns columns t, t+256, ...
__global__ void colsum_kernel(const float* __restrict__ d, int ldd, int64_t n, int C, float* __restrict__ out) {
    const int64_t r0 = (int64_t)blockIdx.x * 512, r1 = r0 + 512 < n ? r0 + 512 : n;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        int64_t r = r0;
        for (; r + 7 < r1; r += 8) {                 // eight independent loads in flight
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = d[(r + q) * ldd + c];
#pragma unroll
            for (int q = 0; q < 8; ++q) a += v[q];
        }
        for (; r < r1; ++r) a += d[r * ldd + c];
        unsafeAtomicAdd(out + c, a);
    }
}

// d *= [h > 0] and out[c] += sum_r d[r][c] in one pass (the masked gradient is the next layer's d_out: its column sums are that
// layer's bias gradient).  C divides 256: the 256 / C row groups of a workgroup walk its 512 rows interleaved.
__global__ void __launch_bounds__(256) relu_mask_colsum_kernel(float* __restrict__ d, int ldd, const float* __restrict__ h, int ldh, int64_t n, int C,
                                                               float* __restrict__ out) {
    __shared__ float s_part[256];
    const int G = 256 / C, g = threadIdx.x / C, c = threadIdx.x % C;
    const int64_t r0 = (int64_t)blockIdx.x * 512, r1 = r0 + 512 < n ? r0 + 512 : n;
    float a = 0.f;
    int64_t r = r0 + g;
    for (; r + 3 * G < r1; r += 4 * G) {             // four rows in flight per thread (one at a time the loop was latency bound: 2.5 TB/s)
        float v[4], hv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = d[(r + q * G) * ldd + c]; hv[q] = h[(r + q * G) * ldh + c]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!(hv[q] > 0.f)) { v[q] = 0.f; d[(r + q * G) * ldd + c] = 0.f; }
            a += v[q];
        }
    }
    for (; r < r1; r += G) {
        float v = d[r * ldd + c];
        if (!(h[r * ldh + c] > 0.f)) { v = 0.f; d[r * ldd + c] = 0.f; }
        a += v;
    }
    s_part[threadIdx.x] = a;
    __syncthreads();
    if (g == 0) {
        for (int q = 1; q < G; ++q) a += s_part[q * C + c];
        unsafeAtomicAdd(out + c, a);
    }
}

// 2^cl column lanes x (256 >> cl) row lanes per workgroup, 16 rows per thread: no per-element 64-bit division (the flat-index form spent
// more on i / C than on the copy); cl is chosen by the launcher so that narrow matrices ([n, 3], [n, 32]) still fill their waves.
// grid = (row groups of 16 * (256 >> cl), column groups of 2^cl).
__global__ void __launch_bounds__(256) copy2d_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int64_t n, int C, int add,
                                                     int cl) {
    const int c = (blockIdx.y << cl) + (threadIdx.x & ((1 << cl) - 1));
    if (c >= C) return;
    const int rl = 256 >> cl;
    const int64_t r0 = (int64_t)blockIdx.x * 16 * rl + (threadIdx.x >> cl);
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int64_t r = r0 + (int64_t)q * rl;
        if (r >= n) break;
        const float v = src[r * lds + c];
        if (add) dst[r * ldd + c] += v; else dst[r * ldd + c] = v;
    }
}

// out[r] = [x(3), sin(2^0 x)(3), sin(2^0 x + pi/2)(3), sin(2^1 x)(3), ...]  (renderer.py:875-916)
__global__ void pe_kernel(const float* __restrict__ in, int ldi, int64_t n, int NF, float* __restrict__ out, int ldo) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    float x[3] = {in[r * ldi], in[r * ldi + 1], in[r * ldi + 2]};
    float* o = out + r * ldo;
    o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
    float f = 1.f;
    for (int q = 0; q < NF; ++q, f *= 2.f)
        for (int a = 0; a < 3; ++a) {
            o[3 + 6 * q + a] = sinf(f * x[a]);
            o[6 + 6 * q + a] = sinf(f * x[a] + 1.57079632679489662f);
        }
}

__global__ void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, int64_t rows,
                              float* __restrict__ y, float* __restrict__ xh, float* __restrict__ inv) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float v[32], mu = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) { v[c] = x[r * 32 + c]; mu += v[c]; }
    mu *= (1.f / 32.f);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) { v[c] -= mu; q += v[c] * v[c]; }
    const float iv = 1.f / sqrtf(q * (1.f / 32.f) + 1e-5f);
    inv[r] = iv;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const float h = v[c] * iv;
        xh[r * 32 + c] = h;
        y[r * 32 + c] = h * w[c] + b[c];
    }
}

// d_x = inv * (d_xh - mean(d_xh) - xh * mean(d_xh * xh)),  d_xh = d_y * w;  dw += d_y * xh, db += d_y
// Eight lanes x float4 per 32-wide row (coalesced 128-byte rows; the two row means are three xor-shuffles), the parameter gradients
// kept in registers over the workgroup's rows and reduced once at the end: lanes of equal channel quad by shuffles, the four waves
// through LDS, one global atomic per channel per workgroup.  (Round 2's form -- one thread per row, 64 LDS atomics per thread on the
// same 32 words -- took 2.35 ms per call on 2 M rows, 26x the time of its memory traffic: profiles/r02_train_step_final_rocprofv3_stats.txt.)
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ xh,
                                                     const float* __restrict__ inv, int64_t rows, float* __restrict__ dx, float* __restrict__ dw,
                                                     float* __restrict__ db, const float* __restrict__ addend) {
    __shared__ float s_red[4][8][8];
    const int l = threadIdx.x & 7, sub = threadIdx.x >> 3;
    const float4 wv = reinterpret_cast<const float4*>(w)[l];
    const float4* dy4 = reinterpret_cast<const float4*>(dy);
    const float4* xh4 = reinterpret_cast<const float4*>(xh);
    float4* dx4 = reinterpret_cast<float4*>(dx);
    float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r0 = (int64_t)blockIdx.x * 32; r0 < rows; r0 += (int64_t)gridDim.x * 32) {      // (uniform trip count: shuffles inside)
        const int64_t r = r0 + sub;
        const bool live = r < rows;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 d = live ? dy4[r * 8 + l] : z, h = live ? xh4[r * 8 + l] : z;
        aw[0] += d.x * h.x; aw[1] += d.y * h.y; aw[2] += d.z * h.z; aw[3] += d.w * h.w;
        ab[0] += d.x; ab[1] += d.y; ab[2] += d.z; ab[3] += d.w;
        const float4 g = make_float4(d.x * wv.x, d.y * wv.y, d.z * wv.z, d.w * wv.w);
        float m1 = (g.x + g.y) + (g.z + g.w), m2 = (g.x * h.x + g.y * h.y) + (g.z * h.z + g.w * h.w);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
        m1 *= (1.f / 32.f); m2 *= (1.f / 32.f);
        if (live) {
            const float iv = inv[r];
            float4 o = make_float4(iv * (g.x - m1 - h.x * m2), iv * (g.y - m1 - h.y * m2), iv * (g.z - m1 - h.z * m2), iv * (g.w - m1 - h.w * m2));
            if (addend) { const float4 a = reinterpret_cast<const float4*>(addend)[r * 8 + l]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }      // (uniform)
            dx4[r * 8 + l] = o;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { aw[q] += __shfl_xor(aw[q], o); ab[q] += __shfl_xor(ab[q], o); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { s_red[wave][lane][q] = aw[q]; s_red[wave][lane][4 + q] = ab[q]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                       // thread -> (which = t >> 5, channel c = t & 31)
        const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
        const float t = (s_red[0][c >> 2][4 * which + (c & 3)] + s_red[1][c >> 2][4 * which + (c & 3)]) +
                        (s_red[2][c >> 2][4 * which + (c & 3)] + s_red[3][c >> 2][4 * which + (c & 3)]);
        unsafeAtomicAdd((which ? db : dw) + c, t);
    }
}

// sixteen consecutive floats of a 64-byte aligned group (a head's slice of q / k / v / o: rows of 432 / 144 floats, offsets multiples of 16) as
// four dwordx4 accesses: the scalar form cost 200 dword loads per thread, every one touching 64 different lines (round 5)
__device__ __forceinline__ void ld16(const float* __restrict__ p, float (&d)[16]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float4 v = q[j]; d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w; }
}
__device__ __forceinline__ void st16(float* __restrict__ p, const float (&d)[16]) {
    float4* q = reinterpret_cast<float4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = make_float4(d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]);
}

// one thread per (sample, head): q_i = qkv[n][i][16h..], k_j = qkv[n][j][48 + 16h..], v_j = qkv[n][j][96 + 16h..]
__global__ void __launch_bounds__(256) attn_fwd_kernel(const float* __restrict__ qkv, int64_t n, float* __restrict__ att, float* __restrict__ o) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 3) return;
    const int64_t s = i / 3;
    const int h = (int)(i % 3);
    const float* base = qkv + s * 432 + 16 * h;
    float q[3][16], k[3][16], v[3][16];
#pragma unroll
    for (int t = 0; t < 3; ++t) { ld16(base + t * 144, q[t]); ld16(base + t * 144 + 48, k[t]); ld16(base + t * 144 + 96, v[t]); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float sc[3], m = -3.0e38f;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) d += q[a][e] * k[b][e];
            sc[b] = d * 0.25f;
            m = fmaxf(m, sc[b]);
        }
        float e0 = expf(sc[0] - m), e1 = expf(sc[1] - m), e2 = expf(sc[2] - m);
        const float iv = 1.f / (e0 + e1 + e2);
        e0 *= iv; e1 *= iv; e2 *= iv;
        float* ap = att + ((s * 3 + h) * 3 + a) * 3;
        ap[0] = e0; ap[1] = e1; ap[2] = e2;
        float ov[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) ov[e] = e0 * v[0][e] + e1 * v[1][e] + e2 * v[2][e];
        st16(o + (s * 3 + a) * 48 + 16 * h, ov);
    }
}

// (launch bounds: without them the compiler budgets for 1024-thread workgroups, 128 registers, and spilled q / k / v / d_o -- 192 floats per thread)
__global__ void __launch_bounds__(256) attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ att, const float* __restrict__ d_o, int64_t n,
                                float* __restrict__ d_qkv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 3) return;
    const int64_t s = i / 3;
    const int h = (int)(i % 3);
    const float* base = qkv + s * 432 + 16 * h;
    float q[3][16], k[3][16], v[3][16], go[3][16], A[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        ld16(base + t * 144, q[t]); ld16(base + t * 144 + 48, k[t]); ld16(base + t * 144 + 96, v[t]);
        ld16(d_o + (s * 3 + t) * 48 + 16 * h, go[t]);
#pragma unroll
        for (int b = 0; b < 3; ++b) A[t][b] = att[((s * 3 + h) * 3 + t) * 3 + b];
    }
    float dS[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float dA[3], dot = 0.f;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) d += go[a][e] * v[b][e];
            dA[b] = d;
            dot += d * A[a][b];
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) dS[a][b] = A[a][b] * (dA[b] - dot) * 0.25f;
    }
    float* ob = d_qkv + s * 432 + 16 * h;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float dq[16], dk[16], dv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dq[e] = dS[t][0] * k[0][e] + dS[t][1] * k[1][e] + dS[t][2] * k[2][e];                    // d_q[t]
            dk[e] = dS[0][t] * q[0][e] + dS[1][t] * q[1][e] + dS[2][t] * q[2][e];                    // d_k[t]
            dv[e] = A[0][t] * go[0][e] + A[1][t] * go[1][e] + A[2][t] * go[2][e];                    // d_v[t]
        }
        st16(ob + t * 144, dq); st16(ob + t * 144 + 48, dk); st16(ob + t * 144 + 96, dv);
    }
}

__global__ void gelu_fwd_kernel(const float* __restrict__ u, int64_t count, float* __restrict__ ge) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) ge[i] = 0.5f * u[i] * (1.f + erff(u[i] * 0.70710678118654752f));
}

__global__ void gelu_bwd_kernel(float* __restrict__ d, const float* __restrict__ u, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float x = u[i];
    d[i] *= 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.39894228040143268f;
}

__global__ void rgb_fwd_kernel(float* __restrict__ lin, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) lin[i] = (1.f / (1.f + expf(-lin[i]))) * 1.002f - 0.001f;
}

__global__ void rgb_bwd_kernel(float* __restrict__ d, const float* __restrict__ rgb, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float s = (rgb[i] + 0.001f) * (1.f / 1.002f);
    d[i] *= 1.002f * s * (1.f - s);
}

// transpose of fold32_kernel (csrc/fold.hip): d_in[g*32 + c][pix] = sum_o W[o][c] d_f[g][pix][o];
// dW[o][c] += sum_{g,pix} d_f[g][pix][o] * in[g*32 + c][pix]  (per-block LDS accumulation, then 1024 atomics per block)
__global__ void __launch_bounds__(256) unfold32_kernel(const float* __restrict__ d_f, const float* __restrict__ W, const float* __restrict__ in,
                                                       int HW, int pix_stride, int64_t group_base, float* __restrict__ d_in,
                                                       float* __restrict__ dW) {
    __shared__ float s_w[32 * 32];          // [o][c]
    __shared__ float s_d[64][33], s_x[64][33];
    for (int i = threadIdx.x; i < 1024; i += 256) s_w[i] = W[i];
    __syncthreads();
    const int g = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    float d[32], x[32];
    const bool live = pix < HW;
    if (live) {
        const float4* dp = reinterpret_cast<const float4*>(d_f + (size_t)g * group_base + (size_t)pix * pix_stride);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 v = dp[q]; d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w; }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            x[c] = in[((size_t)g * 32 + c) * HW + pix];
            float a = 0.f;
#pragma unroll
            for (int o = 0; o < 32; ++o) a += s_w[o * 32 + c] * d[o];
            d_in[((size_t)g * 32 + c) * HW + pix] = a;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) { d[c] = 0.f; x[c] = 0.f; }
    }
    // dW: thread t owns entries (o = t / 8, c = 4 * (t % 8) .. +4); pixels staged 64 at a time
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int o_ = threadIdx.x >> 3, c0 = 4 * (threadIdx.x & 7);
    for (int part = 0; part < 4; ++part) {
        __syncthreads();
        if ((threadIdx.x >> 6) == part) {
            const int r = threadIdx.x & 63;
#pragma unroll
            for (int c = 0; c < 32; ++c) { s_d[r][c] = d[c]; s_x[r][c] = x[c]; }
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r) {
            const float dv = s_d[r][o_];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += dv * s_x[r][c0 + e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dW + o_ * 32 + c0 + e, acc[e]);
}

// act[r][c] = relu(raw[r][c] * scale[c] + shift[c]) for r < *n_rows, 0 beyond (bnparam = [scale | shift | relu(shift)])
__global__ void bn_relu_apply_kernel(const float* __restrict__ raw, const float* __restrict__ bnparam, const int32_t* __restrict__ n_rows,
                                     int64_t cap, int C, float* __restrict__ act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cap * C) return;
    const int64_t r = i / C;
    const int c = (int)(i % C);
    act[i] = r < *n_rows ? fmaxf(raw[i] * bnparam[c] + bnparam[C + c], 0.f) : 0.f;
}

}  // namespace

extern "C" int sherf_bwd_untile(const float* tokens_tiled, const float* extras_tiled, int64_t n, float* tok, float* ext,
                                sherf_stream_t stream) {
    SHERF_CHECK_ARG(tokens_tiled && extras_tiled && tok && ext && n > 0);
    hipLaunchKernelGGL(untile_kernel, SHERF_GRID(n * 24), 0, as_stream(stream), reinterpret_cast<const float4*>(tokens_tiled), extras_tiled, n,
                       tok, ext);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_tile_tokens(const float* d_tok, int64_t n, float* d_tokens_tiled, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d_tok && d_tokens_tiled && n > 0);
    const int64_t npad = ((n + 31) / 32) * 32;
    hipLaunchKernelGGL(tile_kernel, SHERF_GRID(npad * 24), 0, as_stream(stream), d_tok, n, reinterpret_cast<float4*>(d_tokens_tiled));
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_bias_act(float* y, int ldy, const float* bias, int64_t n, int C, int act, sherf_stream_t stream) {
    SHERF_CHECK_ARG(y && n > 0 && C > 0 && ldy >= C && (act == 0 || act == 1));
    hipLaunchKernelGGL(bias_act_kernel, SHERF_GRID(n * C), 0, as_stream(stream), y, ldy, bias, n, C, act);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_relu_mask(float* d, int ldd, const float* h, int ldh, int64_t n, int C, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d && h && n > 0 && C > 0 && ldd >= C && ldh >= C);
    hipLaunchKernelGGL(relu_mask_kernel, SHERF_GRID(n * C), 0, as_stream(stream), d, ldd, h, ldh, n, C);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_relu_mask_colsum(float* d, int ldd, const float* h, int ldh, int64_t n, int C, float* out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d && h && out && n > 0 && C > 0 && C <= 256 && 256 % C == 0 && ldd >= C && ldh >= C);
    hipLaunchKernelGGL(relu_mask_colsum_kernel, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, as_stream(stream), d, ldd, h, ldh, n, C, out);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_colsum(const float* d, int ldd, int64_t n, int C, float* out, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d && out && n > 0 && C > 0 && ldd >= C);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, as_stream(stream), d, ldd, n, C, out);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_copy2d(float* dst, int ldd, const float* src, int lds, int64_t n, int C, int add, sherf_stream_t stream) {
    SHERF_CHECK_ARG(dst && src && n > 0 && C > 0 && ldd >= C && lds >= C);
    int cl = 2;                                     // column lanes: the smallest power of two >= min(C, 64), at least 4
    while ((1 << cl) < C && cl < 6) ++cl;
    const int rows_per_wg = 16 * (256 >> cl);
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg), (unsigned)((C + (1 << cl) - 1) >> cl)), dim3(256), 0,
                       as_stream(stream), dst, ldd, src, lds, n, C, add, cl);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_pe(const float* in, int ldi, int64_t n, int NF, float* out, int ldo, sherf_stream_t stream) {
    SHERF_CHECK_ARG(in && out && n > 0 && NF > 0 && NF <= 10 && ldi >= 3 && ldo >= 3 + 6 * NF);
    hipLaunchKernelGGL(pe_kernel, SHERF_GRID(n), 0, as_stream(stream), in, ldi, n, NF, out, ldo);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_ln_fwd(const float* x, const float* w, const float* b, int64_t rows, float* y, float* xh, float* inv,
                                sherf_stream_t stream) {
    SHERF_CHECK_ARG(x && w && b && y && xh && inv && rows > 0);
    hipLaunchKernelGGL(ln_fwd_kernel, SHERF_GRID(rows), 0, as_stream(stream), x, w, b, rows, y, xh, inv);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_ln_bwd(const float* dy, const float* w, const float* xh, const float* inv, int64_t rows, float* dx,
                                float* dw, float* db, sherf_stream_t stream) {
    SHERF_CHECK_ARG(dy && w && xh && inv && dx && dw && db && rows > 0);
    const int64_t groups = (rows + 31) / 32;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)(groups < 4096 ? groups : 4096)), dim3(256), 0, as_stream(stream), dy, w, xh, inv, rows, dx,
                       dw, db, static_cast<const float*>(nullptr));
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_ln_bwd_add(const float* dy, const float* w, const float* xh, const float* inv, int64_t rows, const float* addend, float* dx,
                                    float* dw, float* db, sherf_stream_t stream) {
    SHERF_CHECK_ARG(dy && w && xh && inv && addend && dx && dw && db && rows > 0 && addend != dx);
    const int64_t groups = (rows + 31) / 32;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)(groups < 4096 ? groups : 4096)), dim3(256), 0, as_stream(stream), dy, w, xh, inv, rows, dx,
                       dw, db, addend);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_attn_fwd(const float* qkv, int64_t n, float* att, float* o, sherf_stream_t stream) {
    SHERF_CHECK_ARG(qkv && att && o && n > 0 && ((reinterpret_cast<size_t>(qkv) | reinterpret_cast<size_t>(o)) & 15) == 0);
    hipLaunchKernelGGL(attn_fwd_kernel, SHERF_GRID(n * 3), 0, as_stream(stream), qkv, n, att, o);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_attn_bwd(const float* qkv, const float* att, const float* d_o, int64_t n, float* d_qkv, sherf_stream_t stream) {
    SHERF_CHECK_ARG(qkv && att && d_o && d_qkv && n > 0 && ((reinterpret_cast<size_t>(qkv) | reinterpret_cast<size_t>(d_o) | reinterpret_cast<size_t>(d_qkv)) & 15) == 0);
    hipLaunchKernelGGL(attn_bwd_kernel, SHERF_GRID(n * 3), 0, as_stream(stream), qkv, att, d_o, n, d_qkv);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_gelu_fwd(const float* u, int64_t count, float* ge, sherf_stream_t stream) {
    SHERF_CHECK_ARG(u && ge && count > 0);
    hipLaunchKernelGGL(gelu_fwd_kernel, SHERF_GRID(count), 0, as_stream(stream), u, count, ge);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_gelu_bwd(float* d, const float* u, int64_t count, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d && u && count > 0);
    hipLaunchKernelGGL(gelu_bwd_kernel, SHERF_GRID(count), 0, as_stream(stream), d, u, count);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_rgb_fwd(float* lin, int64_t count, sherf_stream_t stream) {
    SHERF_CHECK_ARG(lin && count > 0);
    hipLaunchKernelGGL(rgb_fwd_kernel, SHERF_GRID(count), 0, as_stream(stream), lin, count);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_rgb_bwd(float* d, const float* rgb, int64_t count, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d && rgb && count > 0);
    hipLaunchKernelGGL(rgb_bwd_kernel, SHERF_GRID(count), 0, as_stream(stream), d, rgb, count);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_unfold32(const float* d_f, const float* W, const float* in, int HW, int groups, int pix_stride,
                                  int64_t group_base, float* d_in, float* dW, sherf_stream_t stream) {
    SHERF_CHECK_ARG(d_f && W && in && d_in && dW && HW > 0 && groups > 0 && pix_stride >= 32 && pix_stride % 4 == 0);
    hipLaunchKernelGGL(unfold32_kernel, dim3((HW + 255) / 256, groups), dim3(256), 0, as_stream(stream), d_f, W, in, HW, pix_stride,
                       group_base, d_in, dW);
    SHERF_LAUNCH_CHECK();
}

extern "C" int sherf_bwd_bn_relu_apply(const float* raw, const float* bnparam, const int32_t* n_rows, int64_t cap, int C, float* act,
                                       sherf_stream_t stream) {
    SHERF_CHECK_ARG(raw && bnparam && n_rows && act && cap > 0 && C > 0);
    hipLaunchKernelGGL(bn_relu_apply_kernel, SHERF_GRID(cap * C), 0, as_stream(stream), raw, bnparam, n_rows, cap, C, act);
    SHERF_LAUNCH_CHECK();
}

extern "C" const char* sherf_bwd_last_error(void) { return g_sherf_err; }
