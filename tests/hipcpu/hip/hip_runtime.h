// TEST INFRASTRUCTURE: a minimal "HIP on the CPU" shim.  The kernel sources of sherf_amd/csrc that use no gfx950 intrinsics
// (bwd_dense.hip, bwd_encoder.hip, composite.hip, gather.hip, fold.hip) are compiled UNCHANGED with g++ against this header
// and executed on the host: one std::thread per GPU thread of a workgroup, workgroups one after the other, __syncthreads() a
// std::barrier, atomics std::atomic_ref.  Slow (tiny inputs only) but it runs the real kernel code -- indexing, strides,
// reductions -- so the backward kernels, written without GPU time, are checked against their specification before they ever
// reach hardware (tests/test_hipcpu_kernels.py).  Nothing in the product uses this.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipcpu_idx { unsigned x, y, z; };
extern thread_local hipcpu_idx threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern std::barrier<>* hipcpu_barrier;
extern unsigned char hipcpu_dyn[];                       // dynamic shared memory of the current workgroup
inline void __syncthreads() { hipcpu_barrier->arrive_and_wait(); }
// ---- wave64 collectives (for the kernels that use them: the 64 threads of a wave meet at a per-wave barrier and exchange
// through a per-wave scratch line).  Every lane of the wave must reach the call, as on the hardware. ----
extern std::barrier<>* hipcpu_wave_barrier[16];
extern unsigned char hipcpu_wave_scratch[16][64 * 64];
inline int hipcpu_lane() { return (int)(threadIdx.x & 63); }
inline int hipcpu_wave() { return (int)(threadIdx.x >> 6); }
inline void hipcpu_wave_sync() { hipcpu_wave_barrier[hipcpu_wave()]->arrive_and_wait(); }
template <class T> inline T hipcpu_exchange(T v, int src_lane) {
    T* sc = reinterpret_cast<T*>(hipcpu_wave_scratch[hipcpu_wave()]);
    sc[hipcpu_lane()] = v;
    hipcpu_wave_sync();
    const T r = sc[src_lane];
    hipcpu_wave_sync();
    return r;
}
inline float __shfl_xor(float v, int m) { return hipcpu_exchange(v, hipcpu_lane() ^ m); }
inline int __shfl_xor(int v, int m) { return hipcpu_exchange(v, hipcpu_lane() ^ m); }

struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int3 { int x, y, z; };
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
inline uint2 make_uint2(uint32_t a, uint32_t b) { return {a, b}; }
inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return {a, b, c, d}; }
inline int3 make_int3(int a, int b, int c) { return {a, b, c}; }

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipcpu"; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

template <class T> inline T atomicAdd(T* p, T v) { return std::atomic_ref<T>(*p).fetch_add(v); }
inline float unsafeAtomicAdd(float* p, float v) { return std::atomic_ref<float>(*p).fetch_add(v); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return std::atomic_ref<unsigned>(*p).fetch_or(v); }
inline int __popc(uint32_t v) { return __builtin_popcount(v); }
inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline int64_t min(int64_t a, int64_t b) { return a < b ? a : b; }
inline int64_t max(int64_t a, int64_t b) { return a > b ? a : b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

#ifdef __clang__          // kernels with bf16 MFMA are compiled with clang (ext_vector_type, __bf16)
typedef __bf16 hipcpu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hipcpu_f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x16_bf16: D[32x32] = A[32x16] B[16x32] + C.  Lane l = (i = l & 31, h = l >> 5): A operand holds A[i][8h..8h+8),
// B operand holds B[8h..8h+8)[i], C/D register r holds element [row (r & 3) + 8 (r >> 2) + 4h][column i].
inline hipcpu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipcpu_bf16x8 a, hipcpu_bf16x8 b, hipcpu_f32x16 c, int, int, int) {
    struct Line { float a[8], b[8]; };
    Line* sc = reinterpret_cast<Line*>(hipcpu_wave_scratch[hipcpu_wave()]);
    const int l = hipcpu_lane(), col = l & 31, h = l >> 5;
    for (int e = 0; e < 8; ++e) { sc[l].a[e] = (float)a[e]; sc[l].b[e] = (float)b[e]; }
    hipcpu_wave_sync();
    hipcpu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = 0.f;
        for (int hh = 0; hh < 2; ++hh)
            for (int e = 0; e < 8; ++e) acc += sc[row + 32 * hh].a[e] * sc[col + 32 * hh].b[e];
        d[r] += acc;
    }
    hipcpu_wave_sync();
    return d;
}
#endif
inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel) {          // v_perm_b32: byte select from {hi:lo}
    const unsigned long long src = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((src >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
template <class T> inline T __builtin_amdgcn_readfirstlane(T v) { return v; }                 // callers pass wave-uniform values
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline float __expf(float x) { return expf(x); }
inline float __sinf(float x) { return sinf(x); }
inline float __cosf(float x) { return cosf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }

template <class K, class... A>
void hipcpu_launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t, A... args) {
    gridDim = grid; blockDim = block;
    const unsigned nt = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                std::barrier<> bar(nt);
                hipcpu_barrier = &bar;
                std::vector<std::unique_ptr<std::barrier<>>> wbars;
                if (nt % 64 == 0 && nt / 64 <= 16)
                    for (unsigned w = 0; w < nt / 64; ++w) { wbars.emplace_back(new std::barrier<>(64)); hipcpu_wave_barrier[w] = wbars.back().get(); }
                if (smem) memset(hipcpu_dyn, 0, smem);
                std::vector<std::thread> th;
                th.reserve(nt);
                for (unsigned t = 0; t < nt; ++t)
                    th.emplace_back([=, &bar]() {
                        threadIdx = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                        blockIdx = {bx, by, bz};
                        kernel(args...);
                        bar.arrive_and_drop();              // a thread that returned no longer takes part in later barriers
                    });
                for (auto& x : th) x.join();
            }
}
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) hipcpu_launch(kernel, grid, block, smem, stream, __VA_ARGS__)
