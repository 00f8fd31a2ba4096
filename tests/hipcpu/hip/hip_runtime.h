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

template <class K, class... A>
void hipcpu_launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t, A... args) {
    gridDim = grid; blockDim = block;
    const unsigned nt = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                std::barrier<> bar(nt);
                hipcpu_barrier = &bar;
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
