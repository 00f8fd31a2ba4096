// TEST INFRASTRUCTURE: a minimal "HIP on the CPU" shim.  The kernel sources of sherf_amd/csrc that use no gfx950 intrinsics
// (bwd_dense.hip, bwd_encoder.hip, composite.hip, gather.hip, fold.hip) are compiled UNCHANGED with g++ against this header
// and executed on the host: every GPU thread of a workgroup is a cooperative fiber on one OS thread (runtime.cpp), workgroups
// spread over the host cores, __syncthreads() / wave collectives are yields, atomics std::atomic_ref.  Slow (small inputs
// only) but it runs the real kernel code -- indexing, strides,
// reductions -- so the backward kernels, written without GPU time, are checked against their specification before they ever
// reach hardware (tests/test_hipcpu_kernels.py).  Nothing in the product uses this.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
#define __shared__ static thread_local          // one copy per worker OS thread = per workgroup in flight

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipcpu_idx { unsigned x, y, z; };
extern thread_local hipcpu_idx threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern thread_local unsigned char hipcpu_dyn[];          // dynamic shared memory of the current workgroup
void hipcpu_syncthreads();
void hipcpu_wave_sync();
void hipcpu_run_grid(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
inline void __syncthreads() { hipcpu_syncthreads(); }
// ---- wave64 collectives (for the kernels that use them: the 64 threads of a wave meet at a per-wave barrier and exchange
// through a per-wave scratch line).  Every lane of the wave must reach the call, as on the hardware. ----
extern thread_local unsigned char hipcpu_wave_scratch[16][2][64 * 64];     // double-buffered: ONE wave sync per collective
extern thread_local unsigned char hipcpu_wave_flip[1024];                   // per GPU thread: which buffer its next collective uses
unsigned hipcpu_linear_tid();
inline int hipcpu_lane() { return (int)(threadIdx.x & 63); }
inline int hipcpu_wave() { return (int)(threadIdx.x >> 6); }
// A collective = every lane publishes into the wave's current buffer, ONE wave sync, every lane reads.  The next collective
// uses the other buffer: a lane can be at most one collective ahead of the slowest lane of its wave (it has to pass the sync
// in between), so a buffer is never overwritten while someone still reads it.
inline unsigned char* hipcpu_collective_buffer() {
    unsigned char& f = hipcpu_wave_flip[hipcpu_linear_tid()];
    f ^= 1;
    return hipcpu_wave_scratch[hipcpu_wave()][f];
}
template <class T> inline T hipcpu_exchange(T v, int src_lane) {
    T* sc = reinterpret_cast<T*>(hipcpu_collective_buffer());
    sc[hipcpu_lane()] = v;
    hipcpu_wave_sync();
    return sc[src_lane];
}
inline float __shfl_xor(float v, int m) { return hipcpu_exchange(v, hipcpu_lane() ^ m); }
inline int __shfl_xor(int v, int m) { return hipcpu_exchange(v, hipcpu_lane() ^ m); }
inline unsigned __shfl_xor(unsigned v, int m) { return hipcpu_exchange(v, hipcpu_lane() ^ m); }
template <class T> inline T __shfl_down(T v, int d) { const int l = hipcpu_lane() + d; return hipcpu_exchange(v, l < 64 ? l : hipcpu_lane()); }
template <class T> inline T __shfl_up(T v, int d) { const int l = hipcpu_lane() - d; return hipcpu_exchange(v, l >= 0 ? l : hipcpu_lane()); }
template <class T> inline T __shfl(T v, int src) { return hipcpu_exchange(v, src & 63); }
inline unsigned long long __ballot(int pred) {           // every lane of the wave must call it (no divergence), as the shim's other collectives
    unsigned char* sc = hipcpu_collective_buffer();
    sc[hipcpu_lane()] = pred != 0;
    hipcpu_wave_sync();
    unsigned long long m = 0;
    const int live = (int)std::min<unsigned>(64u, blockDim.x * blockDim.y * blockDim.z - 64u * hipcpu_wave());
    for (int l = 0; l < live; ++l) m |= (unsigned long long)sc[l] << l;
    return m;
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return hipcpu_exchange(v, lane); }
#ifdef __clang__
// v_permlane32_swap_b32 vdst, src: lanes 32..63 of vdst <-> lanes 0..31 of src; -> {new vdst, new src}
typedef unsigned hipcpu_u32x2 __attribute__((ext_vector_type(2)));
inline hipcpu_u32x2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
    const int l = hipcpu_lane();
    const unsigned from_src = hipcpu_exchange(src, l >= 32 ? l - 32 : l);       // what the upper half of vdst receives
    const unsigned from_dst = hipcpu_exchange(vdst, l < 32 ? l + 32 : l);       // what the lower half of src receives
    hipcpu_u32x2 r;
    r[0] = l < 32 ? vdst : from_src;
    r[1] = l < 32 ? from_dst : src;
    return r;
}
#endif
inline void __builtin_amdgcn_wave_barrier() { hipcpu_wave_sync(); }       // lockstep ordering points the kernels mark explicitly
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) std::atomic_ref<std::remove_cv_t<std::remove_pointer_t<decltype(p)>>>(*const_cast<std::remove_cv_t<std::remove_pointer_t<decltype(p)>>*>(p)).load()
#define __hip_atomic_store(p, v, order, scope) std::atomic_ref<std::remove_cv_t<std::remove_pointer_t<decltype(p)>>>(*(p)).store(v)
inline int __ffsll(long long v) { return __builtin_ffsll(v); }

struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int3 { int x, y, z; };
struct int2 { int x, y; };
inline int2 make_int2(int a, int b) { return {a, b}; }
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
inline uint2 make_uint2(uint32_t a, uint32_t b) { return {a, b}; }
inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return {a, b, c, d}; }
inline int3 make_int3(int a, int b, int c) { return {a, b, c}; }

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
// streams / events: every launch is synchronous, so ordering calls are no-ops; an event remembers the host clock at its record
// (the library's event sets live for the process: never freed here)
typedef double* hipEvent_t;
constexpr unsigned hipEventDisableTiming = 2;
constexpr unsigned hipEventReleaseToDevice = 0x40000000;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new double(0.0); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new double(0.0); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    *e = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipLaunchHostFunc(hipStream_t, void (*fn)(void*), void* arg) { fn(arg); return hipSuccess; }
// hipGraph capture (csrc/frame.hip replays frames as graphs): every launch of the shim runs at once, so there is nothing to capture --
// hipStreamBeginCapture reports "not supported" and the frame driver takes its launch-by-launch path, which is what the host build tests
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
constexpr int hipStreamCaptureModeThreadLocal = 1;
constexpr unsigned hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 801; }
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 801; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 801; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 801; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 801; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*b - *a); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorName(hipError_t) { return "hipError"; }
inline const char* hipGetErrorString(hipError_t) { return "hipcpu"; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
constexpr int hipDeviceAttributeMultiprocessorCount = 63;
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 2; return hipSuccess; }      // two 'CUs': persistent kernels loop
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
constexpr int hipMemcpyDeviceToHost = 2;
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

template <class T> inline T atomicAdd(T* p, T v) { return std::atomic_ref<T>(*p).fetch_add(v); }
inline float unsafeAtomicAdd(float* p, float v) { return std::atomic_ref<float>(*p).fetch_add(v); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return std::atomic_ref<unsigned>(*p).fetch_or(v); }
inline int atomicOr(int* p, int v) { return std::atomic_ref<int>(*p).fetch_or(v); }
inline unsigned atomicAnd(unsigned* p, unsigned v) { return std::atomic_ref<unsigned>(*p).fetch_and(v); }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return std::atomic_ref<unsigned long long>(*p).fetch_or(v); }
template <class T> inline T atomicMin(T* p, T v) { std::atomic_ref<T> a(*p); T o = a.load(); while (v < o && !a.compare_exchange_weak(o, v)) {} return o; }
template <class T> inline T atomicMax(T* p, T v) { std::atomic_ref<T> a(*p); T o = a.load(); while (v > o && !a.compare_exchange_weak(o, v)) {} return o; }
inline long long __double2ll_rn(double x) { return llrint(x); }
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
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __dmul_rn(double a, double b) { return a * b; }
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
    Line* sc = reinterpret_cast<Line*>(hipcpu_collective_buffer());
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
    return d;
}
// v_mfma_f32_32x32x16_f16: the same shape with fp16 operands
typedef _Float16 hipcpu_f16x8 __attribute__((ext_vector_type(8)));
inline hipcpu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(hipcpu_f16x8 a, hipcpu_f16x8 b, hipcpu_f32x16 c, int, int, int) {
    struct Line { float a[8], b[8]; };
    Line* sc = reinterpret_cast<Line*>(hipcpu_collective_buffer());
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
    return d;
}
// v_cvt_pkrtz_f16_f32: two floats -> packed fp16, rounded TOWARD ZERO (finite overflow saturates to +-65504)
typedef _Float16 hipcpu_f16x2 __attribute__((ext_vector_type(2)));
inline _Float16 hipcpu_f32_to_f16_rtz(float x) {
    _Float16 h = (_Float16)x;                                  // round to nearest even
    const float back = (float)h;
    if (std::isinf(back) && !std::isinf(x)) { unsigned short u = x > 0 ? 0x7bff : 0xfbff; memcpy(&h, &u, 2); return h; }
    if (fabsf(back) > fabsf(x)) {                              // rounded away from zero: step one ulp toward zero
        unsigned short u; memcpy(&u, &h, 2); u -= 1; memcpy(&h, &u, 2);
    }
    return h;
}
inline hipcpu_f16x2 __builtin_amdgcn_cvt_pkrtz(float a, float b) { hipcpu_f16x2 r; r[0] = hipcpu_f32_to_f16_rtz(a); r[1] = hipcpu_f32_to_f16_rtz(b); return r; }
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
// v_fract_f32 / v_sin_f32 / v_cos_f32 (the latter two take REVOLUTIONS)
inline float __builtin_amdgcn_fractf(float x) { return x - floorf(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_sinf(float r) { return (float)sin(6.283185307179586476925 * (double)r); }
inline float __builtin_amdgcn_cosf(float r) { return (float)cos(6.283185307179586476925 * (double)r); }
inline float __expf(float x) { return expf(x); }
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __sinf(float x) { return sinf(x); }
inline float __cosf(float x) { return cosf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }

template <class K, class... A>
void hipcpu_launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t, A... args) {
    const std::function<void()> body = [=]() { kernel(args...); };
    hipcpu_run_grid(grid, block, smem, body);
}
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) hipcpu_launch(kernel, grid, block, smem, stream, __VA_ARGS__)
