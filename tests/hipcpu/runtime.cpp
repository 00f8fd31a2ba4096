// definitions of the shim's globals (one per CPU-built library)
#include <hip/hip_runtime.h>
thread_local hipcpu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;
std::barrier<>* hipcpu_barrier = nullptr;
alignas(16) unsigned char hipcpu_dyn[160 * 1024];
std::barrier<>* hipcpu_wave_barrier[16] = {};
alignas(64) unsigned char hipcpu_wave_scratch[16][64 * 64];
