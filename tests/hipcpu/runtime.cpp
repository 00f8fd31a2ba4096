// definitions of the shim's globals (one per CPU-built library)
#include <hip/hip_runtime.h>
thread_local hipcpu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;
std::barrier<>* hipcpu_barrier = nullptr;
alignas(16) unsigned char hipcpu_dyn[160 * 1024];
