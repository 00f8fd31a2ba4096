#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k1(const f16x8* a, const f16x8* b, float* o) {   // mfma -> valu read
    f16x8 A = a[threadIdx.x], B = b[threadIdx.x];
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    o[threadIdx.x] = c[0] + c[5];
}
__global__ void k2(const f16x8* a, const f16x8* b, float* o) {   // mfma -> mfma srcA (overlap)
    f16x8 A = a[threadIdx.x], B = b[threadIdx.x];
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    f16x8 A2;
    for (int i = 0; i < 8; ++i) A2[i] = (_Float16)c[i];
    f32x16 d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, B, d, 0, 0, 0);
    o[threadIdx.x] = d[0];
}
__global__ void k3(const float* a, const f16x8* b, float* o) {   // valu write -> mfma read
    f16x8 B = b[threadIdx.x];
    float x = a[threadIdx.x];
    f16x8 A2;
    for (int i = 0; i < 8; ++i) A2[i] = (_Float16)(x * (float)i);
    f32x16 d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, B, d, 0, 0, 0);
    o[threadIdx.x] = d[0];
}
__global__ void k4(const f16x8* a, const f16x8* b, float* o) {   // mfma -> mfma different acc reading prev dst as srcC? and direct reuse as SrcB
    f16x8 A = a[threadIdx.x], B = b[threadIdx.x];
    f32x16 c = {0}, d = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);   // srcC = c, dst = d (overlapped different vdst)
    o[threadIdx.x] = d[0] ;
}
__global__ void k5(const f16x8* a, const f16x8* b, float* o, float* o2) {   // mfma -> store of acc (VMEM read)
    f16x8 A = a[threadIdx.x], B = b[threadIdx.x];
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    *(f32x16*)(o + threadIdx.x * 16) = c;
}
__global__ void k6(const f16x8* a, const f16x8* b, float* o, uint32_t sel) {   // permlane32 swap
    float v = o[threadIdx.x];
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    o[threadIdx.x] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
