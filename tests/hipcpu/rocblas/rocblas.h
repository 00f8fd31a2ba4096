// TEST INFRASTRUCTURE: stand-in for rocBLAS in the "HIP on the CPU" build (tests/hipcpu): column-major sgemm by plain loops.
#pragma once
typedef void* rocblas_handle;
typedef int rocblas_status;
constexpr rocblas_status rocblas_status_success = 0;
enum rocblas_operation { rocblas_operation_none = 111, rocblas_operation_transpose = 112 };
inline rocblas_status rocblas_create_handle(rocblas_handle* h) { *h = (void*)1; return 0; }
inline rocblas_status rocblas_set_stream(rocblas_handle, void*) { return 0; }
inline rocblas_status rocblas_sgemm(rocblas_handle, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const float* alpha,
                                    const float* A, int lda, const float* B, int ldb, const float* beta, float* C, int ldc) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int p = 0; p < k; ++p) {
                const float a = ta == rocblas_operation_none ? A[i + (long)p * lda] : A[p + (long)i * lda];
                const float b = tb == rocblas_operation_none ? B[p + (long)j * ldb] : B[j + (long)p * ldb];
                s += (double)a * b;
            }
            C[i + (long)j * ldc] = *alpha * (float)s + (*beta != 0.f ? *beta * C[i + (long)j * ldc] : 0.f);
        }
    return 0;
}
