// Shared helpers of libsherf_hip_ops.so (own error buffer: the library is independent of libsherf_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sherf_hip_ops.h"

extern char g_sherf_ops_err[256];

#define SHERF_CHECK_ARG(cond)                                                                             \
    do {                                                                                                  \
        if (!(cond)) {                                                                                    \
            snprintf(g_sherf_ops_err, sizeof(g_sherf_ops_err), "%s: bad argument: %s", __func__, #cond);  \
            return SHERF_EINVAL;                                                                          \
        }                                                                                                 \
    } while (0)

#define SHERF_LAUNCH_CHECK()                                                                              \
    do {                                                                                                  \
        hipError_t e_ = hipGetLastError();                                                                \
        if (e_ != hipSuccess) {                                                                           \
            snprintf(g_sherf_ops_err, sizeof(g_sherf_ops_err), "%s: launch failed: %s", __func__,         \
                     hipGetErrorString(e_));                                                              \
            return SHERF_ELAUNCH;                                                                         \
        }                                                                                                 \
        return SHERF_OK;                                                                                  \
    } while (0)

static inline hipStream_t as_stream(sherf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
