/* sherf_hip_ops.h -- C ABI of libsherf_hip_ops.so: the reference's own two custom element-wise / FIR operators, used by the
 * per-frame PRODUCERS of the hot path's inputs (StyleGAN2 tri-plane backbone), as HIP kernels.
 *
 * Both kernels are verified against goldens produced by the unmodified reference's `_bias_act_ref` / `_upfirdn2d_ref`: from their unchanged
 * source on the CPU (tests/hipcpu) and on the MI355X (tests/test_gpu_ops.py).  Kept in their own library so that libsherf_hip.so (the
 * rendering hot path) is untouched.
 *
 * Conventions as sherf_hip.h: device pointers, caller-owned dense buffers, launch on `stream`, 0 or a negative error code,
 * sherf_ops_last_error() for the text.  dtype: 0 = float32, 1 = float16 (arithmetic always in float32, as bias_act.cu:22-24).
 */
#ifndef SHERF_HIP_OPS_H
#define SHERF_HIP_OPS_H
#include <stdint.h>

#include "sherf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* sherf_ops_last_error(void);

/* Replaces `_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)` (torch_utils/ops/bias_act.cpp:36-99,
 * kernel bias_act.cu:27-151).  n elements; element i uses bias b[(i / step_b) % size_b] (b may be NULL).
 *   grad 0:  y = clamp(gain * act(x + b))
 *   grad 1:  y = dL/dx given x := dy, with the forward's output `yref` (and input `xref` for swish)
 *   grad 2:  second-order term, with `dy` the first backward's incoming gradient
 * act: 1 linear, 2 relu, 3 lrelu(alpha), 4 tanh, 5 sigmoid, 6 elu, 7 selu, 8 softplus, 9 swish (bias_act.py:23-33).
 * clamp < 0 disables clamping.  xref / yref / dy may be NULL where the formula does not use them. */
int sherf_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n,
                   int64_t step_b, int64_t size_b, int grad, int act, float alpha, float gain, float clamp, int dtype,
                   sherf_stream_t stream);

/* Replaces `_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)` (upfirdn2d.cpp:22-104,
 * kernels upfirdn2d.cu:33-198): zero-upsample by (upx, upy), pad (negative = crop), correlate with the FIR filter f[fh][fw]
 * (float32; flipped unless flip_filter, i.e. a true convolution by default), keep every (downx, downy)-th sample, scale by gain.
 * x [N][C][H][W] -> y [N][C][OH][OW], OH = (H*upy + pady0 + pady1 - fh + downy) / downy (same for W); both dense NCHW. */
int sherf_upfirdn2d(const void* x, const float* f, void* y, int N, int C, int H, int W, int fh, int fw, int upx, int upy,
                    int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip_filter, float gain, int dtype,
                    sherf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
