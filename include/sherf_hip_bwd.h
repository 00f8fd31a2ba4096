/* sherf_hip_bwd.h -- C ABI of libsherf_hip_bwd.so: building blocks of the BACKWARD of SHERF's rendering hot path
 * (BASELINE config 5: forward render + backward through HIP kernels).
 *
 * Every entry point mirrors a function of oracle/backward_explicit.py that is verified on the CPU against autograd and against the
 * unmodified reference's gradients; the kernels are checked from their source on the CPU (tests/hipcpu) and on the MI355X
 * (tests/test_gpu_backward.py).  Kept in its own library so that the forward library libsherf_hip.so carries no training code; no vendor
 * math library is linked (the GEMMs are the MFMA kernels of csrc/bwd_gemm.hip).
 *
 * Conventions: as sherf_hip.h (device pointers, caller-owned, launch on `stream`, 0 / negative error code).  Matrices are
 * row-major fp32 with an explicit leading dimension (`ld*`, in elements).  fp32 storage everywhere; the products run on MFMA with
 * the operands split into three bf16 parts (fp32 range, 24 bits).
 */
#ifndef SHERF_HIP_BWD_H
#define SHERF_HIP_BWD_H
#include <stdint.h>

#include "sherf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* sherf_bwd_last_error(void);

/* C[M][N] = op(A)[M][K] . op(B)[K][N] + beta * C   (row-major; op = transpose when trans* != 0) on `stream`: hand-written MFMA
 * kernels (csrc/bwd_gemm.hip: three-part bf16 operand split, six products, fp32 accumulate -- fp32 grade) for the three patterns of the
 * backward, each with one huge dimension (the valid samples) and two layer-width ones: transA = 0 (M huge: forward recompute and data
 * gradients), transA = 1 / transB = 0 (K huge: weight gradients, partial sums added to C with fp32 atomics -- the summation order is
 * not deterministic); anything else runs on a plain fp32 kernel.  Replaces every `x @ W.t()` / `d.t() @ x` of
 * oracle/backward_explicit.py (decoder_bwd.lin_bwd, transformer_bwd).
 * READABLE ROW PADDING (round 5's streaming kernel): with transA = 0, beta = 0, lda >= 16 * ceil(K / 16), lda % 4 == 0 and A
 * 16-byte aligned, a row of A is read in whole 16-float blocks -- up to 15 floats beyond column K - 1 (masked in registers, never used).  Those
 * floats must be READABLE: the natural case is a row padded to a multiple of 16 floats (lda >= 16 * ceil(K / 16) counted from the row's own
 * first element).  An A that is a column slice of a wider matrix at a non-zero offset (offset % lda + 16 * ceil(K / 16) > lda) reads into the
 * next row -- harmless -- except on the LAST row, where the caller must own 15 more floats behind the buffer, or pass such a slice with an lda
 * that fails one of the conditions above (any lda % 4 != 0 view of it) to take the general kernel.  sherf_amd.backward_dense.Mat.empty_ld pads. */
int sherf_bwd_gemm(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                   float* C, int ldc, float beta, sherf_stream_t stream);
/* The same product with the layer's epilogue fused: C = act(op(A) op(B) + beta C + bias[column]), act 0 identity / 1 ReLU (bias may be
 * NULL).  Fused into the tall-product kernel's store (the forward recompute of the Linear layers: one pass over [n, C] less per
 * layer); a second small pass on the other paths. */
int sherf_bwd_gemm_bias_act(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                            float* C, int ldc, float beta, const float* bias, int act, sherf_stream_t stream);
/* which kernel the last sherf_bwd_gemm call of this process took: 1 = tall MFMA (column-sliced when B exceeds the LDS), 2 = weight-
 * gradient MFMA, 0 = the plain fp32 kernel (tests assert that no layer shape of the path reaches it). */
int sherf_bwd_gemm_last_path(void);
/* The data gradient of a Linear whose input came out of a ReLU (decoder_bwd of oracle/backward_explicit.py: d_h = (d_out . W) * [h > 0], db = sum_rows d_h):
 *   C[M,N] = A[M,K] . B[K,N]  (+ r1_s[row * r1_lds] * r1_w[column]: a second head's K = 1 product, e.g. alpha_linear beside feature_linear)
 *   C = 0 where mask[row * ldm + column] <= 0;   colsum[column] += sum_rows C.
 * r1_s / r1_w (both or neither), mask, colsum: optional.  One kernel for N, K in (96, 128], 16-byte aligned rows of A; otherwise the separate kernels. */
/* C[M,N] = act(A[M,K] . op(B) + bias) + addend[M,N]  (op(B) = B^T [N,K] when transB): a Linear joining a residual stream (renderer.py:979-1005: the
 * transformer's to_out and feed-forward), the residual added in the product's store.  addend != C. */
int sherf_bwd_gemm_bias_act_add(int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                const float* bias, int act, const float* addend, int ld_add, sherf_stream_t stream);
int sherf_bwd_gemm_dgrad_fused(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                               const float* r1_s, int r1_lds, const float* r1_w, const float* mask, int ldm, float* colsum, sherf_stream_t stream);

/* tokens / extras of the forward (tile-major, sherf_gather_tokens) -> row-major tok[n][96], ext[n][12]; and the inverse
 * for d_tokens (rows beyond n are zero filled up to the tile boundary). */
int sherf_bwd_untile(const float* tokens_tiled, const float* extras_tiled, int64_t n, float* tok, float* ext,
                     sherf_stream_t stream);
int sherf_bwd_tile_tokens(const float* d_tok, int64_t n, float* d_tokens_tiled, sherf_stream_t stream);

/* y[r][c] = act(y[r][c] + bias[c]) in place, act 0 = none, 1 = ReLU (bias may be NULL). */
int sherf_bwd_bias_act(float* y, int ldy, const float* bias, int64_t n, int C, int act, sherf_stream_t stream);
/* d[r][c] = h[r][c] > 0 ? d[r][c] : 0  (ReLU backward through the stored post-activation). */
int sherf_bwd_relu_mask(float* d, int ldd, const float* h, int ldh, int64_t n, int C, sherf_stream_t stream);
/* d *= [h > 0] and out[c] += sum_r d[r][c] in one pass (ReLU backward + the bias gradient of the layer that produced h); C divides 256;
 * out accumulates (zeroed by the caller). */
int sherf_bwd_relu_mask_colsum(float* d, int ldd, const float* h, int ldh, int64_t n, int C, float* out, sherf_stream_t stream);
/* out[c] += sum_r d[r][c]   (bias gradients; out must be zeroed by the caller). */
int sherf_bwd_colsum(const float* d, int ldd, int64_t n, int C, float* out, sherf_stream_t stream);
/* dst[r][0..C) (+)= src[r][0..C): strided column-block copy (add != 0: accumulate). */
int sherf_bwd_copy2d(float* dst, int ldd, const float* src, int lds, int64_t n, int C, int add, sherf_stream_t stream);

/* NeRF positional encoding (renderer.py:875-916): out[r] = [x, sin(f0 x), cos(f0 x), sin(f1 x), ...], x = in[r][0..3),
 * 3 + 6*NF columns. */
int sherf_bwd_pe(const float* in, int ldi, int64_t n, int NF, float* out, int ldo, sherf_stream_t stream);

/* LayerNorm over 32 features (eps 1e-5), rows = n * tokens.  fwd keeps xh[rows][32], inv[rows]; bwd returns dx and
 * accumulates dw[32], db[32] (zeroed by the caller).  (transformer_bwd._ln_fwd / _ln_bwd) */
int sherf_bwd_ln_fwd(const float* x, const float* w, const float* b, int64_t rows, float* y, float* xh, float* inv,
                     sherf_stream_t stream);
int sherf_bwd_ln_bwd(const float* dy, const float* w, const float* xh, const float* inv, int64_t rows, float* dx,
                     float* dw, float* db, sherf_stream_t stream);
/* the same with a residual gradient joining: dx = (LayerNorm backward of dy) + addend[rows][32]  (addend != dx) */
int sherf_bwd_ln_bwd_add(const float* dy, const float* w, const float* xh, const float* inv, int64_t rows, const float* addend, float* dx,
                         float* dw, float* db, sherf_stream_t stream);

/* 3-token, 3-head x 16 attention core on qkv[n][3 tok][144] (q | k | v, each 3 heads x 16; renderer.py:949-977):
 * fwd: att[n][3 head][3][3] = softmax(q k^T / 4), o[n][3 tok][48]; bwd: d_o -> d_qkv.  qkv, o, d_o, d_qkv: 16-byte aligned
 * (a head's 16 floats travel as four dwordx4 accesses; SHERF_EINVAL otherwise). */
int sherf_bwd_attn_fwd(const float* qkv, int64_t n, float* att, float* o, sherf_stream_t stream);
int sherf_bwd_attn_bwd(const float* qkv, const float* att, const float* d_o, int64_t n, float* d_qkv, sherf_stream_t stream);

/* exact GELU: fwd ge = u Phi(u); bwd d_u = d_ge (Phi(u) + u phi(u)), in place on d. */
int sherf_bwd_gelu_fwd(const float* u, int64_t count, float* ge, sherf_stream_t stream);
int sherf_bwd_gelu_bwd(float* d, const float* u, int64_t count, sherf_stream_t stream);

/* rgb head (triplane.py:314): fwd rgb = sigmoid(lin) * 1.002 - 0.001 in place; bwd d_lin = d_rgb * 1.002 s (1 - s)
 * from the stored rgb. */
int sherf_bwd_rgb_fwd(float* lin, int64_t count, sherf_stream_t stream);
int sherf_bwd_rgb_bwd(float* d, const float* rgb, int64_t count, sherf_stream_t stream);

/* Transpose of sherf_fold_tables (csrc/fold.hip): the gradient d_f of a folded, channel-last table
 * (element [g][pix][o] at g*group_base + pix*pix_stride + o) goes back to the NCHW source: d_in[g*32+c][pix] =
 * sum_o W[o][c] d_f[g][pix][o], and dW[32 o][32 c] += sum d_f (x) in  (dW zeroed by the caller).
 * (oracle/backward_explicit.py: folded_taps_bwd step (ii), tri-planes and 2-D feature map) */
int sherf_bwd_unfold32(const float* d_f, const float* W, const float* in, int HW, int groups, int pix_stride,
                       int64_t group_base, float* d_in, float* dW, sherf_stream_t stream);

/* act[r][c] = relu(raw[r][c] * scale[c] + shift[c]) for r < *n_rows, zero up to `cap` rows: the activations of a voxel level
 * as the forward's consumers see them (BatchNorm+ReLU applied on the fly from bnparam[3][C]). */
int sherf_bwd_bn_relu_apply(const float* raw, const float* bnparam, const int32_t* n_rows, int64_t cap, int C, float* act,
                            sherf_stream_t stream);

/* ---- sparse voxel encoder backward (oracle/backward_explicit.py: _bn_relu_bwd, encoder_bwd) ----------------------------
 * BatchNorm (batch statistics over the reference's row set) + ReLU backward of one layer: d_out = gradient w.r.t. the
 * per-voxel activation relu(bn(raw)) + (mult - 1) relu(shift); raw [cap][C] the conv output, bnparam [3][C] / stats [2][C]
 * as left by the forward, mult (level 0 only, else NULL), n_total = rows of the reference row set, n_rows = voxels.
 * -> d_raw [cap][C] (zero beyond n_rows), dgamma[C], dbeta[C]; sums[3][C] is scratch.  amax (may be NULL): receives the bit pattern
 * of max |d_raw| -- the input scale of the MFMA input-gradient convolution (include/sherf_hip.h: sherf_svox_conv3_dgrad). */
int sherf_bwd_bn_relu(const float* d_out, const float* raw, const float* bnparam, const float* stats, const float* gamma,
                      const int32_t* mult, const int32_t* n_total, const int32_t* n_rows, int64_t cap, int C, float* sums,
                      float* d_raw, float* dgamma, float* dbeta, uint32_t* amax, sherf_stream_t stream);
/* (fp32 VALU form, kept as the check of the MFMA one the step uses -- sherf_svox_conv3_dgrad, 16.5 ms -> see profiles/ per step.)
 * Sparse conv backward w.r.t. its input: for the rows (keys_i, dims Di..) of the INPUT level,
 * d_in[i][ci] = sum_k sum_co d_raw[o(i,k)][co] W[co][k][ci], W = the reference weight [Cout][27][Cin]; the rows o live in the
 * level described by wp_o (dims Do..): mode 0 submanifold (same level), mode 1 stride-2 (o in the coarser level). */
int sherf_bwd_conv_dgrad(const int32_t* keys_i, const int32_t* n_rows_i, int Di, int Hi, int Wi, const uint32_t* wp_o, int Do,
                         int Ho, int Wo, const float* d_raw, int Cout, const float* W, int Cin, int mode, int max_rows,
                         float* d_in, sherf_stream_t stream);
/* Sparse conv backward w.r.t. its weight: dW[Cout][27][Cin] += sum_o d_raw[o] (x) act(in[nb(o,k)]) with the forward's
 * neighbour rule (mode 0 submanifold, 1 stride-2) and the producer's BatchNorm+ReLU applied on the fly (in_bn / in_mult as
 * in sherf_svox_conv3; NULL = raw input).  dW zeroed by the caller. */
int sherf_bwd_conv_wgrad(const int32_t* keys_o, const int32_t* n_rows_o, int Do, int Ho, int Wo, const uint32_t* wp_i, int Di,
                         int Hi, int Wi, const float* in_raw, int Cin, const float* in_bn, const int32_t* in_mult,
                         const float* d_raw, int Cout, int mode, int max_rows, float* dW, sherf_stream_t stream);
/* Level-0 aggregation backward: d_feat[i][:] = d_g[row of voxel coord[i]][:] (rows that shared a voxel share its gradient). */
int sherf_bwd_gather_rows(const int32_t* coord, int n, int D, int H, int W, const uint32_t* wp, const float* d_g, int C,
                          float* d_feat, sherf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
