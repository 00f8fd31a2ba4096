/* sherf_hip.h -- C ABI of libsherf_hip.so: the MI355X (gfx950) implementation of SHERF's volumetric
 * rendering hot path.
 *
 * The reference has no FFI around this path (it is PyTorch eager code + two CUDA libraries); the
 * drop-in boundary is its Python class API (ImportanceRenderer / MipRayMarcher2 / RaySampler /
 * TriPlaneGenerator).  `sherf_amd/` mirrors those classes and calls the entry points below through
 * ctypes, the way the reference's own native ops are reached through
 * sherf/torch_utils/custom_ops.py:61 (get_plugin) + sherf/torch_utils/ops/bias_act.py:40-88.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in `_host`;
 *   - the callee never allocates persistent memory, never synchronises (two documented exceptions: sherf_render_frame with
 *     SHERF_FRAME_EXACT_GRIDS waits once for the frame's sample count; sherf_profile_frames_read drains the ring), launches on `stream`;
 *   - returns 0 on success, a negative SHERF_E* code on bad arguments / launch failure
 *     (the Python wrapper raises RuntimeError, like TORCH_CHECK in bias_act.cpp:39-55);
 *   - fp32 unless stated; B == 1 (the reference forces a per-GPU batch of one, renderer.py:320-321).
 * Reference citations are relative to /root/reference/sherf/.
 */
#ifndef SHERF_HIP_H
#define SHERF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sherf_stream_t; /* hipStream_t */

#define SHERF_OK 0
#define SHERF_EINVAL (-1)
#define SHERF_ELAUNCH (-2)

#define SHERF_V 6890     /* SMPL vertices (renderer.py:584 hard-codes 6890*3) */
#define SHERF_NJ 24
#define SHERF_MAX_CELLS (64 * 64 * 64)

int sherf_version(void);
const char* sherf_last_error(void);
/* profiling aid: ablation switches (bit0 sampler skips NN, bit1 sampler skips quick reject, bit2/3/4 gather skips
 * voxel / tri-plane / pixel taps, bit5 MLP without weight traffic, bit6 MLP without barriers). Results are WRONG with any
 * of bits 0-6 set.  bit7 only turns off the issue priority of the encoder's waves (results unchanged).  Default 0. */
int sherf_set_debug(int flags);

/* ---------------------------------------------------------------------------------------------
 * a7: SMPL bone transforms.  Replaces get_transform_params_torch + batch_rodrigues_torch +
 * get_rigid_transformation_torch (training/volumetric_rendering/renderer.py:129-157, 76-94, 96-126).
 *   poses[n_sets][72], shapes[n_sets][10]; J_template[24][3] = J_regressor @ v_template,
 *   J_shapedirs[24][3][10] = J_regressor @ shapedirs (constants of the SMPL asset);
 *   parents[24] int32.  Out: A[n_sets][24][12] (rows of the 3x4 rigid transform, rest pose removed)
 *   and posefeat[n_sets][207] = vec(R_1..23 - I) (renderer.py:582-583).
 */
int sherf_smpl_bones(const float* poses, const float* shapes, int n_sets, const float* J_template,
                     const float* J_shapedirs, const int32_t* parents, float* A, float* posefeat,
                     sherf_stream_t stream);

/* Per-vertex pose/shape blend offsets (renderer.py:578-593, 646-670):
 *   PO[n_sets][V][3] = posedirs[V*3][207] @ posefeat[set],  SO[n_sets][V][3] = shapedirs[V][3][10] @ shapes[set]. */
int sherf_smpl_offsets(const float* posedirs, const float* shapedirs, const float* posefeat,
                       const float* shapes, int n_sets, float* PO, float* SO, sherf_stream_t stream);

/* a8 collapsed: coarse_deform_target2c (renderer.py:558-621) depends on the query point only through its
 * nearest posed vertex j, so x_c = P[j] x_s + q[j], v_c = P[j] v_s.  T2C[V][12] = (P row-major 9, q 3).
 *   A_tgt/A_big [24][12]; PO_tgt, SO_tgt, PO_big [V][3]; weights [V][24]. */
int sherf_smpl_t2c_table(const float* weights, const float* A_tgt, const float* A_big, const float* PO_tgt,
                         const float* SO_tgt, const float* PO_big, float* T2C, sherf_stream_t stream);

/* a9+a10 collapsed: coarse_deform_c2source (renderer.py:623-684) + projection (:686-704) keyed by the
 * nearest T-pose vertex k: h = L[k] x_c + l[k] with uv = h.xy / (h.z + 1e-5).  C2S[V][12] = (L 9, l 3).
 *   R_obs[9], Th_obs[3] = obs_params R/Th; cam_R[9], cam_T[3], cam_K[9] = obs_R_all/obs_T_all/obs_K_all. */
int sherf_smpl_c2s_table(const float* weights, const float* A_big, const float* A_obs, const float* PO_big,
                         const float* SO_obs, const float* PO_obs, const float* R_obs, const float* Th_obs,
                         const float* cam_R, const float* cam_T, const float* cam_K, float* C2S,
                         sherf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a6 support: uniform cell list over n (<= 6890) vertices, replacing the brute-force pytorch3d K-NN of
 * renderer.py:315,564,627.  verts_s = (verts - Th) @ R when R/Th are non-null (renderer.py:313-314).
 *   grid_hdr[12] x 32 bit: origin xyz, cell, 1/cell (f32), nx, ny, nz, sub (int32 bit patterns), 3 unused;
 *   cell >= cell_size, enlarged only if an axis would exceed 64 cells; scratch: int32[5*n].
 *   cell_start[SHERF_MAX_CELLS+1] int32, cell_pts[n] float4 (x,y,z,bitcast(id)) sorted by cell. */
int sherf_build_cells(const float* verts, int n, const float* R, const float* Th, float cell_size,
                      float* grid_hdr, int32_t* cell_start, float* cell_pts, int32_t* scratch,
                      uint32_t* near_mask, sherf_stream_t stream);
/* Both per-frame lists in one launch: set 0 = verts_a in the SMPL frame (with near_mask; nullable, see sherf_build_near_lists), set 1 = verts_b untransformed.
 * grid_hdr[2][12], cell_start[2][SHERF_MAX_CELLS+1], cell_pts[2][n][4], scratch[2][5n]. */
int sherf_build_cells2(const float* verts_a, const float* R_a, const float* Th_a, const float* verts_b, int n,
                       float cell_size, float* grid_hdr, int32_t* cell_start, float* cell_pts, int32_t* scratch,
                       uint32_t* near_mask, sherf_stream_t stream);
/* Near lists of a cell list built by sherf_build_cells / sherf_build_cells2 (set 0): for every sub-cell of the near mask the exact
 * set of vertices that can lie within `radius` of a point of the sub-cell (the near mask's criterion: box distance < radius + margin),
 * as u16 indices into cell_pts.  near_hdr: int32[2 * SHERF_NEAR_SUBCELLS + 2] = (start, count) per sub-cell + the allocation cursor
 * (zeroed here); near_list: u16[list_cap], list_cap >= 125 n + 3 * SHERF_NEAR_SUBCELLS (every list padded to four entries); the order
 * of a list's entries is unspecified.  Replaces the per-candidate cell walk of pytorch3d's role (renderer.py:313-318) in
 * sherf_sample_mask_nn: 40 instead of 75 distance tests per candidate on a body, no segment bookkeeping. */
#define SHERF_NEAR_SUBCELLS 524288
int sherf_build_near_lists(const float* grid_hdr, const float* cell_pts, int n, float radius, int32_t* near_hdr,
                           uint16_t* near_list, int64_t list_cap, uint32_t* near_mask, sherf_stream_t stream);
/* near_mask (nullable) here: the near mask's words written from the lists' counts (bit == "list not empty": the same bits
 * sherf_build_cells2 computes) -- then sherf_build_cells2 can be given near_mask = NULL and skips its own mask pass. */
/* near_mask (nullable): uint32[32768]; one bit per sub-cell (edge cell/sub, sub in grid_hdr[8]): set iff the sub-cell's box
 * comes within cell_size of some vertex -- an unset bit proves "no vertex within the query radius". */

/* a4+a5+a6: sample_stratified (renderer.py:458-481, math_utils.py:101-118), sample positions and SMPL-frame
 * transform (renderer.py:304-310), nearest posed vertex + 5 cm shell mask (renderer.py:315-321) and stream
 * compaction, fused.  One wave per ray.
 *   in : ray_o/ray_d [R][3], near/far [R], Rg[9], Th[3] (params R, Th), cell list of the posed vertices.
 *   out: counters[0] = number of valid samples, counters[1..2] = ordered-int min/max of all depths (for the
 *        global clamp of ray_marcher.py:57); ray_base[R], ray_cnt[R]; for compact sample c:
 *        cs_idx[c] = ray*S + k, cs_vid[c] = vertex id, cs_xs[c][4] = (x_s, 0).  Order: ray-major, ascending k
 *        == the order of the reference's boolean-mask indexing (renderer.py:320), found by a two-pass
 *        count / scan / write so it is deterministic and needs no global atomics.
 *   workspace: dense_vid[R*S] int32, ray_mask[R*ceil(S/64)] u64, scan_ws[R + R/1024 + 2] int32 (the last word is the two-pass sampler's
 *   candidate count).  cs_xs doubles as the sampler's candidate list (capacity records of 16 bytes: x_s + dense index; taken when capacity >= R * S) before the compaction writes it:
 *   its contents on entry are lost.  S <= 256. */
/* near_hdr / near_list (both NULL or both set): the near lists of sherf_build_near_lists for the SAME cell list; the two-pass sampler's
 * search then tests one exact list per candidate instead of walking its cell neighbourhood (same results: the comparisons are the same). */
int sherf_sample_mask_nn(const float* ray_o, const float* ray_d, const float* near, const float* far,
                         int R, int S, const float* Rg, const float* Th, const float* grid_hdr,
                         const int32_t* cell_start, const float* cell_pts, const uint32_t* near_mask,
                         int64_t capacity, int32_t* counters, int32_t* ray_base, int32_t* ray_cnt, int32_t* cs_idx,
                         int32_t* cs_vid, float* cs_xs, int32_t* dense_vid, uint64_t* ray_mask,
                         int32_t* scan_ws, const int32_t* near_hdr, const uint16_t* near_list, sherf_stream_t stream);

/* a8+a9+a10 (geometry part): per compact sample: x_c, v_c via T2C[vid]; nearest T-pose vertex (exact, cell
 * list of t_vertices, renderer.py:627); uv via C2S.  geom[c][8] = (x_c.xyz, v_c.xyz, u, v); cs_tvid[c]. */
int sherf_warp_geom(const int32_t* counters, const int32_t* cs_idx, const int32_t* cs_vid, const float* cs_xs,
                    const float* ray_d, int S, const float* Rg, const float* T2C, const float* C2S,
                    const float* t_verts, const float* tgrid_hdr, const int32_t* tcell_start,
                    const float* tcell_pts, int64_t capacity, float* geom, int32_t* cs_tvid,
                    sherf_stream_t stream);

/* Sparse voxel level descriptor used by the gather (a11). All pointers device. */
typedef struct {
    const uint32_t* wp;      /* [n_words][2]: (32 occupancy bits, exclusive popcount prefix) per word; voxel index (z*H+y)*W+x */
    const float* rows;       /* folded features [n_rows][96] */
    int32_t D, H, W;
} sherf_vox_level;

/* a10+a11+a12 taps: pixel-aligned (renderer.py:330-340), tri-plane (:234-243) and voxel trilinear (:744-797)
 * gathers, with the linear layers conv1d_projection/conv1d_reprojection (renderer.py:350,423-424) folded into
 * the tables ahead of time (interpolation is linear).  Writes the transformer input tokens and extras in the
 * tile-major layout the MLP kernel consumes:
 *   tokens[tile][3][8][32][4] floats, extras[tile][12][32] floats (x_c, v_c, tapped rgb, pad), 32 samples/tile.
 *   planes_f [3][P][P][32], feat_f [Hf][Wf][64], img4 [H][W][4], tok_bias[96], bounds[6] (t_world_bounds),
 *   vox_min[3] (xyz of sp_input bounds min), vox_sh[3] (out_sh as z,y,x). */
int sherf_gather_tokens(const int32_t* counters, const float* geom, const float* planes_f, int P,
                        const float* feat_f, int Hf, int Wf, const float* img4, int H, int W,
                        const sherf_vox_level* levels_host, const float* tok_bias, const float* bounds,
                        const float* vox_min, const int32_t* vox_sh_host, int mode, int64_t capacity, float* tokens,
                        float* extras, sherf_stream_t stream);
/* mode 0: all taps; 1: tri-plane + pixel taps only (levels_host may be NULL) -- can run before the voxel encoder has
 * finished; 2: voxel taps only, ADDED onto the tokens written by a mode-1 pass.  `mode | 4`: the voxel-row loads of the 8 corners are
 * issued unconditionally (absent corners read row 0 with weight 0) instead of under one branch per corner -- same sums, a schedule
 * variant (opt-in, rendering_options['gather_branchless']); `mode | 12`: the same compiled for 4 waves / SIMD (128 VGPRs instead of 160).
 * `mode | 16`: planes_f, feat_f and the levels' rows hold fp16 (sherf_fold_tables(out_half), SHERF_FRAME_HALF_TABLES); img4, the
 * arithmetic and the tokens stay fp32.  `mode | part << 8 | nparts << 16` (nparts 2..255): only part `part` of the tile list cut into
 * `nparts` contiguous parts (see sherf_nerf_mlp_part). */

/* Per-frame re-layout NCHW -> channel-last with a 32x32 projection per texel (the linear part of
 * conv1d_reprojection, renderer.py:423-424, commuted with the interpolation):
 *   out[g*group_base + pix*pix_stride + o] = sum_c Wt[c][o] * in[(g*32 + c)*HW + pix],  g < groups.
 * planes [3*32][P*P] -> [3][P*P][32] (pix_stride 32, group_base P*P*32); feature map [2*32][HW] -> [HW][64]
 * (pix_stride 64, group_base 32). */
/* Backward of sherf_gather_tokens (BASELINE config 5): d_tokens (tile-major, like tokens) is scattered with the forward's
 * tap weights into the gradients of the folded tables -- d_planes_f [3][P][P][32], d_feat_f [Hf][Wf][64], d_rows{0,1,2}
 * [n_rows_l][96] of the three voxel levels -- and summed into d_tok_bias[3][32].  All outputs must be zeroed by the caller.
 * fp32 hardware atomics (order dependent in the last ulp).  The direct form: kept as the check of the binned one below, which the
 * training step uses (20.3 vs 4.2 ms at 512 x 512 x 64). */
int sherf_gather_tokens_bwd(const int32_t* counters, const float* geom, const float* d_tokens, int P, int Hf, int Wf,
                            int H, int W, const sherf_vox_level* levels_host, const float* bounds, const float* vox_min,
                            const int32_t* vox_sh_host, int64_t capacity, float* d_planes_f, float* d_feat_f,
                            float* d_rows0, float* d_rows1, float* d_rows2, float* d_tok_bias, sherf_stream_t stream);
/* The same scatter with the samples binned by the coarsest tapped voxel cell they fall in and accumulated per bin in LDS windows (the
 * three voxel levels and the tri-planes; the pixel-aligned taps stay direct): one device atomic per touched address per bin instead of
 * one per sample and tap (csrc/gather.hip).  Same contract and outputs (sums differ in the last ulp, as between any two runs of the
 * direct form); `scratch`: int32 words, at least what sherf_gather_bwd_scratch_words reports for (levels, capacity); not zeroed by the caller. */
int sherf_gather_tokens_bwd_binned(const int32_t* counters, const float* geom, const float* d_tokens, int P, int Hf, int Wf,
                                   int H, int W, const sherf_vox_level* levels_host, const float* bounds, const float* vox_min,
                                   const int32_t* vox_sh_host, int64_t capacity, float* d_planes_f, float* d_feat_f,
                                   float* d_rows0, float* d_rows1, float* d_rows2, float* d_tok_bias, int32_t* scratch,
                                   int64_t scratch_words, sherf_stream_t stream);
int sherf_gather_bwd_scratch_words(const sherf_vox_level* levels_host, int64_t capacity, int64_t* words_host);

int sherf_fold_tables(const float* in, const float* Wt, float* out, int HW, int groups, int pix_stride,
                      int64_t group_base, int out_half, sherf_stream_t stream);
/* out_half != 0: `out` is written as fp16 (same element offsets, 2 bytes each: the first half of the fp32-sized buffer) for
 * sherf_gather_tokens(mode | 16). */
/* obs image [3][HW] -> [HW][4] (rgb0) for the rgb tap of renderer.py:336 */
int sherf_img_to_hwc4(const float* img, float* out, int HW, sherf_stream_t stream);

/* a13+a14: rgb positional encoding -> slot-2 token, 3-token transformer (renderer.py:949-993), pos/view encodings (:875-916) and
 * NeRFDecoder (triplane.py:285-316) as ONE MFMA kernel (csrc/mlp.hip: 4-wave workgroups, two per CU, a 3-slot LDS ring of <= 20 KiB
 * weight steps, two independent accumulator chains per step); weights arrive as the pre-packed fragment stream built by
 * sherf_amd/mlp_pack.py FOR THE SAME `prec`.  prec: 1 = f16x3 (operands split hi + lo in fp16, three MFMAs per product, fp32
 * accumulate: fp32-grade, the default), 0 = bf16 (one product; north_star's nominal precision; on the seeded weights it misses the 1e-3 tolerance).
 * tokens [tile][3][8][32] float4 / extras [tile][12][32] float: 32 samples per tile (sherf_gather_tokens).  out[c] = (r,g,b,sigma). */
int sherf_nerf_mlp(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                   const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream);
/* `prec | SHERF_MLP_NO_TRANSFORMER` in any sherf_nerf_mlp* entry point (and in sherf_frame.mlp_prec): the renderer was built with use_trans = False
 * (renderer.py:261, 427) -- the slot-2 completion and the 3-token transformer are skipped, the decoder reads the fused tokens 0 / 1 as they are. */
#define SHERF_MLP_NO_TRANSFORMER 256
/* sherf_nerf_mlp on ONE contiguous part of the tile list: the compact tiles are cut into `nparts` parts at multiples of 8 tiles (256
 * samples; the cut is computed on the device from counters[0]) and this launch runs part `part` -- so that part k's network can run on
 * one stream beside part k + 1's sherf_gather_tokens (`mode | part << 8 | nparts << 16`: the same cut) on another.  nparts <= 1: everything. */
int sherf_nerf_mlp_part(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                        const float* wbias, int prec, int64_t capacity, float* out, int part, int nparts, sherf_stream_t stream);
/* sherf_nerf_mlp for the single-product precisions (prec 0, 2 only) with TWO 32-sample tiles per wave (csrc/mlp.hip: nerf_mlp2_kernel,
 * round 5): every weight fragment read from the LDS ring feeds two MFMAs, a ring step carries 16 MFMAs in four independent accumulator
 * chains, two 4-wave workgroups of eight tiles per CU.  Same inputs, same outputs bit for bit as sherf_nerf_mlp. */
int sherf_nerf_mlp2(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                    const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream);
/* sherf_nerf_mlp for the single-product precisions (prec 0, 2 only), one tile per wave, with every layer epilogue of the decoder (fp32 -> fp16
 * repack, ReLU, the next bias tiles) issued INSIDE the following ring step's MFMA stream on a second pair of accumulators (csrc/mlp.hip:
 * nerf_mlp3_kernel, round 5).  Same inputs, same outputs bit for bit as sherf_nerf_mlp. */
int sherf_nerf_mlp3(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                    const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream);
/* sherf_nerf_mlp3 on ONE contiguous part of the tile list (the cut of sherf_nerf_mlp_part / sherf_gather_tokens), with the launch's residency as
 * an argument: wgs_per_cu = 3 (or 0) what the kernel allows, 2 = a third of every SIMD's registers stays free for another kernel's waves (the
 * gather of the next part on a second stream).  Same results. */
int sherf_nerf_mlp3_part(const int32_t* counters, const float* tokens, const float* extras, const void* wstream, const float* wbias, int prec,
                         int64_t capacity, float* out, int part, int nparts, int wgs_per_cu, sherf_stream_t stream);
/* Round 6: the positional encodings outside the network kernel.  sherf_gather_tokens_pe = sherf_gather_tokens on fp16 tables (`mode | 16`, mode 0 or
 * 1) that also writes PE6(x_c) (renderer.py:875-916, 39 features -> 3 K-blocks), PE4(v_c) (27 -> 2) and PE5(rgb)[:32] (renderer.py:423; 2) of every
 * sample as fp16 MFMA B-operand fragments: pefrag[tile][q][lane] x 16 bytes (features 16 kb + 8 h .. + 7 of sample j, lane = 32 h + j; q = 0-2 PE6,
 * 3-4 PE4, 5-6 PE5; zero padded), ((capacity + 31) / 32) tiles x 7 KiB.  sherf_nerf_mlp3_pe = sherf_nerf_mlp3 (prec 2 only) reading them: same sin /
 * cos sequence and rounding as the kernel's own evaluation, so the outputs equal sherf_nerf_mlp3's on the same tokens / extras bit for bit. */
int sherf_gather_tokens_pe(const int32_t* counters, const float* geom, const float* planes_f, int P,
                           const float* feat_f, int Hf, int Wf, const float* img4, int H, int W,
                           const sherf_vox_level* levels_host, const float* tok_bias, const float* bounds,
                           const float* vox_min, const int32_t* vox_sh_host, int mode, int64_t capacity, float* tokens,
                           float* extras, void* pefrag, sherf_stream_t stream);
int sherf_nerf_mlp3_pe(const int32_t* counters, const float* tokens, const float* extras, const void* pefrag, const void* wstream,
                       const float* wbias, int prec, int64_t capacity, float* out, sherf_stream_t stream);
/* The same network as TWO launches (csrc/mlp.hip: nerf_tokens_kernel + nerf_decoder_kernel), results bit-identical to sherf_nerf_mlp:
 * launch 1 = slot-fusion remainder + 3-token transformer (renderer.py:423-427, 949-993), barrier-free with its weights resident in LDS;
 * launch 2 = NeRFDecoder (triplane.py:285-316) with every wave in the MFMA-bound phase.  zfrag: scratch for the fused tokens,
 * ((capacity + 31) / 32) tiles x 4 KiB (prec 0, 2) or 8 KiB (prec 1). */
int sherf_nerf_mlp_split(const int32_t* counters, const float* tokens, const float* extras, const void* wstream,
                         const float* wbias, int prec, int64_t capacity, void* zfrag, float* out, sherf_stream_t stream);
/* The weight stream for `prec` built on the device (what sherf_amd/mlp_pack.py: pack() builds on the host, bit for bit): slot i (2 bytes) of
 * stream_out = piece (src[i] & 1: 0 = hi, 1 = lo) of flat[src[i] >> 1], zero where src[i] < 0; bias_out[i] = flat[bias_src[i]] or 0.
 * `flat` = the parameters of mlp_pack.packed_names() concatenated, (src, bias_src) = mlp_pack.stream_index() (device copies).
 * *flag (device) is set to 0, then |= 1 if a packed value is not finite, |= 2 if one exceeds the fp16 range in an fp16 mode: the
 * caller reads it back and raises (the reference has no such failure mode: its Linear layers run in fp32, triplane.py:285-316). */
int sherf_mlp_pack_stream(const float* flat, const int32_t* src, int64_t n_slots, int prec, void* stream_out,
                          const int32_t* bias_src, int n_bias, float* bias_out, int32_t* flag, sherf_stream_t stream);
/* layout of the weight stream the kernel expects for `prec`: *n_steps steps; step_pieces_host[s] = its size in 1 KiB pieces (hi [, lo]
 * fragments, zero-padded to a multiple of 4); units[s * 10 + u] = chunk * 16 + K-block of the u-th (chunk, K-block) unit the kernel
 * consumes in step s (-1 = none / padding).  sherf_amd/mlp_pack.py restates it; tests/test_boundary.py compares the two. */
int sherf_mlp_stream_layout(int prec, int32_t* n_steps, int32_t* step_pieces_host, int32_t* units, int32_t max_steps);

/* a15+a16: scatter-back + MipRayMarcher2 (renderer.py:364-371, ray_marcher.py:25-64) on the compact samples;
 * masked-out samples (sigma=-80) contribute exact zeros so they are skipped.  rgb[R][3], depth[R], acc[R]. */
int sherf_composite_compact(const int32_t* counters, const int32_t* ray_base, const int32_t* ray_cnt,
                            const int32_t* cs_idx, const float* sample_out, const float* ray_d,
                            const float* near, const float* far, int R, int S, int white_back, float* rgb,
                            float* depth, float* acc, sherf_stream_t stream);
/* The same with a bound on the compact samples that sample_out holds (sherf_frame.tok_capacity): a ray whose samples reach beyond
 * tok_cap is written as NaN and counters[3] |= 2 instead of being composited from memory nobody wrote. */
int sherf_composite_compact_cap(int32_t* counters, const int32_t* ray_base, const int32_t* ray_cnt,
                                const int32_t* cs_idx, const float* sample_out, const float* ray_d,
                                const float* near, const float* far, int R, int S, int white_back, int64_t tok_cap,
                                float* rgb, float* depth, float* acc, sherf_stream_t stream);

/* MipRayMarcher2.forward on dense inputs (ray_marcher.py:67-70): colors[R][S][3], sigma[R][S], depths[R][S],
 * rays_d[R][3] -> rgb[R][3], depth[R], weights[R][S].  dminmax[2] (device) = global min/max of depths
 * (ray_marcher.py:57). */
/* Backward of sherf_composite_compact for a loss that reads rgb and acc (BASELINE config 5; the reference's losses do not
 * read the depth map, loss.py:103-176): d_rgb[R][3], d_acc[R] -> d_sample_out[capacity][4] = d/d(rgb, sigma) of every
 * compact sample (autograd of MipRayMarcher2.run_forward, ray_marcher.py:25-64, restricted to the valid samples).
 * Checked against autograd through the oracle on the CPU and on the MI355X (tests/test_gpu_backward.py). */
int sherf_composite_compact_bwd(const int32_t* ray_base, const int32_t* ray_cnt, const int32_t* cs_idx,
                                const float* sample_out, const float* ray_d, const float* near, const float* far,
                                int R, int S, int white_back, const float* d_rgb, const float* d_acc,
                                float* d_sample_out, sherf_stream_t stream);

int sherf_composite_dense(const float* colors, const float* sigma, const float* depths, const float* rays_d,
                          int R, int S, int white_back, const float* dminmax, float* rgb, float* depth,
                          float* weights, sherf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a11: sparse voxel encoder (SparseConvNet, renderer.py:708-871; spconv 2.3.3 semantics restated in
 * oracle/sherf_oracle.py).  Level bookkeeping is bitmap + popcount-prefix (rank == row id, rows sorted by
 * linear voxel index). */
int sherf_svox_mark_rows(const int32_t* coord, int n, int D, int H, int W, uint32_t* bitmap, sherf_stream_t stream);
int sherf_svox_mark_down(const int32_t* keys, const int32_t* n_rows, int D, int H, int W, uint32_t* bitmap_out,
                         int max_rows, sherf_stream_t stream);
int sherf_svox_scan(const uint32_t* bitmap, int n_words, int32_t* prefix, int32_t* n_rows, int32_t* chunk_ws,
                    uint32_t* wp, sherf_stream_t stream); /* chunk_ws: int32[n_words/1024 + 1]; wp: [n_words][2] */
int sherf_svox_keys(const uint32_t* bitmap, const int32_t* prefix, int n_words, int32_t* keys, sherf_stream_t stream);
/* g[row][C] = sum of the input rows in that voxel (acc_fix: zeroed int64[n*C] fixed-point scratch, order independent),
 * mult[row] = how many (zeroed by the caller). */
int sherf_svox_scatter_rows(const int32_t* coord, const float* feat, int n, int C, int D, int H, int W,
                            const uint32_t* bitmap, const int32_t* prefix, const int32_t* n_rows, int64_t* acc_fix,
                            float* g, int32_t* mult, sherf_stream_t stream);
/* Sparse 3x3x3 convolution on MFMA (bf16 operands split hi/lo, fp32 accumulate): out_raw = conv(act(in_raw)) with
 * act = relu(in_bn scale/shift) (+ (mult-1)*v0) applied while gathering (in_bn == NULL: raw input).
 * mode 0 submanifold, 1 stride-2 (k3 p1), 2 pointwise (folds the 1x1 projections into the tapped levels).
 * w_packed: bf16 fragments [ntaps][Cin/16][Cout/32][hi,lo][64 lanes][8] (sherf_amd.voxel.pack_conv_weights).
 * 32 output rows per workgroup; partials[grid][2][Cout] fp64 per-block sums of out and out^2 (NULL: none). */
int sherf_svox_conv3(const int32_t* keys_out, const int32_t* n_rows_out, int Do, int Ho, int Wo,
                     const uint32_t* wp_in, int Di, int Hi, int Wi, const float* in_raw, int Cin,
                     const float* in_bn, const int32_t* in_mult, const void* w_packed, int Cout, int mode,
                     int max_rows, float* out_raw, int64_t* out_acc, sherf_stream_t stream);
/* Input gradient of that convolution on the same kernel (BASELINE config 5): d_in[i][ci] = sum_k sum_co d_raw[o(i,k)][co] W[co][k][ci],
 * o(i,k) = the row at offset 1 - k from i (down == 0) or the coarse voxel (i + 1 - k) / 2 where whole (down == 1; spconv's stride-2
 * SparseConv3d backward, reference renderer.py:1071-1079 layers).  keys_i / n_rows_i / (Di,Hi,Wi): the level of the layer's INPUT
 * rows (those that receive), wp_o / (Do,Ho,Wo): the level of d_raw's rows.  w_packed_t: pack_conv_weights of wt[k'][co][ci] =
 * W[co][26 - k'][ci].  *d_raw_amax (device): bits of max |d_raw| (sherf_bwd_bn_relu writes it) -- gradients are O(1e-7), the fp16
 * operand split needs O(1): the rows are scaled by a power of two into range and the result scaled back, both exact. */
int sherf_svox_conv3_dgrad(const int32_t* keys_i, const int32_t* n_rows_i, int Di, int Hi, int Wi, const uint32_t* wp_o, int Do,
                           int Ho, int Wo, const float* d_raw, int Cout, const uint32_t* d_raw_amax, const void* w_packed_t,
                           int Cin, int down, int max_rows, float* d_in, sherf_stream_t stream);
/* out_acc (optional): [8][2][Cout] int64, zeroed by the caller; the conv adds 2^24-scaled sums of its output rows and of
 * their squares (8 interleaved sub-accumulators) -- the batch statistics of the BatchNorm that follows it. */
/* statistics over the reference's row set (n_total rows, the non-voxel rows being zeros) -> bnparam[3][C] =
 * (scale, shift, relu(shift)); training != 0: batch statistics into stats[2][C], else stats holds running stats. */
int sherf_svox_bn_finalize(const int64_t* acc, const int32_t* n_total_rows, int C, const float* gamma, const float* beta,
                           float* stats, int training, float* bnparam, sherf_stream_t stream);
/* nn.BatchNorm1d's train-mode side effect (renderer.py:812-871: BatchNorm1d(eps 1e-3, momentum 0.01) behind every conv) for all
 * layers of the encoder in one launch: running_mean / running_var <- (1 - momentum) old + momentum batch (variance unbiased by
 * n / (n - 1)), num_batches_tracked += 1.  The arguments are HOST arrays of n_layers (<= SHERF_SVOX_MAX_LAYERS) device pointers /
 * values: stats[l] = [2][C_l] batch mean, biased variance (written by the frame); n_rows[l] = the row count they were taken over. */
int sherf_svox_bn_running_update(int n_layers, const float* const* stats, float* const* running_mean, float* const* running_var,
                                 int64_t* const* num_batches_tracked, const int32_t* const* n_rows, const int32_t* channels,
                                 const float* momentum, sherf_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * a3: RaySampler.forward (training/volumetric_rendering/ray_sampler.py:24-61). cam2world[N][16], intr[N][9]. */
/* ---------------------------------------------------------------------------------------------
 * a11 as ONE native call.  The plan names every persistent buffer of the encoder (caller-owned, sized by the caller,
 * see sherf_amd/voxel.py: SparseConvNet._plan); sherf_svox_encode enqueues the whole chain -- 1 memset, level-0 build,
 * per layer [mark_down + scan] + sparse conv (BatchNorm statistics resolved in the consumer's prologue), and the fold of
 * each tapped level -- on `stream` without reading anything back.  Replaces SparseConvNet.forward (renderer.py:778-871)
 * + the three .dense() volumes.  levels_out_host[3] receives the tapped levels for sherf_gather_tokens.
 */
#define SHERF_SVOX_MAX_LAYERS 16
typedef struct {
    uint32_t* bitmap;        /* [n_words], inside the plan's zero region */
    int32_t* prefix;         /* [n_words] */
    int32_t* n_rows;         /* [1] */
    int32_t* chunk_ws;       /* [n_words/1024 + 2] */
    uint32_t* wp;            /* [n_words][2] */
    int32_t* keys;           /* [cap] */
    int32_t n_words, cap, D, H, W, pad_;
} sherf_svox_level_ws;
typedef struct {
    int32_t cin, cout, down, tap;   /* down: stride-2 SparseConv3d; tap: level is sampled after this layer */
    const void* wt;          /* packed MFMA fragments (voxel.py: pack_conv_weights) */
    const float* gamma;
    const float* beta;
    float* stats;            /* [2][cout] batch stats out (training) / running stats in (eval) */
    float* bnparam;          /* [3][cout] */
    float* out;              /* [cap][cout] raw conv output */
    int64_t* acc;            /* [8][2][cout] fixed-point (2^-24) sums of the output and its squares; inside the zero region */
} sherf_svox_layer;
typedef struct {
    sherf_svox_level_ws lev[4];
    sherf_svox_layer layers[SHERF_SVOX_MAX_LAYERS];
    int32_t n_layers, pad_;
    int64_t* acc_fix;        /* [N][32] fixed-point accumulators (inside the zero region) */
    float* g0;               /* [N][32] summed level-0 features */
    int32_t* mult;           /* [N] rows per voxel (inside the zero region) */
    const int32_t* n_total;  /* [1] == N */
    void* zero_ptr;          /* region cleared at the start of every frame */
    int64_t zero_bytes;
    const float* fold_mat[3]; /* packed [C_l -> 96] pointwise weights per tapped level */
    float* fold_rows[3];      /* [cap_l][96] */
} sherf_svox_plan;
int sherf_svox_encode(const sherf_svox_plan* plan, const int32_t* coord, const float* feat, int n, int training,
                      sherf_vox_level* levels_out_host, sherf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The whole of ImportanceRenderer.forward (renderer.py:286-398) as ONE native call: every pointer the frame touches is
 * named in sherf_frame; sherf_render_frame enqueues ~60-75 kernels on two (three) HIP streams (SMPL tables + voxel encoder on
 * `stream_side`, rays on `stream_main`), joined with events owned by the library (created once per device; the only
 * persistent state the library keeps).  Host cost is a few microseconds per launch instead of one interpreter round
 * trip each.  phase: bit0 = everything up to and including the NeRF MLP, bit1 = compositing (the caller may add
 * density noise to sample_out[:,3] in between, renderer.py:435-436).
 * frame->flags & SHERF_FRAME_EXACT_GRIDS (opt-in): the kernels after the compaction (warp, gather, MLP) are launched for the frame's
 * actual number of valid samples instead of `capacity`: the count is copied to pinned host memory after the compaction and the
 * call WAITS for it (one hipEventSynchronize, with the encoder chain already enqueued) before it enqueues the rest -- the
 * reference synchronises at the same point (boolean-mask indexing, renderer.py:320-321).  Results are identical.
 */
#define SHERF_FRAME_EXACT_GRIDS 1
/* frame->flags & SHERF_FRAME_HALF_TABLES (opt-in; sherf_amd.ImportanceRenderer sets it with the single-product MLP precisions): the
 * folded tri-plane / feature-map tables and the folded voxel rows are written and tapped as fp16 (round to nearest even; half the bytes
 * through L2 in the gather, which is bound there).  Same buffers: their first half is used. */
#define SHERF_FRAME_HALF_TABLES 2
/* frame->flags & SHERF_FRAME_ENCODER_SINGLE (opt-in, set together with HALF_TABLES by sherf_amd.ImportanceRenderer): the sparse
 * convolutions multiply single fp16 products (operands rounded to nearest even) instead of the three of the f16x3 split. */
#define SHERF_FRAME_ENCODER_SINGLE 4
/* frame->flags & SHERF_FRAME_MLP_SPLIT: the per-sample network runs as sherf_nerf_mlp_split (two launches; needs frame->zfrag) instead of
 * sherf_nerf_mlp.  Same results bit for bit; opt-in (measured slower than the one launch on the MI355X: profiles/r04_call_b_mlp_ablations.txt). */
#define SHERF_FRAME_MLP_SPLIT 8
/* frame->flags & SHERF_FRAME_MLP_TWO_TILES: the per-sample network of a single-product precision (mlp_prec 0, 2) runs as sherf_nerf_mlp2 (two tiles
 * per wave) instead of sherf_nerf_mlp; ignored for mlp_prec 1, for SHERF_FRAME_MLP_SPLIT and for mlp_parts > 1.  Same results bit for bit. */
#define SHERF_FRAME_MLP_TWO_TILES 32
/* frame->flags & SHERF_FRAME_MLP_PIPELINED: the per-sample network of a single-product precision (mlp_prec 0, 2) runs as sherf_nerf_mlp3 (one tile
 * per wave, the decoder's layer epilogues inside the next ring step's MFMA stream) instead of sherf_nerf_mlp; takes precedence over
 * SHERF_FRAME_MLP_TWO_TILES; ignored for mlp_prec 1, for SHERF_FRAME_MLP_SPLIT and for mlp_parts > 1.  Same results bit for bit. */
#define SHERF_FRAME_MLP_PIPELINED 64
/* frame->flags & SHERF_FRAME_REPORT_COUNT: counters[0] (the frame's valid samples) is copied to pinned memory right behind the sampler
 * and sherf_frame_count() returns it after waiting for THAT point of the frame only -- the warp, gather, network and compositing are
 * still in flight.  What a caller uses to check tok_capacity on a frame with new inputs without draining the GPU. */
#define SHERF_FRAME_REPORT_COUNT 16
/* frame->flags & SHERF_FRAME_PE_FRAGS (round 6; needs frame->pefrag, SHERF_FRAME_HALF_TABLES, mlp_prec 2 and SHERF_FRAME_MLP_PIPELINED, else ignored):
 * the positional encodings PE6(x_c), PE4(v_c), PE5(rgb) are written by the gather as fp16 MFMA operand fragments (sherf_gather_tokens_pe) and READ by the
 * network kernel (sherf_nerf_mlp3_pe) instead of evaluated in it -- the network kernel runs at the board's power cap, the gather waits on memory.  Same
 * operand bits: the frame is bit-identical. */
#define SHERF_FRAME_PE_FRAGS 128
typedef struct {
    /* SMPL (a7-a9) */
    const float* poses; const float* shapes;           /* [3][72], [3][10]: target, big-pose, observation */
    const float* J_template; const float* J_shapedirs; const int32_t* parents;
    const float* posedirs; const float* shapedirs; const float* weights;
    float* A; float* posefeat; float* PO; float* SO; float* T2C; float* C2S;
    const float* obs_R; const float* obs_Th; const float* cam_R; const float* cam_T; const float* cam_K;
    /* cell lists + sampling (a4-a6) */
    const float* verts; const float* Rg; const float* Th; const float* tverts;
    float* grid_hdr; int32_t* cell_start; float* cell_pts; int32_t* cell_scratch; uint32_t* near_mask;
    const float* ray_o; const float* ray_d; const float* near; const float* far;
    int32_t R, S; int64_t capacity;
    int32_t* counters; int32_t* ray_base; int32_t* ray_cnt; int32_t* cs_idx; int32_t* cs_vid; float* cs_xs;
    int32_t* dense_vid; uint64_t* ray_mask; int32_t* scan_ws;
    /* tables (a10, a12) */
    const float* planes; const float* Wa_t; float* planes_f; int32_t P, flags; /* SHERF_FRAME_* */
    const float* obs_feat; const float* Wb_t; float* feat_f; int32_t Hf, Wf;
    const float* obs_img; float* img4; int32_t H, W;
    /* warp + gather (a8-a12) */
    float* geom; int32_t* cs_tvid;
    const float* tok_bias; const float* bounds; const float* vox_min; int32_t vox_sh[3]; int32_t gather_split; /* bit 0: split passes, bit 1: branchless variant, bit 2: branchless in 128 VGPRs */
    float* tokens; float* extras;
    /* voxel encoder (a11) */
    const sherf_svox_plan* vox_plan; const int32_t* vox_coord; const float* vox_feat; int32_t vox_n, vox_training;
    /* MLP + compositing (a13-a16) */
    const void* wstream; const float* wbias; int32_t mlp_prec, mlp_parts; float* sample_out;   /* mlp_parts 2..8: gather + network in that many parts on two streams (sherf_nerf_mlp_part); 0, 1: whole */
    int32_t white_back;
    int32_t main_after_layer;   /* scheduling: -1 = both streams start at once; k >= 0 = the ray side starts once encoder
                                 * layer k is done (the encoder's small launches are slowed 3-5x by a co-running sampler) */
    float* rgb; float* depth; float* acc;
    void* zfrag;                /* scratch of sherf_nerf_mlp_split (SHERF_FRAME_MLP_SPLIT), else NULL */
    int32_t* near_hdr; uint16_t* near_list; int64_t near_list_cap;   /* sherf_build_near_lists buffers (NULL: the cell-walk search) */
    int64_t tok_capacity;       /* samples that geom / tokens / extras / sample_out hold (0: `capacity`).  The sampler's own buffers stay at
                                 * `capacity` (= R * S for the two-pass sampler); a frame with more valid samples than tok_capacity renders the
                                 * rays it cannot hold as NaN and sets counters[3] bit 1 -- the caller sizes from counters[0] (phase 4) */
    void* pefrag;               /* SHERF_FRAME_PE_FRAGS: ((tok_capacity + 31) / 32 + 8) tiles x 7 KiB for the encodings' fragments, else NULL */
    int32_t* sticky;            /* optional [3] (round 6): before a frame resets `counters` it folds what the previous frame left there into sticky[0] = max valid-sample
                                 * count, sticky[1] = OR of the flag words (counters[3]), sticky[2] += 1 -- a caller may then read its frames' flags every few
                                 * frames (one 32-byte copy + an event cost the caller's stream ~15 us per frame on the MI355X) instead of after every frame */
} sherf_frame;
int sherf_render_frame(const sherf_frame* frame, int phase, sherf_vox_level* levels_out_host, sherf_stream_t stream_main,
                       sherf_stream_t stream_side, sherf_stream_t stream_aux);
/* The valid-sample count of the last frame THE CALLING THREAD enqueued on the current device with SHERF_FRAME_REPORT_COUNT (waits for the
 * sampler of that frame, not for the frame; other threads' frames on the device have their own slots -- a ring of eight per device -- and the
 * wait holds no lock; a thread whose frame has been overtaken by eight later REPORT_COUNT frames on the device gets SHERF_EINVAL instead of
 * another frame's count -- every slot carries the sequence number of the frame that holds it).  State the library keeps per process: the join / report events and two pinned words per device, the profiling ring, and
 * this per-thread slot index; sherf_render_frame serialises ENQUEUES on a device with a mutex (the join events are shared), so it is thread-safe
 * but not lock-free -- the "no global mutable state" of SURVEY section 8(b) holds for every other entry point, not for the frame driver. */
int sherf_frame_count(int32_t* nv_host);
/* hipGraph replay of frames (round 6): the second consecutive sherf_render_frame call (phase 1 or 3) with the same descriptor bytes, encoder plan,
 * streams and debug words is captured from the caller's stream and replayed by one hipGraphLaunch from then on (a frame's launch sequence depends on
 * nothing else: no data-dependent value reaches the host).  Frames with SHERF_FRAME_REPORT_COUNT / _EXACT_GRIDS, profiled frames and new
 * descriptors are enqueued launch by launch as before.  OPT-IN (environment SHERF_FRAME_GRAPH=1, or sherf_frame_graphs(1); (0) turns it off and
 * drops every captured graph) and only for frames WITHOUT a third stream (stream_aux == NULL): measured on the MI355X a replayed frame takes the GPU
 * as long as an enqueued one (1.719 vs 1.722 ms; it saves ~0.25 ms of host time), the three-stream form is 25 us faster than either, and this
 * runtime's hipStreamEndCapture crashes on the three-stream capture (csrc/frame.hip).  sherf_frame_graph_stats: {captures, replays, eagerly
 * enqueued frames, failed captures} since load. */
int sherf_frame_graphs(int enable);
int sherf_frame_graph_stats(int64_t* stats_host, int32_t n);
/* phase: 1 = everything up to the per-sample network, 2 = compositing, 3 = both; 4 (alone) = the sampler only (cell lists, shell mask,
 * nearest vertex, compaction: counters[0] = the frame's number of valid samples) -- what a caller runs once to size tok_capacity. */
/* stream_aux (may be NULL): a third stream on which the occupancy structure of voxel levels 1-3 is built while the
 * level-0 convolutions run on stream_side. */
/* sizeof of {sherf_vox_level, sherf_svox_level_ws, sherf_svox_layer, sherf_svox_plan, sherf_frame} for binding checks */
int sherf_struct_sizes(int32_t* sizes_host, int32_t n);
/* HIP-event timeline of the frames issued by sherf_render_frame (roofline measurement on the launch streams, without a
 * profiler's launch overhead): enable != 0 starts recording into a ring of 64 frames; read synchronises and returns, per
 * frame (oldest first), SHERF_PROF_FIELDS floats in milliseconds:
 *   [0] host time of the enqueue call; then GPU times since the frame's first event: [1] SMPL tables done (side),
 *   [2] encoder done (side), [3] ray side reaches the encoder join, [4] gather done, [5] MLP done, [6] compositing done;
 *   [7] = [5] - [4] = duration of the sherf_nerf_mlp launch. */
#define SHERF_PROF_FIELDS 8
int sherf_profile_frames(int enable);
int sherf_profile_frames_read(float* ms_host, int32_t max_n, int32_t* n_host);

/* a17 / SURVEY 8(f) rank 1 -- the per-frame glue in front of the renderer call (triplane.py:105-137, 174-217), inference time:
 * sherf_vertex_features: per-vertex 32-d features of the observation view.  verts [V][3] observation-pose vertices; tri [F][3] faces
 *   and last_face [3][V] = highest face index listing the vertex in column c, or -1 (the deterministic reading of the reference's
 *   index-assignment normals, renderer.py:50-63; a constant of the SMPL asset); cam_R [9], cam_T [3], cam_K [9] of the observation
 *   camera; feat [64][Hf][Wf] and img [3][H][W] (NCHW as the encoders emit them; taps bilinear, zeros padding, align_corners=True, grid
 *   normalised by the image size); Wp [32][96], bp [32] = conv1d_projection.  -> f3d [V][32] (back-facing rows zero), front [V] (0/1).
 * sherf_voxelize: t_verts [V][3] canonical vertices, can [V][3] canonicalised observation vertices -> bounds [2][3] (min - 5 cm,
 *   max + 5 cm), coord [V][4] = (0, z, y, x) 5 mm voxel indices (round half to even), out_sh [3] = (ceil(extent / 0.005) | 31) + 1
 *   in (z, y, x) order (device memory: the caller reads the three integers back to size the sparse tensor). */
int sherf_vertex_features(const float* verts, const int32_t* tri, const int32_t* last_face, int V, const float* cam_R,
                          const float* cam_T, const float* cam_K, const float* feat, int Hf, int Wf, const float* img, int H, int W,
                          const float* Wp, const float* bp, float* f3d, uint8_t* front, sherf_stream_t stream);
int sherf_voxelize(const float* t_verts, const float* can, int V, float* bounds, int32_t* coord, int32_t* out_sh,
                   sherf_stream_t stream);

int sherf_ray_sampler(const float* cam2world, const float* intrinsics, int N, int res, float* origins,
                      float* dirs, sherf_stream_t stream);
/* a1+a2: get_rays + get_near_far + near/far packing (training/RenderPeople_dataset.py:14-27, 68-101, 121-134) on device, on the
 * reference's own precision ladder: float64 camera algebra -> float32 rays -> float64 slab test -> float32 near / far.
 * K_inv[9], Rc[9], Tc[3], bounds[6] (world min/max of the posed vertices +-5cm) are float64 like the dataset's numpy arrays;
 * a zero direction component comes back as 1e-8 (the reference patches the array it returns, :71). */
int sherf_dataset_rays(const double* K_inv, const double* Rc, const double* Tc, const double* bounds, int H, int W,
                       float* ray_o, float* ray_d, float* near, float* far, uint8_t* mask_at_box,
                       sherf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
