// The taps of one sample (rows a10-a12 of SURVEY.md section 8) for NQ of the 8 channel quads of a slot: the arithmetic of
// csrc/gather.hip's gather_tokens_kernel (same expressions, same order of additions per channel -- the tokens are bit-identical),
// callable from a kernel whose lanes own several quads of a sample.  The fused gather + MLP kernel (csrc/mlp.hip) calls it with
// NQ = 4: lane (j, h) of an MFMA column tile owns quads 2 i + h, i = 0..3, of sample j -- the D layout of its token tiles.
#pragma once
#include "common.h"

struct GatherArgs {
    const float* geom;            // [capacity][8]: x_c (3), v_c (3), pixel in the observation view (2)
    const float4* planes_f;       // [3][P][P][8 quads]
    const float4* feat_f;         // [Hf][Wf][16 quads]
    const float4* img4;           // [H][W] rgb0
    const float4* tok_bias;       // [3][8 quads]
    const float* bounds;          // [6]
    const float* vox_min;         // [3]
    sherf_vox_level lv[3];
    int P, Hf, Wf, H, W;
    int vox_d, vox_h, vox_w;      // out_sh (z, y, x)
};

__device__ __forceinline__ void gt_axpy4(float4& a, float w, const float4 v) {
    a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
}
__device__ __forceinline__ float gt_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// acc[s][k]: slot s, quad q[k]; starts from the slot biases.  rgb: the bilinear tap of the observation image (renderer.py:336).
template <int NQ>
__device__ __forceinline__ void gather_sample(const GatherArgs& ga, const float* __restrict__ gm, const int (&q)[NQ], float4 (&acc)[3][NQ],
                                              float4& rgb) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int k = 0; k < NQ; ++k) acc[s][k] = ga.tok_bias[8 * s + q[k]];
    rgb = make_float4(0.f, 0.f, 0.f, 0.f);
    const float xc[3] = {gm[0], gm[1], gm[2]};
    const int P = ga.P, Hf = ga.Hf, Wf = ga.Wf, H = ga.H, W = ga.W;
    // ---- tri-plane: renderer.py:234-243, align_corners=False, zeros padding ----
    float n[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) n[a] = 2.f * (xc[a] - ga.bounds[a]) / (ga.bounds[3 + a] - ga.bounds[a]) - 1.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#ifdef SHERF_GT_NO_PLANES
        break;
#endif
        const float g0 = p == 2 ? n[2] : n[0];                 // planes (x,y), (x,z), (z,y)
        const float g1 = p == 1 ? n[2] : n[1];
        float px = gt_clampf(((g0 + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
        float py = gt_clampf(((g1 + 1.f) * P - 1.f) * 0.5f, -2.f, (float)P + 1.f);
        float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
        int xi = (int)x0, yi = (int)y0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int xx = xi + dx, yy = yi + dy;
                if (xx >= 0 && xx < P && yy >= 0 && yy < P) {
                    float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                    const float4* t = ga.planes_f + ((size_t)(p * P + yy) * P + xx) * 8;
#pragma unroll
                    for (int k = 0; k < NQ; ++k) gt_axpy4(acc[p][k], w, t[q[k]]);
                }
            }
    }
    // ---- pixel-aligned feature + rgb: renderer.py:330-336, align_corners=True ----
    {
        float gx = 2.0f * gm[6] / (float)W - 1.0f, gy = 2.0f * gm[7] / (float)H - 1.0f;
        float px = gt_clampf((gx + 1.f) * 0.5f * (Wf - 1), -2.f, (float)Wf + 1.f);
        float py = gt_clampf((gy + 1.f) * 0.5f * (Hf - 1), -2.f, (float)Hf + 1.f);
        float x0 = floorf(px), y0 = floorf(py), fx = px - x0, fy = py - y0;
        int xi = (int)x0, yi = (int)y0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int xx = xi + dx, yy = yi + dy;
                if (xx >= 0 && xx < Wf && yy >= 0 && yy < Hf) {
                    float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                    const float4* t = ga.feat_f + ((size_t)yy * Wf + xx) * 16;
#pragma unroll
                    for (int k = 0; k < NQ; ++k) { gt_axpy4(acc[0][k], w, t[q[k]]); gt_axpy4(acc[1][k], w, t[8 + q[k]]); }
                }
            }
        px = gt_clampf((gx + 1.f) * 0.5f * (W - 1), -2.f, (float)W + 1.f);
        py = gt_clampf((gy + 1.f) * 0.5f * (H - 1), -2.f, (float)H + 1.f);
        x0 = floorf(px); y0 = floorf(py); fx = px - x0; fy = py - y0;
        xi = (int)x0; yi = (int)y0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int xx = xi + dx, yy = yi + dy;
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                    gt_axpy4(rgb, w, ga.img4[(size_t)yy * W + xx]);
                }
            }
    }
    // ---- sparse voxel levels: renderer.py:544-556 + 762-782, align_corners=True ----
#ifndef SHERF_GT_NO_VOXELS
    {
        float gz = ((xc[2] - ga.vox_min[2]) / 0.005f) / (float)ga.vox_d * 2.f - 1.f;   // out_sh = (D,H,W) = (z,y,x)
        float gy = ((xc[1] - ga.vox_min[1]) / 0.005f) / (float)ga.vox_h * 2.f - 1.f;
        float gx = ((xc[0] - ga.vox_min[0]) / 0.005f) / (float)ga.vox_w * 2.f - 1.f;
#pragma unroll 1
        for (int L = 0; L < 3; ++L) {      // not unrolled: one level's loads in flight at a time keeps the register count down
            const sherf_vox_level& lev = ga.lv[L];
            float px = gt_clampf((gx + 1.f) * 0.5f * (lev.W - 1), -2.f, (float)lev.W + 1.f);
            float py = gt_clampf((gy + 1.f) * 0.5f * (lev.H - 1), -2.f, (float)lev.H + 1.f);
            float pz = gt_clampf((gz + 1.f) * 0.5f * (lev.D - 1), -2.f, (float)lev.D + 1.f);
            float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
            float fx = px - x0, fy = py - y0, fz = pz - z0;
            int xi = (int)x0, yi = (int)y0, zi = (int)z0;
            // phase 1: the 8 occupancy records, issued back to back (independent loads)
            uint2 rec[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int xx = xi + (t & 1), yy = yi + ((t >> 1) & 1), zz = zi + (t >> 2);
                const bool inb = xx >= 0 && xx < lev.W && yy >= 0 && yy < lev.H && zz >= 0 && zz < lev.D;
                const int key = inb ? (zz * lev.H + yy) * lev.W + xx : 0;
                const uint2 rr = reinterpret_cast<const uint2*>(lev.wp)[key >> 5];
                const uint32_t bit = 1u << (key & 31);
                rec[t] = make_uint2((inb && (rr.x & bit)) ? 1u : 0u, rr.y + __popc(rr.x & (bit - 1u)));
            }
            // phase 2: rows of the occupied corners
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (rec[t].x) {
                    const float w = ((t & 1) ? fx : 1.f - fx) * (((t >> 1) & 1) ? fy : 1.f - fy) * ((t >> 2) ? fz : 1.f - fz);
                    const float4* r = reinterpret_cast<const float4*>(lev.rows) + (size_t)rec[t].y * 24;
#pragma unroll
                    for (int k = 0; k < NQ; ++k) {
                        gt_axpy4(acc[0][k], w, r[q[k]]);
                        gt_axpy4(acc[1][k], w, r[8 + q[k]]);
                        gt_axpy4(acc[2][k], w, r[16 + q[k]]);
                    }
                }
            }
        }
    }
#endif
}
