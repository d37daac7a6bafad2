// tfk_glow.h -- what the one-launch-per-coupling kernel (tfk_glow.hip) and the level kernel (tfk_glow_level.hip: several
// couplings per launch on rows held in the LDS) share: the geometry of a convolutional coupling, the packed weight
// layout, the conv3x3 -> ReLU -> MaxPool -> affine stage and the window arithmetic.
#pragma once

#include <cstdlib>

#include "tfk_common.h"

namespace tfk {

typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef float gf32x2 __attribute__((ext_vector_type(2)));

constexpr int kGlowMaxRows = 16;        // samples per tile = MFMA N
constexpr int kGlowMaxCh = 16;          // 1x1 convolution: target channels kept in registers
#ifndef TFK_GLOW_PB
#define TFK_GLOW_PB 2
#endif
constexpr int kGlowPB = TFK_GLOW_PB;     // target groups (of 16) per wave and step of the Linear / transform loop
constexpr int kGlowFrame = 32;          // ConvModifier's target height / width (classic.py:13-16)

struct GlowGeom {
    int c_in, hi, wi, oy, ox;           // source image; where the modifier's non-constant rectangle starts in the 32x32 frame
    int kh, kw, rh, rw, cm;             // modifier kernel (1 or 2 per axis); the rectangle's size (hi + kh - 1, wi + kw - 1); c_in kh kw
    int a0h, a0w, a0y0, a0x0;           // modifier output buffer (4 channels): dims, frame coordinates of its origin
    int p1h, p1w, p1y0, p1x0;           // pooled-1 window computed per sample (16x16 frame)
    int b1h, b1w, b1y0, b1x0;           // pooled-1 buffer (8 channels)
    int p2h, p2w, p2y0, p2x0;           // pooled-2 window (8x8 frame); its buffer is always 10x10 at (-1,-1)
    int off_p1, off_p2, off_p3, slot_floats;   // float offsets inside a slot
    int fixed_floats;                   // LDS floats in front of the slots
    int slots, tile_rows;               // G, NS
    int T, n_params, n_tiles, D;
    int kind, n_ch, hw;                 // 1x1 convolution: channels, pixels
    int cg1, cg2;                       // output channels per lane task in conv block 1 / 2
    int h_stride;                       // kind 1: floats per sample in the parameter buffer
    int skip;                           // timing ablations only (TFK_GLOW_SKIP): 1 no S0, 2 no conv blocks, 4 no Linear / transform
};

// offsets into the packed fp32 weights of a layer (see tfk.h: tfk_glow_layer.weights)
__host__ __device__ inline int gw_bm(int c_in) { return 4 * c_in; }
__host__ __device__ inline int gw_w1(int c_in) { return 4 * c_in + 4; }
__host__ __device__ inline int gw_b1(int c_in) { return gw_w1(c_in) + 4 * 8 * 9; }
__host__ __device__ inline int gw_w2(int c_in) { return gw_b1(c_in) + 24; }
__host__ __device__ inline int gw_b2(int c_in) { return gw_w2(c_in) + 8 * 8 * 9; }
__host__ __device__ inline int gw_w3(int c_in) { return gw_b2(c_in) + 24; }
__host__ __device__ inline int gw_b3(int c_in) { return gw_w3(c_in) + 8 * 4 * 9; }
__host__ __device__ inline int gw_m2(int c_in) { return gw_b3(c_in) + 4; }
__host__ __device__ inline int gw_total(int c_in) { return gw_m2(c_in) + 5; }

// conv3x3(pad 1) -> ReLU -> MaxPool2d(2) -> per-channel scale / shift for one window of pooled pixels, all resident
// slots at once.  Lane task = one pooled pixel x CG output channels; a wave's 64 tasks share the channel group, so its
// weights are wave-uniform: scalar loads, packed [ci][ky][kx][co] so that two neighbouring output channels are one SGPR
// pair and one v_pk_fma_f32 updates both (the patch value is broadcast by op_sel): 18 CG packed fmas per input channel
// (measured on MI355X, tools/micro/rates.hip: v_pk_fma_f32 55 TMAC/s against 31 for v_fmac_f32).  Input buffer: origin =
// (first conv row - 1, first conv column - 1), so the 4x4 patch of pooled pixel (ly, lx) of the window starts at buffer
// (2 ly, 2 lx); even row width: aligned float2 reads.
template <int CI, int CO, int CG, bool AFFINE>
__device__ __forceinline__ void conv_stage(float *slot0, int slot_floats, int in_off, int ih, int iw, int out_off,
                                           int oh, int ow, int oy0, int ox0, int ph, int pw, int G,
                                           const float *__restrict__ w, const float *__restrict__ bias,
                                           const float *__restrict__ sc, const float *__restrict__ sh)
{
    static_assert(CG % 2 == 0 && CO % CG == 0, "channel groups are whole SGPR pairs");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int per_slot = ph * pw, n = G * per_slot;
    const int chunks = (n + 63) >> 6;
    constexpr int NG = CO / CG, CP = CG / 2;
    const int plane_in = ih * iw, plane_out = oh * ow;
    for (int c = wave; c < chunks * NG; c += nw) {
        const int cgi = __builtin_amdgcn_readfirstlane(c / chunks);
        const int t = (c - cgi * chunks) * 64 + lane;
        const bool active = t < n;
        const int tt = active ? t : n - 1;
        const int slot = tt / per_slot, rem = tt - slot * per_slot;
        const int ly = rem / pw, lx = rem - ly * pw;
        const float *ip = slot0 + slot * slot_floats + in_off + (2 * ly) * iw + 2 * lx;
        gf32x2 acc[CP][4];
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) acc[cp][0] = acc[cp][1] = acc[cp][2] = acc[cp][3] = gf32x2{0.0f, 0.0f};
        const float *wg = w + cgi * CG;
#pragma unroll 1
        for (int ci = 0; ci < CI; ++ci) {
            float p[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 a = *reinterpret_cast<const float2 *>(ip + r * iw);
                const float2 b = *reinterpret_cast<const float2 *>(ip + r * iw + 2);
                p[r][0] = a.x, p[r][1] = a.y, p[r][2] = b.x, p[r][3] = b.y;
            }
            const float *wc = wg + ci * (9 * CO);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int cp = 0; cp < CP; ++cp) {
                        const gf32x2 wv = *reinterpret_cast<const gf32x2 *>(wc + (ky * 3 + kx) * CO + 2 * cp);
                        acc[cp][0] = __builtin_elementwise_fma(wv, gf32x2{p[ky][kx], p[ky][kx]}, acc[cp][0]);
                        acc[cp][1] = __builtin_elementwise_fma(wv, gf32x2{p[ky][kx + 1], p[ky][kx + 1]}, acc[cp][1]);
                        acc[cp][2] = __builtin_elementwise_fma(wv, gf32x2{p[ky + 1][kx], p[ky + 1][kx]}, acc[cp][2]);
                        acc[cp][3] = __builtin_elementwise_fma(wv, gf32x2{p[ky + 1][kx + 1], p[ky + 1][kx + 1]}, acc[cp][3]);
                    }
            ip += plane_in;
        }
        if (active) {
            float *op = slot0 + slot * slot_floats + out_off + (cgi * CG) * plane_out + (oy0 + ly) * ow + ox0 + lx;
#pragma unroll
            for (int co = 0; co < CG; ++co) {
                const int ch = cgi * CG + co, cp = co >> 1, e = co & 1;
                float m = fmaxf(fmaxf(acc[cp][0][e], acc[cp][1][e]), fmaxf(acc[cp][2][e], acc[cp][3][e])) + bias[ch];
                m = fmaxf(m, 0.0f);                       // max of ReLUs = ReLU of the max (bias shared by the window)
                if (AFFINE) m = fmaf(sc[ch], m, sh[ch]);
                op[co * plane_out] = m;
            }
        }
    }
}

template <int CI, int CO, bool AFFINE>
__device__ __forceinline__ void conv_stage_cg(int cg, float *slot0, int slot_floats, int in_off, int ih, int iw,
                                              int out_off, int oh, int ow, int oy0, int ox0, int ph, int pw, int G,
                                              const float *__restrict__ w, const float *__restrict__ bias,
                                              const float *__restrict__ sc, const float *__restrict__ sh)
{
    if (cg == 8) conv_stage<CI, CO, 8, AFFINE>(slot0, slot_floats, in_off, ih, iw, out_off, oh, ow, oy0, ox0, ph, pw, G, w, bias, sc, sh);
    else if (cg == 4) conv_stage<CI, CO, 4, AFFINE>(slot0, slot_floats, in_off, ih, iw, out_off, oh, ow, oy0, ox0, ph, pw, G, w, bias, sc, sh);
    else conv_stage<CI, CO, 2, AFFINE>(slot0, slot_floats, in_off, ih, iw, out_off, oh, ow, oy0, ox0, ph, pw, G, w, bias, sc, sh);
}

// lo + (hi - lo) * sigmoid(h) with (lo, hi) = (-2, 2); the caller hands in h * log2(e) (the rows of W_eff / b_eff arrive
// pre-multiplied), so the exponential is one v_exp_f32; s * 4 is exact, so the reference's multiply-then-add is one fma
__device__ __forceinline__ float bounded4(float h_log2e)
{
    return fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-h_log2e)), 4.0f, -2.0f);
}

// ---- host: windows and buffers --------------------------------------------------------------------------------
static inline int floor2(int v) { return v & ~1; }
static inline int ceil2(int v) { return (v + 1) & ~1; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

// windows and buffers from the image's rectangle (rows [oy, oy + hi), columns [ox, ox + wi) of the 32x32 frame)
[[maybe_unused]] static void glow_windows(GlowGeom &g)
{
    const int y0 = imax(g.oy - 1, 0), y1 = imin(g.oy + g.rh + 1, kGlowFrame);
    const int x0 = imax(g.ox - 1, 0), x1 = imin(g.ox + g.rw + 1, kGlowFrame);
    const int c1y0 = floor2(y0), c1y1 = ceil2(y1), c1x0 = floor2(x0), c1x1 = ceil2(x1);
    g.a0y0 = c1y0 - 1, g.a0x0 = c1x0 - 1, g.a0h = c1y1 - c1y0 + 2, g.a0w = c1x1 - c1x0 + 2;
    g.p1y0 = c1y0 / 2, g.p1x0 = c1x0 / 2, g.p1h = (c1y1 - c1y0) / 2, g.p1w = (c1x1 - c1x0) / 2;
    const int qy0 = imax(g.p1y0 - 1, 0), qy1 = imin(g.p1y0 + g.p1h + 1, 16);
    const int qx0 = imax(g.p1x0 - 1, 0), qx1 = imin(g.p1x0 + g.p1w + 1, 16);
    const int c2y0 = floor2(qy0), c2y1 = ceil2(qy1), c2x0 = floor2(qx0), c2x1 = ceil2(qx1);
    g.b1y0 = c2y0 - 1, g.b1x0 = c2x0 - 1, g.b1h = c2y1 - c2y0 + 2, g.b1w = c2x1 - c2x0 + 2;
    g.p2y0 = c2y0 / 2, g.p2x0 = c2x0 / 2, g.p2h = (c2y1 - c2y0) / 2, g.p2w = (c2x1 - c2x0) / 2;
    g.off_p1 = 4 * g.a0h * g.a0w;
    g.off_p2 = g.off_p1 + 8 * g.b1h * g.b1w;
    g.off_p3 = g.off_p2 + 800;
    g.slot_floats = g.off_p3 + 64;
}

constexpr int kGlowLdsBytes = 160 * 1024;

// validated geometry of one layer (no launch shape): shared by tfk_glow_plan / tfk_glow_coupling / the level packer
int glow_geometry_base(const tfk_glow_layer *L, int32_t D, GlowGeom &g, const char *fn);

}  // namespace tfk
