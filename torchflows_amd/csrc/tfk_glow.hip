// tfk_glow.hip -- one launch per coupling layer of the image / multiscale flows (config 5, AffineGlow).
//
// Replaces, for ONE convolutional coupling (multiscale/base.py:19-114), the whole chain
//   [ActNorm in front of it, layers.py:39-69, as a pending per-element map]
//   x_A = x[..., source_mask].view(constant_shape)                         layers_base.py:145-150
//   ConvModifier -> 3 x (conv3x3, ReLU, MaxPool2d(2), BatchNorm2d) -> ConvModifier -> Linear(100 -> n_params)
//                                                                          multiscale/conditioning/classic.py:8-122
//   h = lo + (hi - lo) * sigmoid(h)    (lo, hi) = (-2, 2)                  conditioning/transforms.py:107-113
//   Affine.forward / inverse (transformers/linear/affine.py:33-59)  or
//   Invertible1x1ConvolutionTransformer + LUTransformer (linear/convolution.py:33-64, linear/matrix.py:11-99)
//   z[..., target_mask] = ...                                              layers_base.py:151-152
// with every activation of the conditioner in the LDS and nothing but the rows crossing HBM: the source part is
// read once, the target part is read and written once, in place.
//
// What makes it cheap (all exact up to fp32 summation order):
//  * ConvModifier pads the (c, h, w) source image to (4, 32, 32) with a 1x1 convolution whose padding exceeds
//    kernel - 1: every output pixel outside the image's own rectangle is the modifier's bias, for every sample.
//    The three conv blocks therefore only differ from a per-layer constant BACKGROUND image (bg1: 8x16x16, bg2: 8x8x8,
//    evaluated once on the host) inside the window that the rectangle's receptive field reaches; only that window is
//    computed per sample (AffineGlow (3,32,32): 3.3 M of the reference's 8.7 M multiply-adds per sample and coupling).
//  * The second modifier pads (4, 4, 4) to (1, 10, 10): 84 of the Linear layer's 100 inputs are its bias; the host
//    folds them (and BatchNorm 3) into the Linear layer: h = W_eff (n_params x 16) v + b_eff.
//  * Squeeze / unsqueeze / chunk (multiscale/base.py:117-175, 271-280) are index arithmetic: the rows keep their
//    (c, h, w) layout from the first to the last layer and every coupling addresses them through two int32 tables
//    (source and target elements in the order the conditioner / transformer expect them).
//  * ActNorm layers are deferred by the host: every row element carries a pending map v = s * raw + t (composed in
//    double), applied where the element is read; a transformed element is stored in final form.
//
// Work split: a workgroup owns `tile_rows` <= 16 samples at a time; `slots` of them have their activations resident
// (LDS bytes per slot depend on the window), so the convolution stages run in tile_rows / slots rounds, lanes over
// (slot, pooled pixel) with the weights of the wave's output-channel group in SGPRs (scalar loads; every lane of
// a wave works on the same channels).  The 16 numbers per sample that feed the Linear layer are collected for the
// whole tile; then the Linear layer runs as v_mfma_f32_16x16x4_f32 tiles (16 parameters x 16 samples, K = 16), each
// lane receiving the 4 parameters = 2 affine targets of one sample, transforms them and accumulates the log-det.
#include "tfk_glow.h"

namespace tfk {

// S0 of k_glow_coupling: the source elements of G resident samples (gathered through the layer's table, pending maps applied)
// through the first ConvModifier into the modifier buffer's rectangle.  ONE: the 1x1 kernel (every BASELINE shape), taps and
// bounds tests compiled out (the general form costs the 1x1 case 5 %, same-box A/B).
template <bool ONE>
__device__ __forceinline__ void s0_modifier(const GlowGeom &g, const float *rows, long long N, long long row_first, int G,
                                            float *slot0, const int *__restrict__ src_idx,
                                            const float2 *__restrict__ src_st, const float *__restrict__ wts)
{
    const int npix = g.hi * g.wi, rpix = g.rh * g.rw;
    const float *bm = wts + gw_bm(g.cm);
    const int taps = ONE ? 1 : g.kh * g.kw;
    for (int t = threadIdx.x; t < G * rpix; t += blockDim.x) {
        const int slot = t / rpix, pix = t - slot * rpix;
        const int iy = pix / g.rw, ix = pix - iy * g.rw;
        long long row = row_first + slot;
        if (row > N - 1) row = N - 1;
        const float *xr = rows + row * g.D;
        float o0 = bm[0], o1 = bm[1], o2 = bm[2], o3 = bm[3];
        for (int tap = 0; tap < taps; ++tap) {
            bool inside = true;
            int spix = pix;
            if (!ONE) {
                const int ky = tap / g.kw, kx = tap - ky * g.kw;
                const int sy = iy - (g.kh - 1) + ky, sx = ix - (g.kw - 1) + kx;      // out[Y] = sum_k W[k] x[Y + k - pad]
                inside = sy >= 0 && sy < g.hi && sx >= 0 && sx < g.wi;
                spix = inside ? sy * g.wi + sx : 0;
            }
            for (int c0 = 0; c0 < g.c_in; c0 += 4) {             // four channels' loads in flight together
                int idx[4];
                float2 st[4];
                float raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + u < g.c_in ? c0 + u : g.c_in - 1;
                    const int e = c * npix + spix;
                    idx[u] = src_idx[e];
                    st[u] = src_st[e];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = xr[idx[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + u;
                    const bool on = c < g.c_in;
                    const int w = (on ? c : 0) * taps + tap;
                    const float v = (on && inside) ? fmaf(st[u].x, raw[u], st[u].y) : 0.0f;
                    o0 = fmaf(wts[w], v, o0);
                    o1 = fmaf(wts[g.cm + w], v, o1);
                    o2 = fmaf(wts[2 * g.cm + w], v, o2);
                    o3 = fmaf(wts[3 * g.cm + w], v, o3);
                }
            }
        }
        float *a = slot0 + slot * g.slot_floats + (g.oy + iy - g.a0y0) * g.a0w + (g.ox + ix - g.a0x0);
        const int plane = g.a0h * g.a0w;
        a[0] = o0, a[plane] = o1, a[2 * plane] = o2, a[3 * plane] = o3;
    }
}

template <int KIND, bool INV>
__global__ __launch_bounds__(1024) void k_glow_coupling(float *rows, float *logdet, long long N, GlowGeom g,
                                                        const int *__restrict__ src_idx,
                                                        const float2 *__restrict__ src_st,
                                                        const int *__restrict__ tgt_idx,
                                                        const float2 *__restrict__ tgt_st,
                                                        const float *__restrict__ wts, const float *__restrict__ bg1,
                                                        const float *__restrict__ bg2, const float4 *__restrict__ w_eff,
                                                        const float4 *__restrict__ b_eff)
{
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    float *V = lds;                                   // [16 samples][16]
    float *ldpart = lds + 256;                        // [16 waves][16 samples]
    float *Hs = lds + 512;                            // kind 1: [16 samples][h_stride]
    float *slot0 = lds + g.fixed_floats;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int G = g.slots;
    const int D = g.D;
    const float *bm = wts + gw_bm(g.cm);

    // ---- once per workgroup: everything of the buffers that no sample changes ----
    {
        const int a0 = 4 * g.a0h * g.a0w, b1 = 8 * g.b1h * g.b1w;
        for (int i = tid; i < G * a0; i += nthr) {
            const int slot = i / a0, e = i - slot * a0;
            const int ch = e / (g.a0h * g.a0w), r = (e / g.a0w) % g.a0h, c = e % g.a0w;
            const int y = g.a0y0 + r, x = g.a0x0 + c;
            const bool in = y >= 0 && y < kGlowFrame && x >= 0 && x < kGlowFrame;
            slot0[slot * g.slot_floats + e] = in ? bm[ch] : 0.0f;
        }
        for (int i = tid; i < G * b1; i += nthr) {
            const int slot = i / b1, e = i - slot * b1;
            const int ch = e / (g.b1h * g.b1w), r = (e / g.b1w) % g.b1h, c = e % g.b1w;
            const int y = g.b1y0 + r, x = g.b1x0 + c;
            const bool in = y >= 0 && y < 16 && x >= 0 && x < 16;
            slot0[slot * g.slot_floats + g.off_p1 + e] = in ? bg1[(ch * 16 + y) * 16 + x] : 0.0f;
        }
        for (int i = tid; i < G * 800; i += nthr) {
            const int slot = i / 800, e = i - slot * 800;
            const int ch = e / 100, r = (e / 10) % 10, c = e % 10;
            const int y = r - 1, x = c - 1;
            const bool in = y >= 0 && y < 8 && x >= 0 && x < 8;
            slot0[slot * g.slot_floats + g.off_p2 + e] = in ? bg2[(ch * 8 + y) * 8 + x] : 0.0f;
        }
    }
    __syncthreads();

    const long long n_tiles_rows = (N + g.tile_rows - 1) / g.tile_rows;
    for (long long tile = blockIdx.x; tile < n_tiles_rows; tile += gridDim.x) {
        const long long row_base = tile * g.tile_rows;
        const int nrows = (int)((N - row_base) < (long long)g.tile_rows ? (N - row_base) : (long long)g.tile_rows);

        for (int s0 = 0; s0 < g.tile_rows; s0 += G) {
            // ---- S0: pending map + ConvModifier (c_in -> 4 channels; kernel 1x1, or 2 wide along an axis whose padding is
            //      odd, classic.py:26-33) into the rectangle of the frame that is not the bias ----
            if (!(g.skip & 1)) {
                if (g.kh * g.kw == 1) s0_modifier<true>(g, rows, N, row_base + s0, G, slot0, src_idx, src_st, wts);
                else s0_modifier<false>(g, rows, N, row_base + s0, G, slot0, src_idx, src_st, wts);
            }
            __syncthreads();
            // ---- S1 .. S3: the three conv blocks on their windows ----
            const int ci = g.cm;
            if (!(g.skip & 2)) {
            conv_stage_cg<4, 8, true>(g.cg1, slot0, g.slot_floats, 0, g.a0h, g.a0w, g.off_p1, g.b1h, g.b1w,
                                      g.p1y0 - g.b1y0, g.p1x0 - g.b1x0, g.p1h, g.p1w, G, wts + gw_w1(ci),
                                      wts + gw_b1(ci), wts + gw_b1(ci) + 8, wts + gw_b1(ci) + 16);
            __syncthreads();
            conv_stage_cg<8, 8, true>(g.cg2, slot0, g.slot_floats, g.off_p1, g.b1h, g.b1w, g.off_p2, 10, 10,
                                      g.p2y0 + 1, g.p2x0 + 1, g.p2h, g.p2w, G, wts + gw_w2(ci), wts + gw_b2(ci),
                                      wts + gw_b2(ci) + 8, wts + gw_b2(ci) + 16);
            __syncthreads();
            conv_stage<8, 4, 2, false>(slot0, g.slot_floats, g.off_p2, 10, 10, g.off_p3, 4, 4, 0, 0, 4, 4, G,
                                       wts + gw_w3(ci), wts + gw_b3(ci), nullptr, nullptr);
            __syncthreads();
            }
            // ---- S4: BatchNorm 3 + second ConvModifier (4 -> 1 channel), folded on the host ----
            for (int t = tid; t < G * 16; t += nthr) {
                const int slot = t >> 4, p = t & 15;
                const float *p3 = slot0 + slot * g.slot_floats + g.off_p3 + p;
                const float *m2 = wts + gw_m2(ci);
                float v = m2[4];
                v = fmaf(m2[0], p3[0], v);
                v = fmaf(m2[1], p3[16], v);
                v = fmaf(m2[2], p3[32], v);
                v = fmaf(m2[3], p3[48], v);
                V[(s0 + slot) * 16 + p] = v;
            }
            // (the next round's S0 writes A0, last read before two barriers; P3 is rewritten after three more)
        }
        __syncthreads();

        // ---- Linear layer on the matrix cores + bounded output + transform ----
        const int j = lane & 15, q = lane >> 4;
        float bq[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[ks] = V[j * 16 + 4 * ks + q];
        if (g.skip & 4) {
        } else if (KIND == 0 || KIND == 2) {
            // (KIND 2, Shift, affine.py:137-159: one tile per group -- the shifts --, z = x +/- h, log-det 0)
            constexpr bool SHIFT = KIND == 2;
            constexpr int TPG = SHIFT ? 1 : 2;               // MFMA tiles per group of 16 targets
            // Samples on the MFMA's M axis, parameters on N: tile 2 m holds the scale logits of targets 16 m .. 16 m + 15
            // (sorted by physical position by the host), tile 2 m + 1 their shifts, so lane (q, jj) receives u and beta
            // of target 16 m + jj for the four samples 4 q + r -- every row access of a wave instruction is four rows x
            // 16 neighbouring targets (whole 128-byte lines), not 16 rows x 8 bytes.  kGlowPB target groups per step: their
            // operands, tables and row elements are requested together (the loop is bound by memory latency otherwise).
            const int n_pairs = g.n_tiles / TPG;
            const float *b_eff1 = reinterpret_cast<const float *>(b_eff);
            float ldr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float *xr0 = rows + (row_base + 4 * q) * D;
            bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ok[r] = 4 * q + r < nrows;
            for (int pb = kGlowPB * wave; pb < n_pairs; pb += kGlowPB * nw) {
                float4 au[kGlowPB], ab[kGlowPB];
                float bu[kGlowPB], bb[kGlowPB], x[kGlowPB][4];
                float2 pst[kGlowPB];
                int ph[kGlowPB];
#pragma unroll
                for (int u = 0; u < kGlowPB; ++u) {
                    const int m = pb + u < n_pairs ? pb + u : n_pairs - 1;
                    au[u] = w_eff[(TPG * m) * 64 + lane];
                    ab[u] = w_eff[(TPG * m + TPG - 1) * 64 + lane];
                    bu[u] = b_eff1[16 * TPG * m + j];
                    bb[u] = b_eff1[16 * TPG * m + 16 * (TPG - 1) + j];
                    ph[u] = tgt_idx[16 * m + j];                 // (tables padded to whole groups of 16)
                    pst[u] = tgt_st[16 * m + j];
                }
#pragma unroll
                for (int u = 0; u < kGlowPB; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][r] = ok[r] ? xr0[(long long)r * D + ph[u]] : 0.0f;
#pragma unroll
                for (int u = 0; u < kGlowPB; ++u) {
                    gf32x4 hu = {bu[u], bu[u], bu[u], bu[u]}, hb = {bb[u], bb[u], bb[u], bb[u]};
                    if (!SHIFT) hu = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[0], au[u].x, hu, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[0], ab[u].x, hb, 0, 0, 0);
                    if (!SHIFT) hu = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[1], au[u].y, hu, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[1], ab[u].y, hb, 0, 0, 0);
                    if (!SHIFT) hu = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[2], au[u].z, hu, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[2], ab[u].z, hb, 0, 0, 0);
                    if (!SHIFT) hu = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[3], au[u].w, hu, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[3], ab[u].w, hb, 0, 0, 0);
                    const bool tgt_ok = pb + u < n_pairs && 16 * (pb + u) + j < g.T;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float beta = bounded4(hb[r]);
                        const float v = fmaf(pst[u].x, x[u][r], pst[u].y);
                        float out, wl = 0.0f;
                        if (SHIFT) {
                            out = INV ? v - beta : v + beta;
                        } else {
                            const float u_ = bounded4(hu[r]);
                            wl = fmaf(u_, 0.5f, kAffC0);                  // affine.py:33-34, log(alpha) up to 1e-10
                            const float alpha = __builtin_amdgcn_exp2f(wl * __int_as_float(0x3fb8aa3b)) + kAffMinScale;   // (|wl| <= 1: 1 ulp)
                            out = INV ? (v - beta) * __builtin_amdgcn_rcpf(alpha) : alpha * v + beta;
                        }
                        if (tgt_ok && ok[r]) {
                            xr0[(long long)r * D + ph[u]] = out;
                            ldr[r] += INV ? -wl : wl;
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) ldr[r] += __shfl_xor(ldr[r], o, kWave);
                if (j == 0) ldpart[wave * 16 + 4 * q + r] = ldr[r];
            }
            __syncthreads();
            if (tid < nrows) {
                float s = 0.0f;
                for (int w = 0; w < nw; ++w) s += ldpart[w * 16 + tid];
                logdet[row_base + tid] += s;
            }
        } else {
            // parameters of the LU factors per sample into the LDS: [diag logits | U above | L below] (matrix.py:22-51)
            for (int tl = wave; tl < g.n_tiles; tl += nw) {
                const float4 a = w_eff[tl * 64 + lane];
                const float4 b = b_eff[tl * 4 + q];
                gf32x4 acc = {b.x, b.y, b.z, b.w};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[3], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 16 * tl + 4 * q + r;
                    float hv = bounded4(acc[r]);
                    hv = e < g.n_ch ? expf(hv) / 10.0f + 1.0f : hv / 10.0f;      // matrix.py:31-36
                    Hs[j * g.h_stride + e] = hv;
                }
            }
            __syncthreads();
            const int n = g.n_ch, HW = g.hw;
            const int n_off = n * (n - 1) / 2;
            for (int t = tid; t < nrows * HW; t += nthr) {
                const int s = t / HW, p = t - s * HW;
                float *xr = rows + (row_base + s) * D;
                const float *hr = Hs + s * g.h_stride;
                float v[kGlowMaxCh];
                int pos[kGlowMaxCh];
                // four channels at a time, indices clamped into the tables: the table reads and then the row reads of a group
                // leave together (a branch per channel serialises them: table -> row element, n times over)
#pragma unroll
                for (int c0 = 0; c0 < kGlowMaxCh; c0 += 4)
                    if (c0 < n) {
                        float2 st[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = c0 + u < n ? c0 + u : n - 1;
                            pos[c0 + u] = tgt_idx[c * HW + p];
                            st[u] = tgt_st[c * HW + p];
                        }
                        float raw[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) raw[u] = xr[pos[c0 + u]];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[c0 + u] = fmaf(st[u].x, raw[u], st[u].y);
                    }
                // U entry (r, c), r < c: hr[n + r n - r (r + 1) / 2 + (c - r - 1)];  L entry (r, c), c < r:
                // hr[n + n_off + r (r - 1) / 2 + c]   (triu_indices / tril_indices order, matrix.py:40-48)
                if (!INV) {
#pragma unroll
                    for (int r = 0; r < kGlowMaxCh; ++r)
                        if (r < n) {
                            float acc = hr[r] * v[r];
                            const int base = n + r * n - (r * (r + 1)) / 2 - r - 1;
#pragma unroll
                            for (int c = 0; c < kGlowMaxCh; ++c)
                                if (c > r && c < n) acc = fmaf(hr[base + c], v[c], acc);
                            v[r] = acc;
                        }
#pragma unroll
                    for (int rr = 0; rr < kGlowMaxCh; ++rr) {
                        const int r = kGlowMaxCh - 1 - rr;
                        if (r < n) {
                            float acc = v[r];
                            const int base = n + n_off + (r * (r - 1)) / 2;
#pragma unroll
                            for (int c = 0; c < kGlowMaxCh; ++c)
                                if (c < r) acc = fmaf(hr[base + c], v[c], acc);
                            v[r] = acc;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < kGlowMaxCh; ++r)
                        if (r < n) {
                            float acc = v[r];
                            const int base = n + n_off + (r * (r - 1)) / 2;
#pragma unroll
                            for (int c = 0; c < kGlowMaxCh; ++c)
                                if (c < r) acc = fmaf(-hr[base + c], v[c], acc);
                            v[r] = acc;
                        }
#pragma unroll
                    for (int rr = 0; rr < kGlowMaxCh; ++rr) {
                        const int r = kGlowMaxCh - 1 - rr;
                        if (r < n) {
                            float acc = v[r];
                            const int base = n + r * n - (r * (r + 1)) / 2 - r - 1;
#pragma unroll
                            for (int c = 0; c < kGlowMaxCh; ++c)
                                if (c > r && c < n) acc = fmaf(-hr[base + c], v[c], acc);
                            v[r] = acc / hr[r];
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < kGlowMaxCh; ++c)
                    if (c < n) xr[pos[c]] = v[c];
                if (p == 0) {                                   // sum_r log U_rr, ONCE per sample
                    float s_ld = 0.0f;
                    for (int r = 0; r < n; ++r) s_ld += logf(hr[r]);
                    logdet[row_base + s] += INV ? -s_ld : s_ld;
                }
            }
        }
        __syncthreads();                                        // V, ldpart, Hs are free again
    }
}

// z = s * raw + t per column, in place: the flush of the pending maps behind the last coupling
__global__ __launch_bounds__(kBlock) void k_rows_fma(float *rows, const float2 *__restrict__ st, long long N, int D4)
{
    const long long total = N * D4;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int c = (int)(i % D4);
        float4 v = reinterpret_cast<float4 *>(rows)[i];
        const float4 a = reinterpret_cast<const float4 *>(st)[2 * c], b = reinterpret_cast<const float4 *>(st)[2 * c + 1];
        v.x = fmaf(a.x, v.x, a.y), v.y = fmaf(a.z, v.y, a.w);
        v.z = fmaf(b.x, v.z, b.y), v.w = fmaf(b.z, v.w, b.w);
        reinterpret_cast<float4 *>(rows)[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void k_rows_fma1(float *rows, const float2 *__restrict__ st, long long N, int D)
{
    const long long total = N * D;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const float2 a = st[(int)(i % D)];
        rows[i] = fmaf(a.x, rows[i], a.y);
    }
}



// Launch shape.  Measured on MI355X over every coupling geometry of AffineGlow((3, 32, 32)) at 65 536 rows
// (tools/glow_tune.py, gpurun_out/glow_tune2.log): 256-thread workgroups, three to a CU, each with as many resident
// samples as 1/3 of the LDS holds (a power of two, so that tiles of 16 samples divide evenly), 4 output channels per
// lane task -- within 3 % of the best shape of the sweep for every geometry; larger workgroups lose to the barrier
// between the conv stages, 2 channels per task to the LDS reads of the patch, 8 to lane tasks that do not fill the waves.
// Any field the caller sets (> 0) is kept.
static bool glow_plan(GlowGeom &g, int want_slots, int want_block, int want_cg1, int want_cg2, int *block_out)
{
    const int fixed_bytes = 4 * g.fixed_floats, slot_bytes = 4 * g.slot_floats;
    const int block = want_block ? want_block : 256;
    if (block != 128 && block != 256 && block != 512 && block != 1024) return false;
    const int budget = want_slots ? kGlowLdsBytes - 1024 : (block == 128 ? kGlowLdsBytes / 6 : block == 256 ? kGlowLdsBytes / 3 : block == 512 ? kGlowLdsBytes / 2 : kGlowLdsBytes) - 1024;
    int slots = want_slots;
    if (!slots) {
        slots = 1;
        while (2 * slots <= kGlowMaxRows && fixed_bytes + 2 * slots * slot_bytes <= budget) slots *= 2;
    }
    if (slots < 1 || slots > kGlowMaxRows || fixed_bytes + slots * slot_bytes > kGlowLdsBytes - 1024) return false;
    const int cg1 = want_cg1 ? want_cg1 : 4, cg2 = want_cg2 ? want_cg2 : 4;
    if ((cg1 != 8 && cg1 != 4 && cg1 != 2) || (cg2 != 8 && cg2 != 4 && cg2 != 2)) return false;
    g.slots = slots, g.cg1 = cg1, g.cg2 = cg2;
    g.tile_rows = slots * (kGlowMaxRows / slots);
    *block_out = block;
    return true;
}

}  // namespace tfk

using namespace tfk;

namespace tfk {

int glow_geometry_base(const tfk_glow_layer *L, int32_t D, GlowGeom &g, const char *fn)
{
    if (!L) return fail(TFK_EINVAL, "%s: null layer", fn);
    if (L->kind < 0 || L->kind > 2) return fail(TFK_EINVAL, "%s: kind %d (0 affine, 1 invertible 1x1 convolution, 2 shift)", fn, L->kind);
    const int kh = L->kh > 0 ? L->kh : 1, kw = L->kw > 0 ? L->kw : 1;
    if (kh > 2 || kw > 2) return fail(TFK_EINVAL, "%s: modifier kernel %dx%d (1 or 2 per axis)", fn, kh, kw);
    if (L->c_in < 1 || L->hi < 1 || L->wi < 1 || L->oy < 0 || L->ox < 0 || L->oy + L->hi + kh - 1 > kGlowFrame ||
        L->ox + L->wi + kw - 1 > kGlowFrame)
        return fail(TFK_EINVAL, "%s: source image (%d, %d, %d), kernel %dx%d, at (%d, %d) does not fit the %dx%d frame", fn,
                    L->c_in, L->hi, L->wi, kh, kw, L->oy, L->ox, kGlowFrame, kGlowFrame);
    if (L->T < 1 || L->n_params < 1 || D < 1) return fail(TFK_EINVAL, "%s: T = %d, n_params = %d, D = %d", fn, L->T, L->n_params, D);
    if (L->kind == 2 && L->n_params != L->T)
        return fail(TFK_EINVAL, "%s: a shift coupling of %d targets takes %d parameters, got %d", fn, L->T, L->T, L->n_params);
    if (L->kind == 0 && L->n_params != 2 * L->T)
        return fail(TFK_EINVAL, "%s: an affine coupling of %d targets takes %d parameters, got %d", fn, L->T, 2 * L->T, L->n_params);
    if (L->kind == 1 && (L->n_ch < 1 || L->n_ch > kGlowMaxCh || L->hw < 1 || L->n_ch * L->hw != L->T ||
                         L->n_params != L->n_ch + L->n_ch * (L->n_ch - 1)))
        return fail(TFK_EINVAL, "%s: 1x1 convolution with %d channels x %d pixels (T = %d, n_params = %d): need "
                    "1 <= channels <= %d, T = channels * pixels, n_params = n + n (n - 1)", fn, L->n_ch, L->hw, L->T,
                    L->n_params, kGlowMaxCh);
    g = GlowGeom{};
    g.c_in = L->c_in, g.hi = L->hi, g.wi = L->wi, g.oy = L->oy, g.ox = L->ox;
    g.kh = kh, g.kw = kw, g.rh = L->hi + kh - 1, g.rw = L->wi + kw - 1, g.cm = L->c_in * kh * kw;
    g.T = L->T, g.n_params = L->n_params, g.D = D;
    g.n_tiles = L->kind == 0 ? 2 * ((L->T + 15) / 16) : (L->n_params + 15) / 16;        // (shift: one tile per 16 targets)
    g.kind = L->kind, g.n_ch = L->n_ch, g.hw = L->hw;
    g.h_stride = g.n_tiles * 16 + 1;
    g.fixed_floats = 512 + (L->kind == 1 ? ((kGlowMaxRows * g.h_stride + 3) & ~3) : 0);
    glow_windows(g);
    {
        const char *e = getenv("TFK_GLOW_SKIP");
        g.skip = e ? atoi(e) : 0;
    }
    return TFK_OK;
}

}  // namespace tfk

namespace {

int glow_geometry(const tfk_glow_layer *L, int32_t D, GlowGeom &g, int *block, const char *fn)
{
    const int rc = glow_geometry_base(L, D, g, fn);
    if (rc != TFK_OK) return rc;
    if (!glow_plan(g, L->slots, L->block, L->cg1, L->cg2, block))
        return fail(TFK_EINVAL, "%s: no launch shape fits (slots %d, block %d, channel groups %d / %d; %d B per slot)",
                    fn, L->slots, L->block, L->cg1, L->cg2, 4 * g.slot_floats);
    return TFK_OK;
}

}  // namespace

extern "C" {

int tfk_glow_plan(const tfk_glow_layer *layer, int32_t D, int32_t *slots, int32_t *block, int32_t *cg1, int32_t *cg2,
                  int32_t *lds_bytes, int32_t *tile_rows)
{
    GlowGeom g;
    int blk = 0;
    const int rc = glow_geometry(layer, D, g, &blk, "tfk_glow_plan");
    if (rc != TFK_OK) return rc;
    if (slots) *slots = g.slots;
    if (block) *block = blk;
    if (cg1) *cg1 = g.cg1;
    if (cg2) *cg2 = g.cg2;
    if (lds_bytes) *lds_bytes = 4 * (g.fixed_floats + g.slots * g.slot_floats);
    if (tile_rows) *tile_rows = g.tile_rows;
    return TFK_OK;
}

int64_t tfk_glow_weight_floats(int32_t c_in_taps) { return c_in_taps < 1 ? 0 : gw_total(c_in_taps); }

int tfk_glow_coupling(float *rows, float *logdet, int64_t N, int32_t D, const tfk_glow_layer *layer, int32_t inverse,
                      void *stream)
{
    const char *fn = "tfk_glow_coupling";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    GlowGeom g;
    int block = 0;
    const int rc = glow_geometry(layer, D, g, &block, fn);
    if (rc != TFK_OK) return rc;
    if (N == 0) return TFK_OK;
    if (!rows || !logdet || !layer->src_idx || !layer->src_st || !layer->tgt_idx || !layer->tgt_st || !layer->weights ||
        !layer->bg1 || !layer->bg2 || !layer->w_eff || !layer->b_eff)
        return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!aligned16(layer->w_eff) || !aligned16(layer->b_eff) || (reinterpret_cast<uintptr_t>(layer->tgt_st) & 7u) ||
        (reinterpret_cast<uintptr_t>(layer->src_st) & 7u))
        return fail(TFK_EINVAL, "%s: w_eff / b_eff need 16-byte, src_st / tgt_st 8-byte alignment", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = 4 * (size_t)(g.fixed_floats + g.slots * g.slot_floats);
    const int wgs_per_cu = imax(1, imin((int)(kGlowLdsBytes / lds), 2048 / block));
    const int64_t tiles = (N + g.tile_rows - 1) / g.tile_rows;
    int64_t grid = layer->grid > 0 ? layer->grid : (int64_t)cu_count() * wgs_per_cu;
    if (grid > tiles) grid = tiles;
    // (the attribute is per DEVICE: set on every call like the other launchers -- a process-wide "done" flag skipped it
    // on a second GPU, whose launch then failed for more than 64 KB of LDS)
#define TFK_GLOW(K, I)                                                                                               \
    do {                                                                                                             \
        auto kern = k_glow_coupling<K, I>;                                                                           \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                kGlowLdsBytes) != hipSuccess)                                                        \
            return fail(TFK_ELAUNCH, "%s: cannot raise the dynamic LDS limit: %s", fn,                               \
                        hipGetErrorString(hipGetLastError()));                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, s, rows, logdet, (long long)N, g,           \
                           layer->src_idx, reinterpret_cast<const float2 *>(layer->src_st), layer->tgt_idx,          \
                           reinterpret_cast<const float2 *>(layer->tgt_st), layer->weights, layer->bg1, layer->bg2,  \
                           reinterpret_cast<const float4 *>(layer->w_eff),                                           \
                           reinterpret_cast<const float4 *>(layer->b_eff));                                          \
    } while (0)
    if (layer->kind == 0 && !inverse) TFK_GLOW(0, false);
    else if (layer->kind == 0) TFK_GLOW(0, true);
    else if (layer->kind == 2 && !inverse) TFK_GLOW(2, false);
    else if (layer->kind == 2) TFK_GLOW(2, true);
    else if (!inverse) TFK_GLOW(1, false);
    else TFK_GLOW(1, true);
#undef TFK_GLOW
    return check_launch(fn);
}

int tfk_rows_fma(float *rows, const float *st, int64_t N, int32_t D, void *stream)
{
    const char *fn = "tfk_rows_fma";
    if (N < 0 || D < 1) return fail(TFK_EINVAL, "%s: N = %lld, D = %d", fn, (long long)N, D);
    if (N == 0) return TFK_OK;
    if (!rows || !st) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((D % 4) == 0 && aligned16(rows) && aligned16(st))
        hipLaunchKernelGGL(k_rows_fma, dim3(grid_for(N * (D / 4), kBlock)), dim3(kBlock), 0, s, rows,
                           reinterpret_cast<const float2 *>(st), (long long)N, D / 4);
    else
        hipLaunchKernelGGL(k_rows_fma1, dim3(grid_for(N * (int64_t)D, kBlock)), dim3(kBlock), 0, s, rows,
                           reinterpret_cast<const float2 *>(st), (long long)N, D);
    return check_launch(fn);
}

}  // extern "C"
