// tfk_glow_level.hip -- SEVERAL consecutive couplings of an image / multiscale flow per launch, rows held in the LDS.
//
// tfk_glow.hip runs ONE convolutional coupling (multiscale/base.py:19-114 + ConvNetConditioner, multiscale/conditioning/
// classic.py:8-145) per launch: it gathers the sources from HBM, and reads and writes the targets there -- for a
// checkerboard layer every 64-byte line of the row twice (rocprofv3: 1.98x the algorithmic bytes), for the 19 couplings
// of AffineGlow((3, 32, 32)) 240 KB per row.  But the couplings of one LEVEL of MultiscaleBijection.forward
// (multiscale/base.py:249-296: checkerboard layers, squeeze, channel-wise layers; then the same on half of the channels)
// all work on the same set of row elements.  Here a workgroup keeps those elements of G = 4 or 8 samples in the LDS
// (12 KB per sample at the first level) and walks the level's couplings over them: S0 gathers the sources from the LDS, the
// transform reads and writes its targets there, the row crosses HBM once per level in each direction.
//
// The stages are those of tfk_glow.hip (tfk_glow.h: same conv stage, same windows, same folded constants) with three
// differences:
//  * the sample-independent cells of the activation buffers (bias frame, bg1 / bg2 outside the computed windows) are
//    rewritten per coupling from a packed cell list -- the buffers are shared by all couplings of the level;
//  * the Linear layer runs as v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 blocks = 4 samples x 64 parameters per
//    instruction, so that FOUR resident samples fill the instruction (the 16-sample tiles of v_mfma_f32_16x16x4_f32 would
//    be three quarters padding); lane (b, j) receives the four parameters of group b -- (u, beta) of two neighbouring
//    targets, or four shifts -- for sample j and transforms them in the LDS row;
//  * the per-sample log-det of the whole level is accumulated in the LDS in layer order and added to HBM once.
#include <cstring>

#include "tfk_glow.h"

namespace tfk {

constexpr int kLevelMagic = 0x476c4c76;     // 'GlLv'
constexpr int kLevelFixed = 544;            // V [16][16] | ldpart [16 waves][8] | ldsum [8] | pad -> Hs (1x1 convolutions)

struct GlowStep {
    GlowGeom g;
    int inverse, n_bg;
    long long bg_off;                       // byte offset of the background cell list inside the blob
    long long wts_off;                      // ... and of a COPY of the layer's packed conv weights: read through the kernel's
                                            // const __restrict__ blob pointer they are scalar loads (weights behind a pointer that
                                            // was itself loaded from memory came in as per-lane global loads: conv stages 2-3x slower)
    const uint16_t *src_loc, *tgt_loc;
    const float2 *src_st, *tgt_st;
    const float *bg1, *bg2;
    const float4 *w4, *b4, *w_eff, *b_eff;
};

struct GlowLevelHeader {
    int magic, n_steps, G, block;
    int Dl, row_stride, slot_floats, fixed_floats;
    int lds_bytes, wgs_per_cu, steps_off, total_bytes;
};

// ---- S0: sources from the LDS rows through the first ConvModifier into the rectangle of A0 -------------------------------
// 1x1 kernel: a thread owns a pixel of the rectangle for all G samples (tables and weights are read once per pixel)
template <int G>
__device__ __forceinline__ void s0_level_one(const GlowGeom &g, const float *rowbuf, int row_stride, float *slot0,
                                             int slot_floats, const uint16_t *__restrict__ src_loc,
                                             const float2 *__restrict__ src_st, const float *__restrict__ wts)
{
    const int npix = g.hi * g.wi;
    const float *bm = wts + gw_bm(g.cm);
    const int plane = g.a0h * g.a0w;
    for (int pix = threadIdx.x; pix < npix; pix += blockDim.x) {
        const int iy = pix / g.wi, ix = pix - iy * g.wi;
        float o[G][4];
#pragma unroll
        for (int s = 0; s < G; ++s) o[s][0] = bm[0], o[s][1] = bm[1], o[s][2] = bm[2], o[s][3] = bm[3];
        for (int c0 = 0; c0 < g.c_in; c0 += 4) {
            int loc[4];
            float2 st[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u < g.c_in ? c0 + u : g.c_in - 1;
                loc[u] = src_loc[c * npix + pix];
                st[u] = src_st[c * npix + pix];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u;
                const bool on = c < g.c_in;
                const int w = on ? c : 0;
                const float w0 = wts[w], w1 = wts[g.cm + w], w2 = wts[2 * g.cm + w], w3 = wts[3 * g.cm + w];
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const float v = on ? fmaf(st[u].x, rowbuf[s * row_stride + loc[u]], st[u].y) : 0.0f;
                    o[s][0] = fmaf(w0, v, o[s][0]);
                    o[s][1] = fmaf(w1, v, o[s][1]);
                    o[s][2] = fmaf(w2, v, o[s][2]);
                    o[s][3] = fmaf(w3, v, o[s][3]);
                }
            }
        }
        float *a = slot0 + (g.oy + iy - g.a0y0) * g.a0w + (g.ox + ix - g.a0x0);
#pragma unroll
        for (int s = 0; s < G; ++s) {
            float *as = a + s * slot_floats;
            as[0] = o[s][0], as[plane] = o[s][1], as[2 * plane] = o[s][2], as[3 * plane] = o[s][3];
        }
    }
}

// modifier kernel 2 wide along an axis whose padding is odd (classic.py:26-33): a thread per (sample, rectangle pixel)
__device__ __forceinline__ void s0_level_taps(const GlowGeom &g, int G, const float *rowbuf, int row_stride, float *slot0,
                                              int slot_floats, const uint16_t *__restrict__ src_loc,
                                              const float2 *__restrict__ src_st, const float *__restrict__ wts)
{
    const int npix = g.hi * g.wi, rpix = g.rh * g.rw;
    const float *bm = wts + gw_bm(g.cm);
    const int taps = g.kh * g.kw;
    for (int t = threadIdx.x; t < G * rpix; t += blockDim.x) {
        const int slot = t / rpix, pix = t - slot * rpix;
        const int iy = pix / g.rw, ix = pix - iy * g.rw;
        const float *xr = rowbuf + slot * row_stride;
        float o0 = bm[0], o1 = bm[1], o2 = bm[2], o3 = bm[3];
        for (int tap = 0; tap < taps; ++tap) {
            const int ky = tap / g.kw, kx = tap - ky * g.kw;
            const int sy = iy - (g.kh - 1) + ky, sx = ix - (g.kw - 1) + kx;          // out[Y] = sum_k W[k] x[Y + k - pad]
            const bool inside = sy >= 0 && sy < g.hi && sx >= 0 && sx < g.wi;
            const int spix = inside ? sy * g.wi + sx : 0;
            for (int c = 0; c < g.c_in; ++c) {
                const int e = c * npix + spix;
                const float2 st = src_st[e];
                const float v = inside ? fmaf(st.x, xr[src_loc[e]], st.y) : 0.0f;
                const int w = c * taps + tap;
                o0 = fmaf(wts[w], v, o0);
                o1 = fmaf(wts[g.cm + w], v, o1);
                o2 = fmaf(wts[2 * g.cm + w], v, o2);
                o3 = fmaf(wts[3 * g.cm + w], v, o3);
            }
        }
        float *a = slot0 + slot * slot_floats + (g.oy + iy - g.a0y0) * g.a0w + (g.ox + ix - g.a0x0);
        const int plane = g.a0h * g.a0w;
        a[0] = o0, a[plane] = o1, a[2 * plane] = o2, a[3 * plane] = o3;
    }
}

// ---- Linear layer (v_mfma_f32_4x4x1_16B_f32) + bounded output + affine / shift transform in the LDS rows ----------------
// Block b of the instruction (lanes 4 b .. 4 b + 3) multiplies the column (W[64 t + 4 b + i][k])_i, held by lane 4 b + i,
// with the row (V[j][k])_j, held by lane 4 b + j: after the 16 k-steps lane (b, j) holds in its four accumulators
// h[j][64 t + 4 b + i], i = 0 .. 3 -- affine: (u, beta) of targets 32 t + 2 b and + 1; shift: the shifts of targets
// 64 t + 4 b + i -- of sample j of the quad.
template <int QUADS, bool SHIFT, bool INV>
__device__ __forceinline__ void transform_level(const GlowStep &st, const float *V, float *rowbuf, int row_stride,
                                                float *ldpart)
{
    constexpr int TPL = SHIFT ? 4 : 2;               // targets per lane and tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int b = lane >> 2, j = lane & 3;
    float bv[QUADS][16];
#pragma unroll
    for (int q = 0; q < QUADS; ++q)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 v = *reinterpret_cast<const float4 *>(V + (4 * q + j) * 16 + 4 * k4);
            bv[q][4 * k4] = v.x, bv[q][4 * k4 + 1] = v.y, bv[q][4 * k4 + 2] = v.z, bv[q][4 * k4 + 3] = v.w;
        }
    float ldr[QUADS];
#pragma unroll
    for (int q = 0; q < QUADS; ++q) ldr[q] = 0.0f;
    const int T = st.g.T;
    const int n_tiles = (T + 16 * TPL - 1) / (16 * TPL);
    for (int t = wave; t < n_tiles; t += nw) {
        const float4 *wp = st.w4 + (size_t)(64 * t + lane) * 4;
        const float4 a0 = wp[0], a1 = wp[1], a2 = wp[2], a3 = wp[3];
        const float4 bias = st.b4[16 * t + b];
        const int p0 = 16 * TPL * t + TPL * b;           // first target of this lane (tables padded to whole tiles)
        int loc[TPL];
        float2 pst[TPL];
        if constexpr (SHIFT) {
            const uint2 l2 = *reinterpret_cast<const uint2 *>(st.tgt_loc + p0);
            loc[0] = l2.x & 0xffff, loc[1] = l2.x >> 16, loc[2] = l2.y & 0xffff, loc[3] = l2.y >> 16;
        } else {
            const unsigned l1 = *reinterpret_cast<const unsigned *>(st.tgt_loc + p0);
            loc[0] = l1 & 0xffff, loc[1] = l1 >> 16;
        }
#pragma unroll
        for (int u = 0; u < TPL; ++u) pst[u] = st.tgt_st[p0 + u];
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            float *xr = rowbuf + (4 * q + j) * row_stride;
            float x[TPL];
#pragma unroll
            for (int u = 0; u < TPL; ++u) x[u] = xr[loc[u]];
            gf32x4 acc = {bias.x, bias.y, bias.z, bias.w};
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, bv[q][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, bv[q][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, bv[q][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, bv[q][3], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, bv[q][4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, bv[q][5], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, bv[q][6], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, bv[q][7], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.x, bv[q][8], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.y, bv[q][9], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.z, bv[q][10], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.w, bv[q][11], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3.x, bv[q][12], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3.y, bv[q][13], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3.z, bv[q][14], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3.w, bv[q][15], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < TPL; ++u) {
                const float v = fmaf(pst[u].x, x[u], pst[u].y);
                float out, wl = 0.0f;
                if (SHIFT) {
                    const float beta = bounded4(acc[u]);
                    out = INV ? v - beta : v + beta;
                } else {
                    const float u_ = bounded4(acc[2 * u]), beta = bounded4(acc[2 * u + 1]);
                    wl = fmaf(u_, 0.5f, kAffC0);                          // affine.py:33-34, log(alpha) up to 1e-10
                    const float alpha = __builtin_amdgcn_exp2f(wl * __int_as_float(0x3fb8aa3b)) + kAffMinScale;
                    out = INV ? (v - beta) * __builtin_amdgcn_rcpf(alpha) : alpha * v + beta;
                }
                if (p0 + u < T) {
                    xr[loc[u]] = out;
                    ldr[q] += INV ? -wl : wl;
                }
            }
        }
    }
    if (!SHIFT) {
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            float s = ldr[q];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) s += __shfl_xor(s, o, kWave);          // over the 16 groups b
            if (b == 0) ldpart[wave * 8 + 4 * q + j] = s;
        }
    }
}

// ---- invertible 1x1 convolution (LU factors per sample, matrix.py:11-99; linear/convolution.py:33-64) in the LDS rows ----
template <bool INV>
__device__ __forceinline__ void lu_level(const GlowStep &st, int G, const float *V, float *Hs, float *rowbuf, int row_stride,
                                         float *ldsum)
{
    const GlowGeom &g = st.g;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int j = lane & 15, q = lane >> 4;
    float bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[ks] = V[j * 16 + 4 * ks + q];            // (rows >= G of V are zero)
    for (int tl = wave; tl < g.n_tiles; tl += nw) {
        const float4 a = st.w_eff[tl * 64 + lane];
        const float4 bb = st.b_eff[tl * 4 + q];
        gf32x4 acc = {bb.x, bb.y, bb.z, bb.w};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[3], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 16 * tl + 4 * q + r;
            float hv = bounded4(acc[r]);
            hv = e < g.n_ch ? expf(hv) / 10.0f + 1.0f : hv / 10.0f;            // matrix.py:31-36
            Hs[j * g.h_stride + e] = hv;
        }
    }
    __syncthreads();
    const int n = g.n_ch, HW = g.hw;
    const int n_off = n * (n - 1) / 2;
    for (int t = tid; t < G * HW; t += nthr) {
        const int s = t / HW, p = t - s * HW;
        float *xr = rowbuf + s * row_stride;
        const float *hr = Hs + s * g.h_stride;
        float v[kGlowMaxCh];
        int pos[kGlowMaxCh];
#pragma unroll
        for (int c = 0; c < kGlowMaxCh; ++c)
            if (c < n) {
                pos[c] = st.tgt_loc[c * HW + p];
                const float2 m = st.tgt_st[c * HW + p];
                v[c] = fmaf(m.x, xr[pos[c]], m.y);
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
            ldsum[s] += INV ? -s_ld : s_ld;
        }
    }
}

template <int QUADS>
__global__ __launch_bounds__(1024) void k_glow_level(const float *__restrict__ rows_in, float *rows_out, float *logdet,
                                                     long long N, int D, const int *__restrict__ row_idx,
                                                     const unsigned char *__restrict__ blob)
{
    constexpr int G = 4 * QUADS;
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    const GlowLevelHeader &H = *reinterpret_cast<const GlowLevelHeader *>(blob);
    const GlowStep *steps = reinterpret_cast<const GlowStep *>(blob + H.steps_off);
    float *V = lds;                                   // [16][16]; rows >= G stay zero
    float *ldpart = lds + 256;                        // [16 waves][8]
    float *ldsum = lds + 384;                         // [8]
    float *Hs = lds + kLevelFixed;                    // [16][h_stride] (levels with a 1x1 convolution)
    const int row_stride = H.row_stride, slot_floats = H.slot_floats, Dl = H.Dl;
    float *rowbuf = lds + H.fixed_floats;
    float *slot0 = rowbuf + G * row_stride;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nw = nthr >> 6;
    const bool ident = row_idx == nullptr;

    for (int i = tid; i < 256; i += nthr) V[i] = 0.0f;

    const long long n_tiles = (N + G - 1) / G;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row_base = tile * G;
        // ---- the level's elements of G samples into the LDS (rows past N: zeros, never written back) ----
        if (ident && (Dl & 3) == 0) {
            const int d4 = Dl >> 2;
#pragma unroll
            for (int s = 0; s < G; ++s) {
                const bool ok = row_base + s < N;
                const float4 *src = reinterpret_cast<const float4 *>(rows_in + (row_base + (ok ? s : 0)) * D);
                float *dst = rowbuf + s * row_stride;
                for (int i = tid; i < d4; i += nthr) {
                    const float4 v = ok ? nt_load4(src + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    dst[4 * i] = v.x, dst[4 * i + 1] = v.y, dst[4 * i + 2] = v.z, dst[4 * i + 3] = v.w;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < G; ++s) {
                const bool ok = row_base + s < N;
                const float *src = rows_in + (row_base + (ok ? s : 0)) * D;
                float *dst = rowbuf + s * row_stride;
                for (int i = tid; i < Dl; i += nthr) dst[i] = ok ? src[ident ? i : row_idx[i]] : 0.0f;
            }
        }
        if (tid < G) ldsum[tid] = row_base + tid < N ? logdet[row_base + tid] : 0.0f;
        __syncthreads();

        for (int si = 0; si < H.n_steps; ++si) {
            const GlowStep &st = steps[si];
            const GlowGeom &g = st.g;
            const float *wts = reinterpret_cast<const float *>(blob + st.wts_off);
            // ---- cells of the activation buffers that no sample changes (the buffers serve every coupling of the level) ----
            if (!(g.skip & 8)) {
                const unsigned *cells = reinterpret_cast<const unsigned *>(blob + st.bg_off);
                const float *bm = wts + gw_bm(g.cm);
                for (int i = tid; i < st.n_bg; i += nthr) {
                    const unsigned e = cells[i];
                    const int off = e & 0xffff, kind = (e >> 16) & 3, idx = e >> 18;
                    const float v = kind == 0 ? 0.0f : (kind == 1 ? bm[idx] : (kind == 2 ? st.bg1[idx] : st.bg2[idx]));
#pragma unroll
                    for (int s = 0; s < G; ++s) slot0[s * slot_floats + off] = v;
                }
            }
            // ---- S0: pending map + first ConvModifier into the rectangle (disjoint from the cells above) ----
            if (!(g.skip & 1)) {
                if (g.kh * g.kw == 1) s0_level_one<G>(g, rowbuf, row_stride, slot0, slot_floats, st.src_loc, st.src_st, wts);
                else s0_level_taps(g, G, rowbuf, row_stride, slot0, slot_floats, st.src_loc, st.src_st, wts);
            }
            __syncthreads();
            const int ci = g.cm;
            if (!(g.skip & 2)) {
                conv_stage_cg<4, 8, true>(g.cg1, slot0, slot_floats, 0, g.a0h, g.a0w, g.off_p1, g.b1h, g.b1w,
                                          g.p1y0 - g.b1y0, g.p1x0 - g.b1x0, g.p1h, g.p1w, G, wts + gw_w1(ci),
                                          wts + gw_b1(ci), wts + gw_b1(ci) + 8, wts + gw_b1(ci) + 16);
                __syncthreads();
                conv_stage_cg<8, 8, true>(g.cg2, slot0, slot_floats, g.off_p1, g.b1h, g.b1w, g.off_p2, 10, 10,
                                          g.p2y0 + 1, g.p2x0 + 1, g.p2h, g.p2w, G, wts + gw_w2(ci), wts + gw_b2(ci),
                                          wts + gw_b2(ci) + 8, wts + gw_b2(ci) + 16);
                __syncthreads();
                conv_stage<8, 4, 2, false>(slot0, slot_floats, g.off_p2, 10, 10, g.off_p3, 4, 4, 0, 0, 4, 4, G,
                                           wts + gw_w3(ci), wts + gw_b3(ci), nullptr, nullptr);
                __syncthreads();
            }
            // ---- S4: BatchNorm 3 + second ConvModifier (4 -> 1 channel), folded on the host ----
            if (!(g.skip & 16))
            for (int t = tid; t < G * 16; t += nthr) {
                const int slot = t >> 4, p = t & 15;
                const float *p3 = slot0 + slot * slot_floats + g.off_p3 + p;
                const float *m2 = wts + gw_m2(ci);
                float v = m2[4];
                v = fmaf(m2[0], p3[0], v);
                v = fmaf(m2[1], p3[16], v);
                v = fmaf(m2[2], p3[32], v);
                v = fmaf(m2[3], p3[48], v);
                V[slot * 16 + p] = v;
            }
            __syncthreads();
            // ---- Linear layer + bounded output + transform, in the LDS rows ----
            if (g.skip & 4) {
            } else if (g.kind == 1) {
                if (st.inverse) lu_level<true>(st, G, V, Hs, rowbuf, row_stride, ldsum);
                else lu_level<false>(st, G, V, Hs, rowbuf, row_stride, ldsum);
                __syncthreads();
            } else if (g.kind == 2) {
                if (st.inverse) transform_level<QUADS, true, true>(st, V, rowbuf, row_stride, ldpart);
                else transform_level<QUADS, true, false>(st, V, rowbuf, row_stride, ldpart);
                __syncthreads();
            } else {
                if (st.inverse) transform_level<QUADS, false, true>(st, V, rowbuf, row_stride, ldpart);
                else transform_level<QUADS, false, false>(st, V, rowbuf, row_stride, ldpart);
                __syncthreads();
                if (tid < G) {
                    float s = 0.0f;
                    for (int w = 0; w < nw; ++w) s += ldpart[w * 8 + tid];
                    ldsum[tid] += s;
                }
                // (ldpart is rewritten only behind several more barriers; ldsum is next touched by these threads)
            }
        }

        // ---- back to HBM ----
        if (ident && (Dl & 3) == 0) {
            const int d4 = Dl >> 2;
#pragma unroll
            for (int s = 0; s < G; ++s) {
                if (row_base + s >= N) break;
                float4 *dst = reinterpret_cast<float4 *>(rows_out + (row_base + s) * D);
                const float *src = rowbuf + s * row_stride;
                for (int i = tid; i < d4; i += nthr)
                    nt_store4(dst + i, make_float4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]));
            }
        } else {
#pragma unroll
            for (int s = 0; s < G; ++s) {
                if (row_base + s >= N) break;
                float *dst = rows_out + (row_base + s) * D;
                const float *src = rowbuf + s * row_stride;
                for (int i = tid; i < Dl; i += nthr) dst[ident ? i : row_idx[i]] = src[i];
            }
        }
        if (tid < G && row_base + tid < N) logdet[row_base + tid] = ldsum[tid];
        __syncthreads();                                        // the row buffer is free again
    }
}

}  // namespace tfk

using namespace tfk;

namespace {

struct LevelPlan {
    GlowLevelHeader H;
    GlowGeom geom[64];
    int n_bg[64];
    bool has_lu;
};

// cells of the three activation buffers that the stages themselves never write: (offset in the slot) | kind << 16 | index << 18,
// kind 0 = zero (outside the frame), 1 = first modifier's bias [channel], 2 = bg1 [(ch 16 + y) 16 + x], 3 = bg2 [(ch 8 + y) 8 + x]
int bg_cells(const GlowGeom &g, uint32_t *out)
{
    int n = 0;
    auto put = [&](int off, int kind, int idx) {
        if (out) out[n] = (uint32_t)off | ((uint32_t)kind << 16) | ((uint32_t)idx << 18);
        ++n;
    };
    for (int ch = 0; ch < 4; ++ch)
        for (int r = 0; r < g.a0h; ++r)
            for (int c = 0; c < g.a0w; ++c) {
                const int y = g.a0y0 + r, x = g.a0x0 + c;
                if (y >= g.oy && y < g.oy + g.rh && x >= g.ox && x < g.ox + g.rw) continue;      // S0 writes the rectangle
                const bool in = y >= 0 && y < kGlowFrame && x >= 0 && x < kGlowFrame;
                put((ch * g.a0h + r) * g.a0w + c, in ? 1 : 0, in ? ch : 0);
            }
    for (int ch = 0; ch < 8; ++ch)
        for (int r = 0; r < g.b1h; ++r)
            for (int c = 0; c < g.b1w; ++c) {
                const int y = g.b1y0 + r, x = g.b1x0 + c;
                if (y >= g.p1y0 && y < g.p1y0 + g.p1h && x >= g.p1x0 && x < g.p1x0 + g.p1w) continue;   // conv block 1 writes its window
                const bool in = y >= 0 && y < 16 && x >= 0 && x < 16;
                put(g.off_p1 + (ch * g.b1h + r) * g.b1w + c, in ? 2 : 0, in ? (ch * 16 + y) * 16 + x : 0);
            }
    for (int ch = 0; ch < 8; ++ch)
        for (int r = 0; r < 10; ++r)
            for (int c = 0; c < 10; ++c) {
                const int y = r - 1, x = c - 1;
                if (y >= g.p2y0 && y < g.p2y0 + g.p2h && x >= g.p2x0 && x < g.p2x0 + g.p2w) continue;
                const bool in = y >= 0 && y < 8 && x >= 0 && x < 8;
                put(g.off_p2 + ch * 100 + r * 10 + c, in ? 3 : 0, in ? (ch * 8 + y) * 8 + x : 0);
            }
    return n;
}

int level_plan(const tfk_glow_level_step *steps, int32_t n_steps, int32_t D, int32_t Dl, int32_t samples, int32_t block,
               LevelPlan &P, const char *fn)
{
    if (!steps || n_steps < 1 || n_steps > 64) return fail(TFK_EINVAL, "%s: %d steps (1 .. 64)", fn, n_steps);
    if (D < 1 || Dl < 1 || Dl > D || Dl > 65535) return fail(TFK_EINVAL, "%s: D = %d, D_level = %d (<= D, <= 65535)", fn, D, Dl);
    if (samples != 0 && samples != 4 && samples != 8) return fail(TFK_EINVAL, "%s: samples = %d (0, 4 or 8)", fn, samples);
    if (block != 0 && block != 256 && block != 512 && block != 1024) return fail(TFK_EINVAL, "%s: block = %d", fn, block);
    int slot_floats = 0, hs_floats = 0;
    P.has_lu = false;
    int64_t bytes = sizeof(GlowLevelHeader);
    bytes = (bytes + 15) & ~15ll;
    const int64_t steps_off = bytes;
    bytes += (int64_t)n_steps * sizeof(GlowStep);
    for (int i = 0; i < n_steps; ++i) {
        GlowGeom &g = P.geom[i];
        const int rc = glow_geometry_base(&steps[i].layer, D, g, fn);
        if (rc != TFK_OK) return rc;
        if (g.slot_floats > 65535) return fail(TFK_EINVAL, "%s: activation buffers of %d floats per sample", fn, g.slot_floats);
        const int cg1 = steps[i].layer.cg1 ? steps[i].layer.cg1 : 4, cg2 = steps[i].layer.cg2 ? steps[i].layer.cg2 : 4;
        if ((cg1 != 8 && cg1 != 4 && cg1 != 2) || (cg2 != 8 && cg2 != 4 && cg2 != 2))
            return fail(TFK_EINVAL, "%s: channel groups %d / %d", fn, cg1, cg2);
        g.cg1 = cg1, g.cg2 = cg2;
        if (g.slot_floats > slot_floats) slot_floats = g.slot_floats;
        if (g.kind == 1) {
            P.has_lu = true;
            const int h = (kGlowMaxRows * g.h_stride + 3) & ~3;
            if (h > hs_floats) hs_floats = h;
        }
        P.n_bg[i] = bg_cells(g, nullptr);
        bytes = (bytes + 15) & ~15ll;
        bytes += 4ll * P.n_bg[i];
        bytes = (bytes + 15) & ~15ll;
        bytes += 4ll * gw_total(g.cm);
    }
    bytes = (bytes + 15) & ~15ll;
    GlowLevelHeader &H = P.H;
    H = GlowLevelHeader{};
    H.magic = kLevelMagic, H.n_steps = n_steps, H.Dl = Dl;
    H.row_stride = Dl | 1;                          // odd: the four samples of a quad land on different banks
    H.slot_floats = (slot_floats + 3) & ~3;
    H.fixed_floats = kLevelFixed + hs_floats;
    auto lds_for = [&](int G) { return 4 * (H.fixed_floats + G * (H.row_stride + H.slot_floats) + 4); };
    int G = samples;
    if (!G) {
        // as many workgroups per CU as the LDS holds with four resident samples each; eight samples where even one
        // workgroup of four leaves more than half of the LDS unused is not worth its registers (QUADS = 2): four
        G = 4;
    }
    if (lds_for(G) > kGlowLdsBytes - 1024)
        return fail(TFK_EINVAL, "%s: %d samples x (%d row + %d activation floats) do not fit the LDS", fn, G, H.row_stride,
                    H.slot_floats);
    H.G = G;
    H.lds_bytes = lds_for(G);
    int per_cu = kGlowLdsBytes / H.lds_bytes;
    if (per_cu < 1) per_cu = 1;
    int blk = block;
    if (!blk) blk = per_cu >= 4 ? 256 : (per_cu >= 2 ? 512 : 1024);
    if (per_cu * blk > 2048) per_cu = 2048 / blk;
    H.block = blk, H.wgs_per_cu = per_cu;
    H.steps_off = (int)steps_off, H.total_bytes = (int)bytes;
    return TFK_OK;
}

}  // namespace

extern "C" {

int64_t tfk_glow_level_blob_bytes(const tfk_glow_level_step *steps, int32_t n_steps, int32_t D, int32_t D_level,
                                  int32_t samples, int32_t block)
{
    LevelPlan P;
    const int rc = level_plan(steps, n_steps, D, D_level, samples, block, P, "tfk_glow_level_blob_bytes");
    return rc == TFK_OK ? (int64_t)P.H.total_bytes : -(int64_t)rc;
}

int tfk_glow_level_pack(const tfk_glow_level_step *steps, int32_t n_steps, int32_t D, int32_t D_level, int32_t samples,
                        int32_t block, void *blob_host, int64_t blob_bytes)
{
    const char *fn = "tfk_glow_level_pack";
    LevelPlan P;
    const int rc = level_plan(steps, n_steps, D, D_level, samples, block, P, fn);
    if (rc != TFK_OK) return rc;
    if (!blob_host || blob_bytes < P.H.total_bytes)
        return fail(TFK_EINVAL, "%s: blob of %lld bytes, need %d", fn, (long long)blob_bytes, P.H.total_bytes);
    unsigned char *base = static_cast<unsigned char *>(blob_host);
    memset(base, 0, (size_t)P.H.total_bytes);
    memcpy(base, &P.H, sizeof(P.H));
    int64_t off = P.H.steps_off + (int64_t)n_steps * sizeof(GlowStep);
    for (int i = 0; i < n_steps; ++i) {
        const tfk_glow_level_step &S = steps[i];
        const tfk_glow_layer &L = S.layer;
        if (!S.src_loc || !S.tgt_loc || !L.src_st || !L.tgt_st || !S.weights_host || !L.bg1 || !L.bg2)
            return fail(TFK_EINVAL, "%s: step %d: null pointer", fn, i);
        if (L.kind == 1 ? (!L.w_eff || !L.b_eff) : (!S.w4 || !S.b4))
            return fail(TFK_EINVAL, "%s: step %d: the Linear layer's operands are missing", fn, i);
        if ((L.kind != 1 && (!aligned16(S.w4) || !aligned16(S.b4))) || (L.kind == 1 && (!aligned16(L.w_eff) || !aligned16(L.b_eff))) ||
            (reinterpret_cast<uintptr_t>(L.tgt_st) & 7u) || (reinterpret_cast<uintptr_t>(L.src_st) & 7u) ||
            (reinterpret_cast<uintptr_t>(S.tgt_loc) & 7u))
            return fail(TFK_EINVAL, "%s: step %d: w4 / b4 / w_eff / b_eff need 16-byte, src_st / tgt_st / tgt_loc 8-byte alignment", fn, i);
        GlowStep st{};
        st.g = P.geom[i];
        st.g.slots = P.H.G, st.g.tile_rows = P.H.G;
        st.inverse = S.inverse ? 1 : 0;
        st.n_bg = P.n_bg[i];
        off = (off + 15) & ~15ll;
        st.bg_off = off;
        bg_cells(P.geom[i], reinterpret_cast<uint32_t *>(base + off));
        off += 4ll * P.n_bg[i];
        off = (off + 15) & ~15ll;
        st.wts_off = off;
        memcpy(base + off, S.weights_host, 4u * (size_t)gw_total(P.geom[i].cm));
        off += 4ll * gw_total(P.geom[i].cm);
        st.src_loc = S.src_loc, st.tgt_loc = S.tgt_loc;
        st.src_st = reinterpret_cast<const float2 *>(L.src_st), st.tgt_st = reinterpret_cast<const float2 *>(L.tgt_st);
        st.bg1 = L.bg1, st.bg2 = L.bg2;
        st.w4 = reinterpret_cast<const float4 *>(S.w4), st.b4 = reinterpret_cast<const float4 *>(S.b4);
        st.w_eff = reinterpret_cast<const float4 *>(L.w_eff), st.b_eff = reinterpret_cast<const float4 *>(L.b_eff);
        memcpy(base + P.H.steps_off + (int64_t)i * sizeof(GlowStep), &st, sizeof(st));
    }
    return TFK_OK;
}

int tfk_glow_level_info(const void *blob_host, int32_t *samples, int32_t *block, int32_t *lds_bytes, int32_t *wgs_per_cu)
{
    const GlowLevelHeader *H = static_cast<const GlowLevelHeader *>(blob_host);
    if (!H || H->magic != kLevelMagic) return fail(TFK_EINVAL, "tfk_glow_level_info: not a level blob");
    if (samples) *samples = H->G;
    if (block) *block = H->block;
    if (lds_bytes) *lds_bytes = H->lds_bytes;
    if (wgs_per_cu) *wgs_per_cu = H->wgs_per_cu;
    return TFK_OK;
}

int tfk_glow_level(const float *rows_in, float *rows_out, float *logdet, int64_t N, int32_t D, const int32_t *row_idx,
                   const void *blob_host, const void *blob_dev, void *stream)
{
    const char *fn = "tfk_glow_level";
    const GlowLevelHeader *H = static_cast<const GlowLevelHeader *>(blob_host);
    if (!H || H->magic != kLevelMagic) return fail(TFK_EINVAL, "%s: not a level blob", fn);
    if (N < 0 || D < H->Dl) return fail(TFK_EINVAL, "%s: N = %lld, D = %d, D_level = %d", fn, (long long)N, D, H->Dl);
    if (N == 0) return TFK_OK;
    if (!rows_in || !rows_out || !logdet || !blob_dev) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (row_idx && rows_in != rows_out) return fail(TFK_EINVAL, "%s: a level on a subset of the row works in place", fn);
    if (!row_idx && H->Dl != D) return fail(TFK_EINVAL, "%s: row_idx = NULL takes D_level = D (%d != %d)", fn, H->Dl, D);
    if (!row_idx && (D & 3) == 0 && (!aligned16(rows_in) || !aligned16(rows_out)))
        return fail(TFK_EINVAL, "%s: rows need 16-byte alignment", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t tiles = (N + H->G - 1) / H->G;
    int64_t grid = (int64_t)cu_count() * H->wgs_per_cu;
    if (grid > tiles) grid = tiles;
    const unsigned char *bd = static_cast<const unsigned char *>(blob_dev);
#define TFK_LEVEL(Q)                                                                                                     \
    do {                                                                                                                 \
        auto kern = k_glow_level<Q>;                                                                                     \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                kGlowLdsBytes) != hipSuccess)                                                            \
            return fail(TFK_ELAUNCH, "%s: cannot raise the dynamic LDS limit: %s", fn, hipGetErrorString(hipGetLastError())); \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(H->block), (size_t)H->lds_bytes, s, rows_in, rows_out, logdet, \
                           (long long)N, D, row_idx, bd);                                                                \
    } while (0)
    if (H->G == 4) TFK_LEVEL(1);
    else TFK_LEVEL(2);
#undef TFK_LEVEL
    return check_launch(fn);
}

}  // extern "C"
