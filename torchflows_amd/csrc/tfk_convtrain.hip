// tfk_convtrain.hip -- the Glow ConvNet conditioner in TRAINING: forward with batch statistics and reverse mode
// (reference multiscale/conditioning/classic.py:45-122: ConvModifier -> 3 x [conv3x3 -> ReLU -> MaxPool2d(2) ->
// BatchNorm2d] -> ConvModifier -> Linear; `Flow.fit` runs it with BatchNorm in training mode, flows.py:333).
//
// The library route is ~70 launches per coupling and step (MIOpen Winograd + layout transposes, ReLU, pooling, three
// BatchNorm kernels per block and their backward counterparts, a 958 us weight-gradient GEMM for the Linear layer whose
// contraction runs over the batch): 22 ms per step of MultiscaleRealNVP((1, 28, 28)) on 1 000 images, all of it tiny
// kernels.  Here a block is ONE forward launch and ONE reverse-mode launch:
//   forward   conv3x3 (+ the previous block's BatchNorm applied on load) -> ReLU -> max-pool, pooled pre-normalisation
//             output + the arg-max of every 2x2 window (one byte) written once; per-channel sum / sum of squares handed in
//             per workgroup, and the LAST workgroup to finish (ticket counter) adds them in index order, in fp64, and
//             leaves the BatchNorm's scale / shift / mean / 1/std (and the running statistics' update) for the next launch
//   backward  BatchNorm backward as dy = c1 g + c2 y + c3 per channel (c1..c3 from the batch sums the PRODUCER of g handed
//             in), routed to the arg-max where the pooled output is positive, then the convolution's three gradients from
//             LDS: dW (thread per (c_out, c_in) pair x 9 taps, sparse in the pooled cells), db, dX (dense gather), and
//             the batch sums the previous BatchNorm's backward needs -- again finished by the last workgroup.
// Every sum over the batch is a fixed-order sum of per-workgroup partials: deterministic for a given grid.
// Batch-coupled quantities (BatchNorm statistics) are why this is a launch per BLOCK and not per coupling: each block's
// output depends on every sample of the batch through the normalisation that precedes the next block.
#include "tfk_common.h"

namespace tfk {

constexpr int kCtMaxGrid = 1024;          // workgroups that hand in partial sums (rows of the workspace)
constexpr int kCtMaxK = 640;              // floats per workgroup (block backward 8 -> 8: 576 + 8 + 16)
constexpr int kCtHeaderBytes = 1024;      // ticket counters: 1 + kCtMaxGrid / kCtCluster of them

// Every workgroup hands in K floats (vals, LDS); true in the last workgroup to finish, with tot[k] = the sum over all
// workgroups (fp64) in a FIXED order.  Two levels, because one workgroup walking a thousand rows of partial sums costs
// more than the convolution it follows (an agent-scope load is ~0.6 us, measured: 37 us for 256 rows x 312 columns):
// workgroups form clusters of kCtCluster; the last one of a cluster to finish adds the cluster's rows in index order and
// hands in ONE row for the cluster; the last cluster to do so adds the cluster rows in index order.  The result does not
// depend on which workgroup came last at either level.
// Visibility across the XCDs' L2s as in finish_sum_f64 (tfk_common.h): agent-scope atomic stores, awaited before the
// agent-scope ticket; the adding workgroup reads with agent-scope atomic loads.  counters: [0] = top level, [1 + c] =
// cluster c; all left at zero.  partial: gridDim.x rows, then ceil(gridDim.x / kCtCluster) cluster rows.
constexpr int kCtCluster = 16;

template <int BLOCK>
__device__ __forceinline__ void ct_store_row(float *row, const float *vals, int K)
{
    for (int k = threadIdx.x; k < K; k += BLOCK)
        __hip_atomic_store(row + k, vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// tot[k] = rows[0][k] + rows[1][k] + ... (R <= 64 rows of K floats), every lane a column, 16 loads in flight
template <int BLOCK>
__device__ __forceinline__ void ct_add_rows(const float *rows, int R, int K, double *tot)
{
    for (int k = threadIdx.x; k < K; k += BLOCK) {
        double a = 0.0;
        int r = 0;
        for (; r + 16 <= R; r += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = __hip_atomic_load(rows + (size_t)(r + j) * K + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < 16; ++j) a += (double)v[j];
        }
        for (; r < R; ++r)
            a += (double)__hip_atomic_load(rows + (size_t)r * K + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[k] = a;
    }
    __syncthreads();
}

template <int BLOCK>
__device__ __forceinline__ bool hand_in(float *vals, int K, float *partial, unsigned long long *counters,
                                        double *tot, int *flag)
{
    __syncthreads();
    const unsigned G = gridDim.x, NC = (G + kCtCluster - 1) / kCtCluster;
    const unsigned cluster = blockIdx.x / kCtCluster;
    const unsigned first = cluster * kCtCluster, members = (first + kCtCluster <= G) ? kCtCluster : G - first;
    ct_store_row<BLOCK>(partial + (size_t)blockIdx.x * K, vals, K);
    if (threadIdx.x == 0) {
        const unsigned long long ticket =
            __hip_atomic_fetch_add(counters + 1 + cluster, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (ticket == (unsigned long long)members - 1ull) ? 1 : 0;
    }
    __syncthreads();
    if (!*flag) return false;
    // last of its cluster
    if (threadIdx.x == 0) __hip_atomic_store(counters + 1 + cluster, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ct_add_rows<BLOCK>(partial + (size_t)first * K, (int)members, K, tot);
    if (NC == 1) return true;
    for (int k = threadIdx.x; k < K; k += BLOCK) vals[k] = (float)tot[k];
    __syncthreads();
    float *crows = partial + (size_t)G * K;
    ct_store_row<BLOCK>(crows + (size_t)cluster * K, vals, K);
    if (threadIdx.x == 0) {
        const unsigned long long ticket =
            __hip_atomic_fetch_add(counters, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (ticket == (unsigned long long)NC - 1ull) ? 1 : 0;
    }
    __syncthreads();
    if (!*flag) return false;
    if (threadIdx.x == 0) __hip_atomic_store(counters, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ct_add_rows<BLOCK>(crows, (int)NC, K, tot);
    return true;
}

// sum of v over the workgroup's lanes -> dst[k] (LDS), K values; wred: 4 x K floats of LDS
template <int K>
__device__ __forceinline__ void block_sums(const float (&v)[K], float *wred, float *dst)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
        if (lane == 0) wred[wave * K + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K)
        dst[threadIdx.x] = (wred[threadIdx.x] + wred[K + threadIdx.x]) + (wred[2 * K + threadIdx.x] + wred[3 * K + threadIdx.x]);
    __syncthreads();
}

struct BnFwd {              // the BatchNorm2d behind a block (functional.py batch_norm semantics)
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long *num_batches;
    float *stats;           // out: scale | shift | mean | 1/std, 4 * C floats
    float eps, momentum;
    int training, update;
};

struct BnBwd {              // the BatchNorm2d whose OUTPUT gradient a backward launch produces
    const float *stats;     // its forward's scale | shift | mean | 1/std (null: none)
    float *coef;            // out: c1 | c2 | c3 with d(loss)/d(input) = c1 g + c2 y + c3
    float *dgamma, *dbeta;  // out
    double count;           // elements per channel in the batch
    int training;
};

// sg[c] = sum g, sgy[c] = sum g * y (y = the BatchNorm's input), over the batch
__device__ __forceinline__ void bn_backward_finish(const double *sg, const double *sgy, int C, const BnBwd &bn)
{
    if (!bn.stats) return;
    for (int c = threadIdx.x; c < C; c += kBlock) {
        const double scale = bn.stats[c], mean = bn.stats[2 * C + c], invstd = bn.stats[3 * C + c];
        const double dbeta = sg[c], dgamma = invstd * (sgy[c] - mean * sg[c]);
        bn.dbeta[c] = (float)dbeta;
        bn.dgamma[c] = (float)dgamma;
        double c2 = 0.0, c3 = 0.0;
        if (bn.training) {      // dy = scale (g - mean(g) - xhat mean(g xhat)),  xhat = (y - mean) / std
            const double mg = dbeta / bn.count, mgx = dgamma / bn.count;
            c2 = -scale * invstd * mgx;
            c3 = -scale * mg + scale * invstd * mgx * mean;
        }
        bn.coef[c] = (float)scale;
        bn.coef[C + c] = (float)c2;
        bn.coef[2 * C + c] = (float)c3;
    }
}

// ---- forward of one block ---------------------------------------------------------------------------------------
// one lane per pooled pixel, all output channels (as k_conv3x3_relu_pool_affine, tfk_convblock.hip); in_aff: the previous
// BatchNorm's scale | shift applied to the input inside the image (the zero padding stays zero), or null.
template <int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void k_ct_block_fwd(
    const float *__restrict__ x, const float *__restrict__ in_aff, const float *__restrict__ w,
    const float *__restrict__ bias, float *__restrict__ y, unsigned char *__restrict__ amax, float *partial,
    unsigned long long *counter, BnFwd bn, long long N, int H, int W)
{
    __shared__ __attribute__((aligned(16))) float ws[CIN * COUT * 12];
    __shared__ float bs[COUT], ia[2 * CIN], vals[2 * COUT], wred[4 * 2 * COUT];
    __shared__ double tot[2 * COUT];
    __shared__ int flag;
    for (int i = threadIdx.x; i < CIN * COUT * 12; i += kBlock) {
        const int tap = i % 12, co = (i / 12) % COUT, ci = i / (12 * COUT);
        ws[i] = tap < 9 ? w[(co * CIN + ci) * 9 + tap] : 0.0f;
    }
    for (int i = threadIdx.x; i < COUT; i += kBlock) bs[i] = bias[i];
    for (int i = threadIdx.x; i < CIN; i += kBlock) {
        ia[i] = in_aff ? in_aff[i] : 1.0f;
        ia[CIN + i] = in_aff ? in_aff[CIN + i] : 0.0f;
    }
    __syncthreads();
    const int PH = H >> 1, PW = W >> 1;
    const long long total = N * (long long)PH * PW;
    const long long plane = (long long)H * W;
    float s[2 * COUT];
#pragma unroll
    for (int k = 0; k < 2 * COUT; ++k) s[k] = 0.0f;
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kBlock) {
        const int pw = (int)(idx % PW);
        const int ph = (int)((idx / PW) % PH);
        const long long n = idx / ((long long)PW * PH);
        float acc[COUT][2][2];
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) acc[co][oy][ox] = bs[co];
        const int ih0 = 2 * ph - 1, iw0 = 2 * pw - 1;
#pragma unroll 1
        for (int ci = 0; ci < CIN; ++ci) {
            const float *src = x + (n * CIN + ci) * plane;
            const float sc = ia[ci], sh = ia[CIN + ci];
            float p[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ih = ih0 + r;
                const bool rok = (ih >= 0) && (ih < H);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int iw = iw0 + c;
                    p[r][c] = (rok && iw >= 0 && iw < W) ? fmaf(src[(long long)ih * W + iw], sc, sh) : 0.0f;
                }
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float4 w0 = *reinterpret_cast<const float4 *>(ws + (ci * COUT + co) * 12);
                const float4 w1 = *reinterpret_cast<const float4 *>(ws + (ci * COUT + co) * 12 + 4);
                const float4 w2 = *reinterpret_cast<const float4 *>(ws + (ci * COUT + co) * 12 + 8);
                const float k[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x};
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        float a = acc[co][oy][ox];
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a = fmaf(k[3 * ky + kx], p[oy + ky][ox + kx], a);
                        acc[co][oy][ox] = a;
                    }
            }
        }
        const long long o0 = (n * COUT) * (long long)PH * PW + (long long)ph * PW + pw;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float m = acc[co][0][0];
            int a = 0;
            if (acc[co][0][1] > m) { m = acc[co][0][1]; a = 1; }
            if (acc[co][1][0] > m) { m = acc[co][1][0]; a = 2; }
            if (acc[co][1][1] > m) { m = acc[co][1][1]; a = 3; }
            const float v = fmaxf(m, 0.0f);                      // relu and max commute
            y[o0 + (long long)co * PH * PW] = v;
            amax[o0 + (long long)co * PH * PW] = (unsigned char)a;
            s[co] += v;
            s[COUT + co] = fmaf(v, v, s[COUT + co]);
        }
    }
    if (!bn.training) {             // inference statistics: nothing to add up
        if (blockIdx.x == 0 && threadIdx.x < COUT) {
            const int c = threadIdx.x;
            const float invstd = 1.0f / sqrtf(bn.running_var[c] + bn.eps);
            const float scale = bn.gamma[c] * invstd;
            bn.stats[c] = scale;
            bn.stats[COUT + c] = bn.beta[c] - bn.running_mean[c] * scale;
            bn.stats[2 * COUT + c] = bn.running_mean[c];
            bn.stats[3 * COUT + c] = invstd;
        }
        return;
    }
    block_sums<2 * COUT>(s, wred, vals);
    if (!hand_in<kBlock>(vals, 2 * COUT, partial, counter, tot, &flag)) return;
    if (threadIdx.x < COUT) {
        const int c = threadIdx.x;
        const double M = (double)total;
        const double mean = tot[c] / M;
        double var = tot[COUT + c] / M - mean * mean;           // biased: what the batch is normalised with
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)bn.eps);
        const double scale = (double)bn.gamma[c] * invstd;
        bn.stats[c] = (float)scale;
        bn.stats[COUT + c] = (float)((double)bn.beta[c] - mean * scale);
        bn.stats[2 * COUT + c] = (float)mean;
        bn.stats[3 * COUT + c] = (float)invstd;
        if (bn.update) {            // running statistics: unbiased variance, exponential average
            const double m = bn.momentum, unb = M > 1.0 ? var * M / (M - 1.0) : var;
            bn.running_mean[c] = (float)((1.0 - m) * (double)bn.running_mean[c] + m * mean);
            bn.running_var[c] = (float)((1.0 - m) * (double)bn.running_var[c] + m * unb);
        }
    }
    if (threadIdx.x == 0 && bn.update && bn.num_batches) bn.num_batches[0] += 1;
}

// ---- ConvModifier (classic.py:8-42): one convolution with a 1- or 2-wide kernel per axis whose padding exceeds
// kernel - 1, i.e. channel mixing (+ a 2-tap blur on an axis whose size difference is odd) inside a frame that equals
// the bias.  weight (COUT, C, KH, KW); out[i][j] = bias + sum w[.][dy][dx] z[i - ph + dy][j - pw + dx], z = 0 outside.
template <int COUT>
__global__ __launch_bounds__(kBlock) void k_ct_frame_fwd(
    const float *__restrict__ x, const float *__restrict__ in_aff, const float *__restrict__ weight,
    const float *__restrict__ bias, float *__restrict__ out, long long N, int C, int H, int W, int HT, int WT,
    int ph, int pw, int KH, int KW)
{
    const long long total = N * (long long)HT * WT;
    const long long plane_in = (long long)H * W, plane_out = (long long)HT * WT;
    const int T = KH * KW;
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kBlock) {
        const int j = (int)(idx % WT);
        const int i = (int)((idx / WT) % HT);
        const long long n = idx / plane_out;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
        for (int dy = 0; dy < KH; ++dy) {
            const int ii = i - ph + dy;
            if (ii < 0 || ii >= H) continue;
            for (int dx = 0; dx < KW; ++dx) {
                const int jj = j - pw + dx;
                if (jj < 0 || jj >= W) continue;
                const float *src = x + n * C * plane_in + (long long)ii * W + jj;
                for (int c = 0; c < C; ++c) {
                    float v = src[c * plane_in];
                    if (in_aff) v = fmaf(v, in_aff[c], in_aff[C + c]);
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
                        acc[co] = fmaf(weight[(co * C + c) * T + dy * KW + dx], v, acc[co]);
                }
            }
        }
        float *dst = out + n * COUT * plane_out + (long long)i * WT + j;
#pragma unroll
        for (int co = 0; co < COUT; ++co) dst[co * plane_out] = acc[co] + bias[co];
    }
}

// Reverse mode of the above.  g_out (N, COUT, HT, WT) -> g_in (N, C, H, W) = d(loss)/d(the affine'd input); per launch
// sums: dW (COUT x C x KH x KW) | sum g_in (C) | sum g_in x (C) | db (COUT)   [x = the RAW input: what the BatchNorm in
// front saw].  One sample per workgroup pass; dynamic LDS: the window of g_out the input can reach (COUT x (H + KH - 1)
// x (W + KW - 1)), x (C x HW), g_in (C x HW).  NO: sums per lane (K1 <= NO * 256).
template <int COUT, int NO>
__global__ __launch_bounds__(kBlock) void k_ct_frame_bwd(
    const float *__restrict__ g_out, const float *__restrict__ x, const float *__restrict__ in_aff,
    const float *__restrict__ weight, float *__restrict__ g_in, float *partial, unsigned long long *counter,
    float *sums_out, BnBwd bn, long long N, int C, int H, int W, int HT, int WT, int ph, int pw, int KH, int KW)
{
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int HW = H * W, HTWT = HT * WT, T = KH * KW;
    const int GH = H + KH - 1, GW = W + KW - 1, GHW = GH * GW;
    const int HWP = HW | 1, GHWP = GHW | 1;         // odd plane pitches: lanes that differ in the channel meet no bank
    const int K1 = COUT * C * T + 2 * C, K = K1 + COUT;
    const int SL = (NO == 1) ? kBlock / K1 : 1;     // slices of the pixels per sum
    double *tot = ldsd;                             // K doubles
    float *gI = reinterpret_cast<float *>(tot + K), *xr = gI + COUT * GHWP, *gin = xr + C * HWP, *wl = gin + C * HWP;
    float *ia = wl + COUT * C * T, *vals = ia + 2 * C, *wred = vals + K;       // 2 C | K | 4 COUT floats
    __shared__ int flag;
    for (int i = threadIdx.x; i < COUT * C * T; i += kBlock) wl[i] = weight[i];
    for (int i = threadIdx.x; i < C; i += kBlock) {
        ia[i] = in_aff ? in_aff[i] : 1.0f;
        ia[C + i] = in_aff ? in_aff[C + i] : 0.0f;
    }
    float dbl[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) dbl[co] = 0.0f;
    float acc[NO];
#pragma unroll
    for (int q = 0; q < NO; ++q) acc[q] = 0.0f;
    const int r0 = ph - (KH - 1), s0 = pw - (KW - 1);          // g_out position of window entry (0, 0)
    for (long long n = blockIdx.x; n < N; n += gridDim.x) {
        __syncthreads();                            // (the previous pass is done with the LDS; wl / ia are written)
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float *src = g_out + (n * COUT + co) * HTWT;
            for (int pos = threadIdx.x; pos < HTWT; pos += kBlock) {
                const float g = src[pos];
                dbl[co] += g;
                const int r = pos / WT - r0, q = pos % WT - s0;
                if (r >= 0 && r < GH && q >= 0 && q < GW) gI[co * GHWP + r * GW + q] = g;
            }
        }
        for (int i = threadIdx.x; i < C * HW; i += kBlock) xr[(i / HW) * HWP + i % HW] = x[n * C * HW + i];
        __syncthreads();
        for (int i = threadIdx.x; i < C * HW; i += kBlock) {
            const int c = i / HW, pix = i - c * HW, yy = pix / W, xx = pix - yy * W;
            float gi = 0.0f;
#pragma unroll
            for (int co = 0; co < COUT; ++co)
                for (int dy = 0; dy < KH; ++dy)
                    for (int dx = 0; dx < KW; ++dx)
                        gi = fmaf(wl[(co * C + c) * T + dy * KW + dx],
                                  gI[co * GHWP + (yy + KH - 1 - dy) * GW + xx + KW - 1 - dx], gi);
            gin[c * HWP + pix] = gi;
            g_in[n * C * HW + i] = gi;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            // sum number o: NO == 1 -> its pixels in SL slices over the lanes (o = lane % K1, slice = lane / K1)
            const int o = (NO == 1) ? (int)threadIdx.x % K1 : (int)threadIdx.x + q * kBlock;
            const int sl = (NO == 1) ? (int)threadIdx.x / K1 : 0;
            float a = acc[q];
            if (sl >= SL) {
            } else if (o < COUT * C * T) {
                const int tap = o % T, cc = o / T, co = cc / C, c = cc - co * C, dy = tap / KW, dx = tap - dy * KW;
                const float sc = ia[c], sh = ia[C + c];
                const float *gsrc = gI + co * GHWP + (KH - 1 - dy) * GW + KW - 1 - dx;
                for (int pix = sl; pix < HW; pix += SL) {
                    const int yy = pix / W, xx = pix - yy * W;
                    a = fmaf(gsrc[yy * GW + xx], fmaf(xr[c * HWP + pix], sc, sh), a);
                }
            } else if (o < COUT * C * T + C) {
                const int c = o - COUT * C * T;
                for (int pix = sl; pix < HW; pix += SL) a += gin[c * HWP + pix];
            } else if (o < K1) {
                const int c = o - COUT * C * T - C;
                for (int pix = sl; pix < HW; pix += SL) a = fmaf(gin[c * HWP + pix], xr[c * HWP + pix], a);
            }
            acc[q] = a;
        }
    }
    __syncthreads();
    if (NO == 1) {          // slices of one sum, added in order
        float *red = wred + 4 * COUT;       // kBlock floats behind the rest
        red[threadIdx.x] = acc[0];
        __syncthreads();
        if ((int)threadIdx.x < K1) {
            float t = 0.0f;
            for (int q = 0; q < SL; ++q) t += red[q * K1 + threadIdx.x];
            vals[threadIdx.x] = t;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int q = 0; q < NO; ++q)
            if ((int)threadIdx.x + q * kBlock < K1) vals[threadIdx.x + q * kBlock] = acc[q];
    }
    block_sums<COUT>(dbl, wred, vals + K1);
    if (!hand_in<kBlock>(vals, K, partial, counter, tot, &flag)) return;
    for (int k = threadIdx.x; k < K; k += kBlock) sums_out[k] = (float)tot[k];
    bn_backward_finish(tot + COUT * C * T, tot + COUT * C * T + C, C, bn);
}

// ---- reverse mode of one block ------------------------------------------------------------------------------------
// gz (N, COUT, H/2, H/2): d(loss)/d(BatchNorm output); coef: that BatchNorm's c1 | c2 | c3; y, amax: the forward's pooled
// output and arg-max; in (N, CIN, H, H) + in_aff: the forward's input.  Writes g_in = d(loss)/d(affine'd input) and hands
// in dW (COUT x CIN x 9) | db (COUT) | sum g_in (CIN) | sum g_in * in (CIN).  Square planes, H in {8, 16, 32}.
template <int CIN, int COUT, int H>
__global__ __launch_bounds__(kBlock) void k_ct_block_bwd(
    const float *__restrict__ gz, const float *__restrict__ coef, const float *__restrict__ y,
    const unsigned char *__restrict__ amax, const float *__restrict__ in, const float *__restrict__ in_aff,
    const float *__restrict__ w, float *__restrict__ g_in, float *partial, unsigned long long *counter,
    float *sums_out, BnBwd bn, long long N)
{
    constexpr int P = H + 2, PP = P * P, HP = H / 2, CELLS = HP * HP, HH = H * H;
    constexpr int PAIRS = COUT * CIN, GROUPS = kBlock / PAIRS;
    constexpr int K = PAIRS * 9 + COUT + 2 * CIN;
    static_assert(GROUPS >= 1 && GROUPS * PAIRS == kBlock, "pairs must divide the workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *zin = lds;                               // CIN x P x P, zero halo
    float *dc = zin + CIN * PP;                     // COUT x P x P, zero halo: d(loss)/d(convolution output)
    // (pitches CELLS + 1 floats / CELLS + 4 bytes: lanes that differ in c_out at the same cell would otherwise meet
    // in one bank -- 38 % of this kernel's LDS cycles were bank conflicts before)
    constexpr int CP = CELLS + 1, CPB = CELLS + 4;
    float *dv = dc + COUT * PP;                     // COUT x CP: the one non-zero of every 2x2 window
    unsigned char *da = reinterpret_cast<unsigned char *>(dv + COUT * CP);          // COUT x CPB: its position
    __shared__ float cf[3 * COUT], ia[2 * CIN], vals[K], wred[4 * 2 * CIN];
    __shared__ double tot[K];
    __shared__ int flag;
    for (int i = threadIdx.x; i < (CIN + COUT) * PP; i += kBlock) lds[i] = 0.0f;
    for (int i = threadIdx.x; i < 3 * COUT; i += kBlock) cf[i] = coef[i];
    for (int i = threadIdx.x; i < CIN; i += kBlock) {
        ia[i] = in_aff ? in_aff[i] : 1.0f;
        ia[CIN + i] = in_aff ? in_aff[CIN + i] : 0.0f;
    }
    const int pair = threadIdx.x % PAIRS, grp = threadIdx.x / PAIRS;
    const int pco = pair / CIN, pci = pair % CIN;
    float aw[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) aw[k] = 0.0f;
    float sg[2 * CIN];
#pragma unroll
    for (int k = 0; k < 2 * CIN; ++k) sg[k] = 0.0f;
    for (long long n = blockIdx.x; n < N; n += gridDim.x) {
        __syncthreads();
        // (a) the convolution-output gradient: BatchNorm backward, un-pool, ReLU
        for (int it = threadIdx.x; it < COUT * CELLS; it += kBlock) {
            const int co = it / CELLS, cell = it - co * CELLS;
            const long long gi = n * COUT * CELLS + it;
            const float g = gz[gi], yv = y[gi];
            const int a = amax[gi];
            const float d = yv > 0.0f ? fmaf(cf[co], g, fmaf(cf[COUT + co], yv, cf[2 * COUT + co])) : 0.0f;
            dv[co * CP + cell] = d;
            da[co * CPB + cell] = (unsigned char)a;
            const int py = cell / HP, px = cell - py * HP;
            float *dst = dc + co * PP + (2 * py + 1) * P + 2 * px + 1;
            dst[0] = a == 0 ? d : 0.0f;
            dst[1] = a == 1 ? d : 0.0f;
            dst[P] = a == 2 ? d : 0.0f;
            dst[P + 1] = a == 3 ? d : 0.0f;
        }
        for (int it = threadIdx.x; it < CIN * HH; it += kBlock) {
            const int ci = it / HH, r = it - ci * HH, yy = r / H, xx = r - yy * H;
            zin[ci * PP + (yy + 1) * P + xx + 1] = fmaf(in[n * CIN * HH + it], ia[ci], ia[CIN + ci]);
        }
        __syncthreads();
        // (c) weight gradient: this thread's (c_out, c_in) pair, its share of the pooled cells, 9 taps
        for (int cell = grp; cell < CELLS; cell += GROUPS) {
            const float d = dv[pco * CP + cell];
            const int a = da[pco * CPB + cell];
            const int py = cell / HP, px = cell - py * HP;
            const float *src = zin + pci * PP + (2 * py + (a >> 1)) * P + 2 * px + (a & 1);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) aw[3 * ky + kx] = fmaf(d, src[ky * P + kx], aw[3 * ky + kx]);
            aw[9] += d;
        }
        // (b) input gradient: full correlation with the flipped kernel, all input channels of one position
        for (int pos = threadIdx.x; pos < HH; pos += kBlock) {
            const int yy = pos / H, xx = pos - yy * H;
            float gi[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) gi[ci] = 0.0f;
#pragma unroll 1
            for (int co = 0; co < COUT; ++co) {         // rolled: one output channel's CIN x 9 weights in SGPRs at a time
                const float *src = dc + co * PP + yy * P + xx;
                const float *wc = w + co * CIN * 9;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float dval = src[(2 - ky) * P + (2 - kx)];
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci)
                            gi[ci] = fmaf(wc[(ci * 3 + ky) * 3 + kx], dval, gi[ci]);
                    }
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const long long at = (n * CIN + ci) * HH + pos;
                g_in[at] = gi[ci];
                sg[ci] += gi[ci];
                sg[CIN + ci] = fmaf(gi[ci], in[at], sg[CIN + ci]);
            }
        }
    }
    __syncthreads();
    float *red = lds;                               // GROUPS x PAIRS x 10 (the launch reserves at least that much)
#pragma unroll
    for (int k = 0; k < 10; ++k) red[(grp * PAIRS + pair) * 10 + k] = aw[k];
    __syncthreads();
    for (int o = threadIdx.x; o < PAIRS * 9; o += kBlock) {
        const int pr = o / 9, k = o - pr * 9;
        float s = 0.0f;
        for (int g = 0; g < GROUPS; ++g) s += red[(g * PAIRS + pr) * 10 + k];
        vals[o] = s;
    }
    if (threadIdx.x < COUT) {
        float s = 0.0f;
        for (int g = 0; g < GROUPS; ++g) s += red[(g * PAIRS + threadIdx.x * CIN) * 10 + 9];
        vals[PAIRS * 9 + threadIdx.x] = s;
    }
    block_sums<2 * CIN>(sg, wred, vals + PAIRS * 9 + COUT);
    if (!hand_in<kBlock>(vals, K, partial, counter, tot, &flag)) return;
    for (int k = threadIdx.x; k < PAIRS * 9 + COUT; k += kBlock) sums_out[k] = (float)tot[k];
    bn_backward_finish(tot + PAIRS * 9 + COUT, tot + PAIRS * 9 + COUT + CIN, CIN, bn);
}

// ---- weight gradient of the Linear layer behind the second ConvModifier ---------------------------------------------
// Its input equals the modifier's bias outside the 16 interior pixels a (N, 16), so
//   dW[m][f] = frame(f) ? bias * db[m] : sum_n g[n][m] a[n][interior index of f],   db[m] = sum_n g[n][m]:
// a contraction over the batch with 16 columns on one side (the library GEMM runs it as one tile: 958 us at N = 1 000,
// M = 784).  A lane owns one output row m and keeps its 16 sums in registers: g[n][m .. m + 63] is one coalesced load
// per wavefront and row, a[n][0..15] the same 64 bytes for every lane (staged 64 rows at a time in the wavefront's own LDS
// tile, read back as broadcasts).  The eight wavefronts of a workgroup take an eighth of the rows each, in order, and
// their sums are added in wavefront order: deterministic.  The workgroup's 64 x F block of dW is assembled in LDS and
// written as one contiguous stretch.
constexpr int kWgradWaves = 8;      // wavefronts (row slices) per workgroup of k_ct_linear_wgrad

__global__ __launch_bounds__(kWgradWaves * 64) void k_ct_linear_wgrad(
    const float *__restrict__ g, const float *__restrict__ a, const float *__restrict__ frame_bias,
    float *__restrict__ dW, float *__restrict__ db, long long N, int M, int HT, int WT, int top, int left)
{
    extern __shared__ __attribute__((aligned(16))) float lw[];
    constexpr int NW = kWgradWaves, BLOCK = NW * 64;
    const int F = HT * WT;
    float *part = lw;                       // NW x 64 x 17: per wavefront, per lane: 16 sums + the plain sum
    float *atile = lw + NW * 64 * 17;       // NW x 64 x 16: the wavefront's current 64 rows of a
    float *tile = atile + NW * 64 * 16;     // 64 x F
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int m0 = blockIdx.x * 64, m = m0 + lane;
    const bool live = m < M;
    const long long per = (N + NW - 1) / NW, lo = wave * per, hi = (lo + per < N) ? lo + per : N;
    float acc[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) acc[j] = 0.0f;
    float4 *mine = reinterpret_cast<float4 *>(atile + (wave * 64 + lane) * 16);
    const float4 *rows4 = reinterpret_cast<const float4 *>(atile + wave * 64 * 16);
    for (long long c0 = 0; c0 < per; c0 += 64) {        // (the same trip count in every wavefront: barriers inside)
        const long long n0 = lo + c0;
        __syncthreads();
        if (n0 + lane < hi) {
            const float4 *src = reinterpret_cast<const float4 *>(a + (n0 + lane) * 16);
            mine[0] = src[0]; mine[1] = src[1]; mine[2] = src[2]; mine[3] = src[3];
        }
        __syncthreads();
        const int cnt = (hi - n0 >= 64) ? 64 : (hi > n0 ? (int)(hi - n0) : 0);
        if (!live) continue;
        int r = 0;
        for (; r + 8 <= cnt; r += 8) {
            float gv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) gv[q] = g[(n0 + r + q) * M + m];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 a0 = rows4[(r + q) * 4], a1 = rows4[(r + q) * 4 + 1], a2 = rows4[(r + q) * 4 + 2],
                             a3 = rows4[(r + q) * 4 + 3];
                const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w,
                                      a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(gv[q], av[j], acc[j]);
                acc[16] += gv[q];
            }
        }
        for (; r < cnt; ++r) {
            const float gv = g[(n0 + r) * M + m];
            const float *ar = atile + (wave * 64 + r) * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(gv, ar[j], acc[j]);
            acc[16] += gv;
        }
    }
#pragma unroll
    for (int j = 0; j < 17; ++j) part[(wave * 64 + lane) * 17 + j] = acc[j];
    __syncthreads();
    const float fb = frame_bias[0];
    for (int i = threadIdx.x; i < 64 * 17; i += BLOCK) {       // the slices' sums, added in wavefront order
        const int r = i / 17, j = i - r * 17;
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += part[(w * 64 + r) * 17 + j];
        part[r * 17 + j] = t;               // (wavefront 0's slot: read above by this lane only)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * F; i += BLOCK) {
        const int r = i / F, f = i - r * F;
        const int ii = f / WT - top, jj = f % WT - left;
        tile[i] = (ii >= 0 && ii < 4 && jj >= 0 && jj < 4) ? part[r * 17 + ii * 4 + jj] : fb * part[r * 17 + 16];
    }
    __syncthreads();
    const int rows = (M - m0 < 64) ? M - m0 : 64;
    float *dst = dW + (long long)m0 * F;
    for (int i = threadIdx.x; i < rows * F; i += BLOCK) dst[i] = tile[i];
    if ((int)threadIdx.x < rows) db[m0 + threadIdx.x] = part[threadIdx.x * 17 + 16];
}

// ---- the Linear layer behind the second ConvModifier, forward and input gradient ------------------------------------
// Its input is the modifier's bias outside the 4 x 4 interior, so out[n][m] = b_eff[m] + sum_{j<16} a16[n][j] W16[m][j]
// with W16 = the 16 interior columns of the weight and b_eff = bias + frame_bias * (sum of the other columns): a
// 16-term product instead of a 100-term one (the same fold image_program.py applies at inference), written out so
// that the network has no GEMM-library call (a training step with them cannot be captured into a hipGraph on this stack)
// and every sum has a fixed order.
__global__ __launch_bounds__(kBlock) void k_ct_linear_prep(
    const float *__restrict__ Wt, const float *__restrict__ bias, const float *__restrict__ frame_bias,
    float *__restrict__ W16, float *__restrict__ b_eff, float *__restrict__ wfr, int M, int HT, int WT, int top, int left)
{
    // one wavefront per output row m: its F weights read coalesced, the frame columns added over the lanes
    const int m = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const int F = HT * WT;
    const float *row = Wt + (long long)m * F;
    float fr = 0.0f;
    for (int f = lane; f < F; f += kWave) {
        const int ii = f / WT - top, jj = f % WT - left;
        const float v = row[f];
        if (ii >= 0 && ii < 4 && jj >= 0 && jj < 4) W16[m * 16 + ii * 4 + jj] = v;
        else fr += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fr += __shfl_xor(fr, o, kWave);
    if (lane == 0) {
        wfr[m] = fr;
        b_eff[m] = bias[m] + frame_bias[0] * fr;
    }
}

// out[n][m] = b_eff[m] + sum_j a16[n][j] W16[m][j]: a lane keeps ITS row of W16 in registers and walks 16 rows n (their
// a16 rows are one address per wavefront); the write of out is the traffic that counts.
__global__ __launch_bounds__(kBlock) void k_ct_linear16_fwd(
    const float *__restrict__ a16, const float *__restrict__ W16, const float *__restrict__ b_eff,
    float *__restrict__ out, long long N, int M)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    const float4 *wr = reinterpret_cast<const float4 *>(W16 + (long long)m * 16);
    const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
    const float be = b_eff[m];
    const long long n0 = (long long)blockIdx.y * 16;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const long long n = n0 + r;
        if (n >= N) break;
        const float4 *ar = reinterpret_cast<const float4 *>(a16 + n * 16);
        const float4 a0 = ar[0], a1 = ar[1], a2 = ar[2], a3 = ar[3];
        float acc = be;
        acc = fmaf(a0.x, w0.x, acc); acc = fmaf(a0.y, w0.y, acc); acc = fmaf(a0.z, w0.z, acc); acc = fmaf(a0.w, w0.w, acc);
        acc = fmaf(a1.x, w1.x, acc); acc = fmaf(a1.y, w1.y, acc); acc = fmaf(a1.z, w1.z, acc); acc = fmaf(a1.w, w1.w, acc);
        acc = fmaf(a2.x, w2.x, acc); acc = fmaf(a2.y, w2.y, acc); acc = fmaf(a2.z, w2.z, acc); acc = fmaf(a2.w, w2.w, acc);
        acc = fmaf(a3.x, w3.x, acc); acc = fmaf(a3.y, w3.y, acc); acc = fmaf(a3.z, w3.z, acc); acc = fmaf(a3.w, w3.w, acc);
        out[n * M + m] = acc;
    }
}

// g16[n][j] = sum_m g[n][m] W16[m][j]: a workgroup takes 4 rows n; wavefront w walks the m with m / 8 % 4 == w (eight
// loads in flight per lane), lane (row, j); the four partial sums are added in wavefront order.
__global__ __launch_bounds__(kBlock) void k_ct_linear16_bwd_input(
    const float *__restrict__ g, const float *__restrict__ W16, float *__restrict__ g16, long long N, int M)
{
    __shared__ float part[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (lane >> 4);
    const int j = lane & 15;
    float acc = 0.0f;
    if (n < N) {
        const float *gr = g + n * M;
        const float *wc = W16 + j;
        int m = wave * 8;
        for (; m + 8 <= M; m += 32) {
            float gv[8], wv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                gv[q] = gr[m + q];
                wv[q] = wc[(m + q) * 16];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fmaf(gv[q], wv[q], acc);
        }
        for (; m < M; ++m) acc = fmaf(gr[m], wc[m * 16], acc);      // (the ragged last block of 8, if it is this wavefront's)
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && n < N) g16[n * 16 + j] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// out[0] += sum_m a[m] b[m], one workgroup, lanes in index order then a fixed tree: deterministic
__global__ __launch_bounds__(kBlock) void k_ct_dot_add(const float *__restrict__ a, const float *__restrict__ b,
                                                       float *out, int M)
{
    __shared__ float red[kBlock];
    float t = 0.0f;
    for (int m = threadIdx.x; m < M; m += kBlock) t = fmaf(a[m], b[m], t);
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += red[0];
}

// The same product for M % 4 == 0 (every affine / shift coupling: M = 2 x targets): a workgroup takes 16 rows n; lane (row,
// quarter q of the 16 outputs) keeps four sums and reads W16[m][4q .. 4q + 3] as ONE 16-byte load and g[n][m .. m + 3] as
// one -- 5 loads per 16 multiply-adds where the kernel above issues 16 per 8; wavefront w takes the m with m / 4 % 4 == w, and
// the four partial sums are added in wavefront order.
__global__ __launch_bounds__(kBlock) void k_ct_linear16_bwd_input_v4(
    const float *__restrict__ g, const float *__restrict__ W16, float *__restrict__ g16, long long N, int M)
{
    __shared__ float4 part[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 16 + (lane >> 2);
    const int q = lane & 3;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (n < N) {
        const float *gr = g + n * M;
        const float4 *wq = reinterpret_cast<const float4 *>(W16) + q;            // row m: wq[4 m]
        for (int m = wave * 4; m < M; m += 16) {
            const float4 gv = *reinterpret_cast<const float4 *>(gr + m);
            const float4 w0 = wq[4 * m], w1 = wq[4 * m + 4], w2 = wq[4 * m + 8], w3 = wq[4 * m + 12];
            acc.x = fmaf(gv.x, w0.x, acc.x); acc.y = fmaf(gv.x, w0.y, acc.y); acc.z = fmaf(gv.x, w0.z, acc.z); acc.w = fmaf(gv.x, w0.w, acc.w);
            acc.x = fmaf(gv.y, w1.x, acc.x); acc.y = fmaf(gv.y, w1.y, acc.y); acc.z = fmaf(gv.y, w1.z, acc.z); acc.w = fmaf(gv.y, w1.w, acc.w);
            acc.x = fmaf(gv.z, w2.x, acc.x); acc.y = fmaf(gv.z, w2.y, acc.y); acc.z = fmaf(gv.z, w2.z, acc.z); acc.w = fmaf(gv.z, w2.w, acc.w);
            acc.x = fmaf(gv.w, w3.x, acc.x); acc.y = fmaf(gv.w, w3.y, acc.y); acc.z = fmaf(gv.w, w3.z, acc.z); acc.w = fmaf(gv.w, w3.w, acc.w);
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && n < N) {
        const float4 a0 = part[0][lane], a1 = part[1][lane], a2 = part[2][lane], a3 = part[3][lane];
        float4 t;
        t.x = (a0.x + a1.x) + (a2.x + a3.x);
        t.y = (a0.y + a1.y) + (a2.y + a3.y);
        t.z = (a0.z + a1.z) + (a2.z + a3.z);
        t.w = (a0.w + a1.w) + (a2.w + a3.w);
        reinterpret_cast<float4 *>(g16 + n * 16)[q] = t;
    }
}

}  // namespace tfk

using namespace tfk;

namespace {

bool block_ok(int c_in, int c_out) { return (c_in == 4 && c_out == 8) || (c_in == 8 && c_out == 8) || (c_in == 8 && c_out == 4); }

int bwd_grid(int64_t N, int per_cu)
{
    int64_t g = (int64_t)cu_count() * per_cu;
    if (g > kCtMaxGrid) g = kCtMaxGrid;
    return (int)(N < g ? N : g);
}

}  // namespace

extern "C" {

int64_t tfk_convnet_train_workspace_bytes(void)
{
    return kCtHeaderBytes + (int64_t)(kCtMaxGrid + kCtMaxGrid / kCtCluster) * kCtMaxK * (int64_t)sizeof(float);
}

int tfk_convnet_train_block_supported(int32_t c_in, int32_t c_out, int32_t H)
{
    // (the reference's network: 4 x 32 x 32 -> 8 x 16 x 16 -> 8 x 8 x 8 -> 4 x 4 x 4, classic.py:88-104)
    return ((c_in == 4 && c_out == 8 && H == 32) || (c_in == 8 && c_out == 8 && H == 16) || (c_in == 8 && c_out == 4 && H == 8)) ? 1 : 0;
}

int tfk_convnet_train_block_fwd(const float *x, const float *in_affine, const float *weight, const float *bias,
                                float *y, uint8_t *argmax, const float *bn_weight, const float *bn_bias,
                                float *running_mean, float *running_var, int64_t *num_batches_tracked, float eps,
                                float momentum, int32_t training, int32_t update_running, float *stats,
                                void *workspace, int64_t N, int32_t c_in, int32_t c_out, int32_t H, int32_t W,
                                void *stream)
{
    const char *fn = "tfk_convnet_train_block_fwd";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (!block_ok(c_in, c_out)) return fail(TFK_EINVAL, "%s: channels %d -> %d (4->8, 8->8, 8->4)", fn, c_in, c_out);
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return fail(TFK_EINVAL, "%s: H = %d, W = %d must be even and >= 2", fn, H, W);
    if (N == 0) return TFK_OK;
    if (!x || !weight || !bias || !y || !argmax || !bn_weight || !bn_bias || !running_mean || !running_var || !stats ||
        !workspace)
        return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t blocks = (N * (int64_t)(H / 2) * (W / 2) + kBlock - 1) / kBlock;
    const int grid = (int)(blocks < kCtMaxGrid ? blocks : kCtMaxGrid);
    unsigned long long *counter = static_cast<unsigned long long *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + kCtHeaderBytes);
    BnFwd bn{bn_weight, bn_bias, running_mean, running_var, reinterpret_cast<long long *>(num_batches_tracked), stats,
             eps, momentum, training ? 1 : 0, (training && update_running) ? 1 : 0};
#define TFK_CT(CI, CO)                                                                                            \
    hipLaunchKernelGGL((k_ct_block_fwd<CI, CO>), dim3(grid), dim3(kBlock), 0, s, x, in_affine, weight, bias, y, \
                       argmax, partial, counter, bn, (long long)N, H, W)
    if (c_in == 4) TFK_CT(4, 8);
    else if (c_out == 8) TFK_CT(8, 8);
    else TFK_CT(8, 4);
#undef TFK_CT
    return check_launch(fn);
}

// padding of the ConvModifier's convolution along one axis (classic.py:17-29): out = in + 2 pad - kernel + 1
static bool frame_axis(int in, int out, int kernel, int *pad)
{
    if (kernel < 1 || kernel > 2 || out < in) return false;
    const int twice = out - in + kernel - 1;
    if (twice & 1) return false;
    *pad = twice / 2;
    return *pad >= kernel - 1;
}

int tfk_convnet_train_frame_fwd(const float *x, const float *in_affine, const float *weight, const float *bias,
                                float *out, int64_t N, int32_t c_in, int32_t c_out, int32_t H, int32_t W,
                                int32_t H_out, int32_t W_out, int32_t kh, int32_t kw, void *stream)
{
    const char *fn = "tfk_convnet_train_frame_fwd";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (c_out != 1 && c_out != 4) return fail(TFK_EINVAL, "%s: c_out = %d (1 or 4)", fn, c_out);
    int ph = 0, pw = 0;
    if (c_in < 1 || H < 1 || W < 1 || !frame_axis(H, H_out, kh, &ph) || !frame_axis(W, W_out, kw, &pw))
        return fail(TFK_EINVAL, "%s: need c_in >= 1 and a %d x %d kernel that maps %dx%d to %dx%d with padding >= kernel - 1",
                    fn, kh, kw, H, W, H_out, W_out);
    if (N == 0) return TFK_OK;
    if (!x || !weight || !bias || !out) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = grid_for(N * (int64_t)H_out * W_out, kBlock);
    if (c_out == 4)
        hipLaunchKernelGGL((k_ct_frame_fwd<4>), dim3(grid), dim3(kBlock), 0, s, x, in_affine, weight, bias, out,
                           (long long)N, c_in, H, W, H_out, W_out, ph, pw, kh, kw);
    else
        hipLaunchKernelGGL((k_ct_frame_fwd<1>), dim3(grid), dim3(kBlock), 0, s, x, in_affine, weight, bias, out,
                           (long long)N, c_in, H, W, H_out, W_out, ph, pw, kh, kw);
    return check_launch(fn);
}

int tfk_convnet_train_frame_bwd(const float *g_out, const float *x, const float *in_affine, const float *weight,
                                float *g_in, float *sums, const float *bn_stats, float *bn_coef, float *bn_dweight,
                                float *bn_dbias, int32_t bn_training, void *workspace, int64_t N, int32_t c_in,
                                int32_t c_out, int32_t H, int32_t W, int32_t H_out, int32_t W_out, int32_t kh,
                                int32_t kw, void *stream)
{
    const char *fn = "tfk_convnet_train_frame_bwd";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (c_out != 1 && c_out != 4) return fail(TFK_EINVAL, "%s: c_out = %d (1 or 4)", fn, c_out);
    int ph = 0, pw = 0;
    if (c_in < 1 || H < 1 || W < 1 || !frame_axis(H, H_out, kh, &ph) || !frame_axis(W, W_out, kw, &pw))
        return fail(TFK_EINVAL, "%s: need c_in >= 1 and a %d x %d kernel that maps %dx%d to %dx%d with padding >= kernel - 1",
                    fn, kh, kw, H, W, H_out, W_out);
    const int T = kh * kw, K1 = c_out * c_in * T + 2 * c_in, K = K1 + c_out;
    if (K1 > 3 * kBlock || K > kCtMaxK)
        return fail(TFK_EINVAL, "%s: %d x %d channels x %d taps: too many sums for one workgroup", fn, c_out, c_in, T);
    const int64_t lds_floats = (int64_t)c_out * (((H + kh - 1) * (W + kw - 1)) | 1) + 2 * (int64_t)c_in * ((H * W) | 1) +
                               c_out * c_in * T + 2 * c_in + K + 4 * c_out + 2 * (int64_t)K + kBlock;
    if (lds_floats * 4 > 150 * 1024) return fail(TFK_EINVAL, "%s: %d x %d x %d input does not fit the LDS", fn, c_in, H, W);
    if (bn_stats && (!bn_coef || !bn_dweight || !bn_dbias)) return fail(TFK_EINVAL, "%s: BatchNorm outputs missing", fn);
    if (N == 0) return TFK_OK;
    if (!g_out || !x || !weight || !g_in || !sums || !workspace) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = bwd_grid(N, 4);            // (few sums per workgroup: the work per sample decides)
    unsigned long long *counter = static_cast<unsigned long long *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + kCtHeaderBytes);
    BnBwd bn{bn_stats, bn_coef, bn_dweight, bn_dbias, (double)N * H * W, bn_training ? 1 : 0};
    const size_t lds = (size_t)lds_floats * 4;
#define TFK_CT(CO, NO)                                                                                                 \
    do {                                                                                                               \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ct_frame_bwd<CO, NO>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                   \
            return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);                                  \
        hipLaunchKernelGGL((k_ct_frame_bwd<CO, NO>), dim3(grid), dim3(kBlock), lds, s, g_out, x, in_affine, weight,    \
                           g_in, partial, counter, sums, bn, (long long)N, c_in, H, W, H_out, W_out, ph, pw, kh, kw);  \
    } while (0)
    if (c_out == 1) TFK_CT(1, 1);
    else if (K1 <= kBlock) TFK_CT(4, 1);
    else TFK_CT(4, 3);
#undef TFK_CT
    return check_launch(fn);
}

int tfk_convnet_train_block_bwd(const float *gz, const float *coef, const float *y, const uint8_t *argmax,
                                const float *x, const float *in_affine, const float *weight, float *g_in, float *sums,
                                const float *bn_stats, float *bn_coef, float *bn_dweight, float *bn_dbias,
                                int32_t bn_training, void *workspace, int64_t N, int32_t c_in, int32_t c_out, int32_t H,
                                void *stream)
{
    const char *fn = "tfk_convnet_train_block_bwd";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (!tfk_convnet_train_block_supported(c_in, c_out, H))
        return fail(TFK_EINVAL, "%s: %d -> %d channels on %d x %d planes (4->8 on 32, 8->8 on 16, 8->4 on 8)", fn, c_in, c_out, H, H);
    if (bn_stats && (!bn_coef || !bn_dweight || !bn_dbias)) return fail(TFK_EINVAL, "%s: BatchNorm outputs missing", fn);
    if (N == 0) return TFK_OK;
    if (!gz || !coef || !y || !argmax || !x || !weight || !g_in || !sums || !workspace)
        return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long *counter = static_cast<unsigned long long *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + kCtHeaderBytes);
    BnBwd bn{bn_stats, bn_coef, bn_dweight, bn_dbias, (double)N * H * H, bn_training ? 1 : 0};
    const int P = H + 2, cells = (H / 2) * (H / 2);
    size_t lds = (size_t)(c_in + c_out) * P * P * 4 + (size_t)c_out * (cells + 1) * 4 + (size_t)c_out * (cells + 4) + 16;
    if (lds < (size_t)kBlock * 10 * 4) lds = (size_t)kBlock * 10 * 4;       // (the cross-group reduction of dW)
    const int grid = bwd_grid(N, 2);        // (two workgroups per CU fit the LDS; measured 71 -> 49 us at 1 024 samples)
#define TFK_CT(CI, CO, HH)                                                                                             \
    do {                                                                                                               \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ct_block_bwd<CI, CO, HH>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                   \
            return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);                                \
        hipLaunchKernelGGL((k_ct_block_bwd<CI, CO, HH>), dim3(grid), dim3(kBlock), lds, s, gz, coef, y, argmax, x,     \
                           in_affine, weight, g_in, partial, counter, sums, bn, (long long)N);                         \
    } while (0)
    if (c_in == 4) TFK_CT(4, 8, 32);
    else if (c_out == 8) TFK_CT(8, 8, 16);
    else TFK_CT(8, 4, 8);
#undef TFK_CT
    return check_launch(fn);
}

int tfk_convnet_train_linear_prep(const float *weight, const float *bias, const float *frame_bias, float *W16,
                                  float *b_eff, float *w_frame, int32_t M, int32_t H_out, int32_t W_out, void *stream)
{
    const char *fn = "tfk_convnet_train_linear_prep";
    if (M < 1 || H_out < 4 || W_out < 4 || ((H_out - 4) & 1) || ((W_out - 4) & 1))
        return fail(TFK_EINVAL, "%s: M = %d, a 4 x 4 interior in the middle of an even frame (%d x %d)", fn, M, H_out, W_out);
    if (!weight || !bias || !frame_bias || !W16 || !b_eff || !w_frame) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(k_ct_linear_prep, dim3((M + 3) / 4), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       weight, bias, frame_bias, W16, b_eff, w_frame, M, H_out, W_out, (H_out - 4) / 2, (W_out - 4) / 2);
    return check_launch(fn);
}

int tfk_convnet_train_linear_fwd(const float *a16, const float *W16, const float *b_eff, float *out, int64_t N, int32_t M,
                                 void *stream)
{
    const char *fn = "tfk_convnet_train_linear_fwd";
    if (N < 0 || M < 1) return fail(TFK_EINVAL, "%s: N = %lld, M = %d", fn, (long long)N, M);
    if (N == 0) return TFK_OK;
    if (!a16 || !W16 || !b_eff || !out) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!aligned16(a16) || !aligned16(W16)) return fail(TFK_EINVAL, "%s: a16 / W16 must be 16-byte aligned", fn);
    if ((N + 15) / 16 > 65535) return fail(TFK_EINVAL, "%s: N = %lld (at most 16 * 65535 rows per call)", fn, (long long)N);
    hipLaunchKernelGGL(k_ct_linear16_fwd, dim3((unsigned)((M + kBlock - 1) / kBlock), (unsigned)((N + 15) / 16)), dim3(kBlock),
                       0, static_cast<hipStream_t>(stream), a16, W16, b_eff, out, (long long)N, M);
    return check_launch(fn);
}

int tfk_convnet_train_linear_bwd_input(const float *g, const float *W16, float *g16, int64_t N, int32_t M, void *stream)
{
    const char *fn = "tfk_convnet_train_linear_bwd_input";
    if (N < 0 || M < 1) return fail(TFK_EINVAL, "%s: N = %lld, M = %d", fn, (long long)N, M);
    if (N == 0) return TFK_OK;
    if (!g || !W16 || !g16) return fail(TFK_EINVAL, "%s: null pointer", fn);
    // (measured at M = 3 072: 1 024 rows 21 us scalar / 30 us vector -- a quarter of the workgroups --, 8 192 rows 103 / 64 us)
    if (N >= 16 * cu_count() && (M & 3) == 0 && aligned16(g) && aligned16(W16) && aligned16(g16))
        hipLaunchKernelGGL(k_ct_linear16_bwd_input_v4, dim3((unsigned)((N + 15) / 16)), dim3(kBlock), 0,
                           static_cast<hipStream_t>(stream), g, W16, g16, (long long)N, M);
    else
        hipLaunchKernelGGL(k_ct_linear16_bwd_input, dim3((unsigned)((N + 3) / 4)), dim3(kBlock), 0,
                           static_cast<hipStream_t>(stream), g, W16, g16, (long long)N, M);
    return check_launch(fn);
}

int tfk_convnet_train_linear_wgrad(const float *g, const float *a, const float *frame_bias, float *dW, float *db,
                                   int64_t N, int32_t M, int32_t H_out, int32_t W_out, void *stream)
{
    const char *fn = "tfk_convnet_train_linear_wgrad";
    if (N < 0 || M < 1) return fail(TFK_EINVAL, "%s: N = %lld, M = %d", fn, (long long)N, M);
    if (H_out < 4 || W_out < 4 || ((H_out - 4) & 1) || ((W_out - 4) & 1))
        return fail(TFK_EINVAL, "%s: the interior is 4 x 4 in the middle of an even frame (%d x %d)", fn, H_out, W_out);
    if (!g || !a || !frame_bias || !dW || !db) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)(kWgradWaves * 64 * (17 + 16) + 64 * H_out * W_out) * 4;
    if (lds > 150 * 1024) return fail(TFK_EINVAL, "%s: a %d x %d frame does not fit the LDS tile", fn, H_out, W_out);
    if (N > 0 && !aligned16(a)) return fail(TFK_EINVAL, "%s: a16 must be 16-byte aligned", fn);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ct_linear_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);
    hipLaunchKernelGGL(k_ct_linear_wgrad, dim3((M + 63) / 64), dim3(kWgradWaves * 64), lds, s, g, a, frame_bias, dW, db,
                       (long long)N, M, H_out, W_out, (H_out - 4) / 2, (W_out - 4) / 2);
    return check_launch(fn);
}

int64_t tfk_convnet_train_sums_floats(int32_t c, int32_t kh, int32_t kw, int32_t M)
{
    return (int64_t)(4 * c * kh * kw + 2 * c + 4) + (288 + 8) + (576 + 8) + (288 + 4) + 13 + 101 * (int64_t)M;
}

int tfk_convnet_train_forward(const tfk_convnet_train_plan *p, const float *x, float *theta, int64_t N, int32_t training,
                              int32_t update_running, void *stream)
{
    const char *fn = "tfk_convnet_train_forward";
    if (!p) return fail(TFK_EINVAL, "%s: null plan", fn);
    if (N <= 0) return N == 0 ? TFK_OK : fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    static const int CI[3] = {4, 8, 8}, CO[3] = {8, 8, 4}, HH[3] = {32, 16, 8};
    int rc = tfk_convnet_train_frame_fwd(x, nullptr, p->mod1_w, p->mod1_b, p->a0, N, p->c, 4, p->h, p->w, 32, 32, p->kh,
                                         p->kw, stream);
    for (int k = 0; k < 3 && rc == TFK_OK; ++k)
        rc = tfk_convnet_train_block_fwd(k == 0 ? p->a0 : p->y[k - 1], k == 0 ? nullptr : p->stats[k - 1], p->conv_w[k],
                                         p->conv_b[k], p->y[k], p->amax[k], p->bn_w[k], p->bn_b[k], p->bn_mean[k],
                                         p->bn_var[k], p->bn_count[k], p->bn_eps[k], p->bn_momentum[k], training,
                                         update_running, p->stats[k], p->workspace, N, CI[k], CO[k], HH[k], HH[k], stream);
    if (rc != TFK_OK) return rc;
    rc = tfk_convnet_train_frame_fwd(p->y[2], p->stats[2], p->mod2_w, p->mod2_b, p->a16, N, 4, 1, 4, 4, 4, 4, 1, 1, stream);
    if (rc != TFK_OK) return rc;
    float *W16 = p->lin_fold, *b_eff = W16 + 16 * (int64_t)p->M, *w_frame = b_eff + p->M;
    rc = tfk_convnet_train_linear_prep(p->lin_w, p->lin_b, p->mod2_b, W16, b_eff, w_frame, p->M, 10, 10, stream);
    if (rc != TFK_OK) return rc;
    return tfk_convnet_train_linear_fwd(p->a16, W16, b_eff, theta, N, p->M, stream);
}

int tfk_convnet_train_backward(const tfk_convnet_train_plan *p, const float *x, const float *g_theta, float *g_x,
                               float *scratch, float *bn_out, float *sums, int64_t N, int32_t training, void *stream)
{
    const char *fn = "tfk_convnet_train_backward";
    if (!p || !scratch || !bn_out || !sums) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (N <= 0) return N == 0 ? TFK_OK : fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    const int M = p->M;
    float *g16 = scratch, *gz3 = g16 + N * 16, *gz2 = gz3 + N * 64, *gz1 = gz2 + N * 512, *g_a0 = gz1 + N * 2048;
    float *s_m1 = sums, *s_b1 = s_m1 + (4 * p->c * p->kh * p->kw + 2 * p->c + 4), *s_b2 = s_b1 + 296, *s_b3 = s_b2 + 584,
          *s_m2 = s_b3 + 292, *dW_lin = s_m2 + 13, *db_lin = dW_lin + 100 * (int64_t)M;
    float *bn1 = bn_out, *bn2 = bn_out + 40, *bn3 = bn_out + 80;          // coef 3 C | d weight C | d bias C
    const float *W16 = p->lin_fold, *w_frame = W16 + 17 * (int64_t)M;
    int rc = tfk_convnet_train_linear_bwd_input(g_theta, W16, g16, N, M, stream);
    if (rc == TFK_OK) rc = tfk_convnet_train_linear_wgrad(g_theta, p->a16, p->mod2_b, dW_lin, db_lin, N, M, 10, 10, stream);
    if (rc == TFK_OK)
        rc = tfk_convnet_train_frame_bwd(g16, p->y[2], p->stats[2], p->mod2_w, gz3, s_m2, p->stats[2], bn3, bn3 + 12,
                                         bn3 + 16, training, p->workspace, N, 4, 1, 4, 4, 4, 4, 1, 1, stream);
    if (rc != TFK_OK) return rc;
    hipLaunchKernelGGL(k_ct_dot_add, dim3(1), dim3(kBlock), 0, static_cast<hipStream_t>(stream), db_lin, w_frame, s_m2 + 12, M);
    rc = check_launch(fn);
    if (rc == TFK_OK)
        rc = tfk_convnet_train_block_bwd(gz3, bn3, p->y[2], p->amax[2], p->y[1], p->stats[1], p->conv_w[2], gz2, s_b3,
                                         p->stats[1], bn2, bn2 + 24, bn2 + 32, training, p->workspace, N, 8, 4, 8, stream);
    if (rc == TFK_OK)
        rc = tfk_convnet_train_block_bwd(gz2, bn2, p->y[1], p->amax[1], p->y[0], p->stats[0], p->conv_w[1], gz1, s_b2,
                                         p->stats[0], bn1, bn1 + 24, bn1 + 32, training, p->workspace, N, 8, 8, 16, stream);
    if (rc == TFK_OK)
        rc = tfk_convnet_train_block_bwd(gz1, bn1, p->y[0], p->amax[0], p->a0, nullptr, p->conv_w[0], g_a0, s_b1, nullptr,
                                         nullptr, nullptr, nullptr, training, p->workspace, N, 4, 8, 32, stream);
    if (rc == TFK_OK)
        rc = tfk_convnet_train_frame_bwd(g_a0, x, nullptr, p->mod1_w, g_x, s_m1, nullptr, nullptr, nullptr, nullptr, training,
                                         p->workspace, N, p->c, 4, p->h, p->w, 32, 32, p->kh, p->kw, stream);
    return rc;
}

}  // extern "C"
