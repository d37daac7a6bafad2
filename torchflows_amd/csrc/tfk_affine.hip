// tfk_affine.hip -- affine / shift coupling, elementwise affine (ElementwiseAffine,
// ActNorm), permutation, diagonal-Gaussian log-prob and the fp64 sum.
//
// All of these are HBM-bound streaming kernels (0.1 FLOP/B): rows are laid out
// (N, D) row-major, a row is shared by G consecutive lanes of one 64-wide wavefront
// (G a power of two, so a wave carries 64/G rows), every lane moves 16-byte vectors,
// and the per-row log-det is reduced with __shfl_xor inside the G-lane group -- no
// LDS, no atomics, deterministic.  Grids are capped at 8 blocks per CU and
// grid-strided.  Arithmetic follows the reference's fp32 op order (file is built
// with -ffp-contract=off); reference lines are cited at each kernel.
#include "tfk_common.h"

// whole-row streams of the permutation / base-density kernels: non-temporal (-DTFK_NT_EW=0: plain; same-call A/B at 2^20
// rows x 64: permute 105 -> 100 us, the coupling that follows 109.6 -> 106; the elementwise-affine kernel 87 -> 91, so it
// keeps plain accesses)
#ifndef TFK_NT_EW
#define TFK_NT_EW 1
#endif
#if TFK_NT_EW
#define TFK_EW_LOAD4(p) nt_load4(p)
#define TFK_EW_STORE4(p, v) nt_store4(p, v)
#else
#define TFK_EW_LOAD4(p) (*(p))
#define TFK_EW_STORE4(p, v) (*(p) = (v))
#endif

namespace tfk {

// ---------------------------------------------------------------------------
// Affine coupling, HalfSplit fast path: S == T, D % 8 == 0, contiguous target.
// layers_base.py:145-163 + affine.py:39-59.
// Lane j of a row-group owns source vector j and target vector Sv + j, so every
// lane does the same work: 2 x-loads, 2 h-loads (h row = (T,2) interleaved =
// 2 float4 per target float4), 2 z-stores.
// ---------------------------------------------------------------------------
template <bool INVERSE, bool INPLACE>
__global__ __launch_bounds__(kBlock) void k_affine_half_v4(
    const float4 *x, const float4 *__restrict__ h, float4 *z, float *logdet,
    long long N, int Sv, int G, int accumulate)
{
    const int lane = threadIdx.x & (G - 1);
    const int rows_per_block = kBlock / G;
    const int Dv = 2 * Sv;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row = (long long)blockIdx.x * rows_per_block + threadIdx.x / G; row < N;
         row += stride) {
        const float4 *xr = x + row * Dv;
        const float4 *hr = h + row * Dv;   // T*2 floats == Dv float4
        float4 *zr = z + row * Dv;
        float acc = 0.0f;
        for (int j = lane; j < Sv; j += G) {
            float4 xs;
            if (!INPLACE) xs = xr[j];
            const float4 xt = xr[Sv + j];
            const float4 h0 = nt_load4(hr + 2 * j);       // (u0, b0, u1, b1); read once: non-temporal
            const float4 h1 = nt_load4(hr + 2 * j + 1);   // (u2, b2, u3, b3)
            const float a0 = aff_alpha(h0.x), a1 = aff_alpha(h0.z);
            const float a2 = aff_alpha(h1.x), a3 = aff_alpha(h1.z);
            float4 o;
            if (!INVERSE) {
                o.x = a0 * xt.x + h0.y;            // affine.py:48
                o.y = a1 * xt.y + h0.w;
                o.z = a2 * xt.z + h1.y;
                o.w = a3 * xt.w + h1.w;
            } else {
                o.x = (xt.x - h0.y) / a0;          // affine.py:59
                o.y = (xt.y - h0.w) / a1;
                o.z = (xt.z - h1.y) / a2;
                o.w = (xt.w - h1.w) / a3;
            }
            acc += log_normal(a0);                        // affine.py:42,47
            acc += log_normal(a1);
            acc += log_normal(a2);
            acc += log_normal(a3);
            if (!INPLACE) zr[j] = xs;
            nt_store4(zr + Sv + j, o);
        }
        acc = group_sum(acc, G);
        if (lane == 0) {
            const float ld = INVERSE ? -acc : acc;
            logdet[row] = accumulate ? logdet[row] + ld : ld;   // base.py:222
        }
    }
}

// ---------------------------------------------------------------------------
// Generic coupling (any D, T, mask): scalar accesses, still coalesced across the
// G lanes of a row.  MODE 0/1 = affine fwd/inv, 2/3 = shift fwd/inv.
// Dynamic LDS: D bytes of target mask, used only when tgt_idx != NULL && !inplace.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_coupling_generic(
    const float *x, const float *__restrict__ h, float *z, float *logdet,
    long long N, int D, const int *__restrict__ tgt_idx, int T, int G,
    int accumulate, int inplace)
{
    extern __shared__ unsigned char is_tgt[];
    const bool use_mask = (tgt_idx != nullptr) && !inplace;
    if (use_mask) {
        for (int e = threadIdx.x; e < D; e += kBlock) is_tgt[e] = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += kBlock) is_tgt[tgt_idx[t]] = 1;
        __syncthreads();
    }
    constexpr int P = (MODE < 2) ? 2 : 1;
    const int lane = threadIdx.x & (G - 1);
    const int rows_per_block = kBlock / G;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row = (long long)blockIdx.x * rows_per_block + threadIdx.x / G; row < N;
         row += stride) {
        const float *xr = x + row * D;
        float *zr = z + row * D;
        if (!inplace) {
            for (int e = lane; e < D; e += G) {
                const bool tgt = tgt_idx ? (is_tgt[e] != 0) : (e >= D - T);
                if (!tgt) zr[e] = xr[e];                     // clone, layers_base.py:146
            }
        }
        const float *hr = h + row * (long long)T * P;
        float acc = 0.0f;
        for (int t = lane; t < T; t += G) {
            const int idx = tgt_idx ? tgt_idx[t] : D - T + t;
            const float v = xr[idx];
            float o;
            if (MODE < 2) {
                const float a = aff_alpha(nt_load(hr + 2 * t));
                const float b = nt_load(hr + 2 * t + 1);
                o = (MODE == 0) ? a * v + b : (v - b) / a;
                acc += log_normal(a);
            } else {
                const float sh = nt_load(hr + t);
                o = (MODE == 2) ? v + sh : v - sh;           // affine.py:150,158
            }
            zr[idx] = o;
        }
        if (MODE < 2) {
            acc = group_sum(acc, G);
            if (lane == 0) {
                const float ld = (MODE == 1) ? -acc : acc;
                logdet[row] = accumulate ? logdet[row] + ld : ld;
            }
        } else if (!accumulate && logdet && lane == 0) {
            logdet[row] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------
// Elementwise affine with global parameters value (D,2).
// layers_base.py:300-318 (prepare_h + transformer), affine.py:39-70.
// Block prologue: alpha/beta of all D columns into LDS and the row-constant
// log-det sum_D log(alpha) by a fixed-order block tree.  DIVIDE selects
// (x - beta) / alpha, otherwise alpha * x + beta.
// Dynamic LDS: 2*D floats (alpha | beta) + kBlock floats.
// ---------------------------------------------------------------------------
template <bool DIVIDE, bool VEC4>
__global__ __launch_bounds__(kBlock) void k_elementwise_affine(
    const float *x, const float *__restrict__ value, float *z, float *logdet,
    long long N, int D, int G, int accumulate)
{
    extern __shared__ float smem[];
    float *alpha_s = smem;
    float *beta_s = smem + D;
    float *red = smem + 2 * D;
    float part = 0.0f;
    for (int e = threadIdx.x; e < D; e += kBlock) {
        const float a = aff_alpha(value[2 * e]);
        alpha_s[e] = a;
        beta_s[e] = value[2 * e + 1];
        part += log_normal(a);
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float ld = DIVIDE ? -red[0] : red[0];

    const int lane = threadIdx.x & (G - 1);
    const int rows_per_block = kBlock / G;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row = (long long)blockIdx.x * rows_per_block + threadIdx.x / G; row < N;
         row += stride) {
        if (VEC4) {
            const int Dv = D >> 2;
            const float4 *xr = reinterpret_cast<const float4 *>(x + row * D);
            float4 *zr = reinterpret_cast<float4 *>(z + row * D);
            const float4 *a4 = reinterpret_cast<const float4 *>(alpha_s);
            const float4 *b4 = reinterpret_cast<const float4 *>(beta_s);
            for (int v = lane; v < Dv; v += G) {
                const float4 xv = xr[v], a = a4[v], b = b4[v];    // (plain: non-temporal measured 3.5 % slower here)
                float4 o;
                if (!DIVIDE) {
                    o.x = a.x * xv.x + b.x; o.y = a.y * xv.y + b.y;
                    o.z = a.z * xv.z + b.z; o.w = a.w * xv.w + b.w;
                } else {
                    o.x = (xv.x - b.x) / a.x; o.y = (xv.y - b.y) / a.y;
                    o.z = (xv.z - b.z) / a.z; o.w = (xv.w - b.w) / a.w;
                }
                zr[v] = o;
            }
        } else {
            const float *xr = x + row * D;
            float *zr = z + row * D;
            for (int e = lane; e < D; e += G) {
                const float xv = xr[e], a = alpha_s[e], b = beta_s[e];
                zr[e] = DIVIDE ? (xv - b) / a : a * xv + b;
            }
        }
        if (lane == 0) logdet[row] = accumulate ? logdet[row] + ld : ld;
    }
}


// The same for event sizes whose parameters do not fit the LDS (images beyond (3, 32, 32): every multiscale block
// starts with an ActNorm over the whole image).  Columns go through the LDS in tiles of kColTile; one wavefront
// per row and tile (a row tile is >= 16 KB, so a wave's loads are long coalesced runs); the row-constant log-det is
// the same fixed-order sum as above (thread t adds columns t, t + 256, ... in that order), written after the last
// tile.
constexpr int kColTile = 4096;

template <bool DIVIDE, bool VEC4>
__global__ __launch_bounds__(kBlock) void k_elementwise_affine_tiled(
    const float *x, const float *__restrict__ value, float *z, float *logdet,
    long long N, int D, int accumulate)
{
    __shared__ __attribute__((aligned(16))) float alpha_s[kColTile];
    __shared__ __attribute__((aligned(16))) float beta_s[kColTile];
    __shared__ float red[kBlock];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    constexpr int rows_per_block = kBlock / kWave;
    const long long stride = (long long)gridDim.x * rows_per_block;
    float part = 0.0f;
    for (int c0 = 0; c0 < D; c0 += kColTile) {
        const int cn = (D - c0 < kColTile) ? D - c0 : kColTile;
        __syncthreads();                                  // the previous tile's readers are done
        for (int e = threadIdx.x; e < cn; e += kBlock) {
            const float a = aff_alpha(value[2 * (c0 + e)]);
            alpha_s[e] = a;
            beta_s[e] = value[2 * (c0 + e) + 1];
            part += log_normal(a);
        }
        __syncthreads();
        for (long long row = (long long)blockIdx.x * rows_per_block + wave; row < N; row += stride) {
            if (VEC4) {
                const float4 *xr = reinterpret_cast<const float4 *>(x + row * D + c0);
                float4 *zr = reinterpret_cast<float4 *>(z + row * D + c0);
                const float4 *a4 = reinterpret_cast<const float4 *>(alpha_s);
                const float4 *b4 = reinterpret_cast<const float4 *>(beta_s);
                for (int v = lane; v < (cn >> 2); v += kWave) {
                    const float4 xv = xr[v], a = a4[v], b = b4[v];
                    float4 o;
                    if (!DIVIDE) {
                        o.x = a.x * xv.x + b.x; o.y = a.y * xv.y + b.y;
                        o.z = a.z * xv.z + b.z; o.w = a.w * xv.w + b.w;
                    } else {
                        o.x = (xv.x - b.x) / a.x; o.y = (xv.y - b.y) / a.y;
                        o.z = (xv.z - b.z) / a.z; o.w = (xv.w - b.w) / a.w;
                    }
                    zr[v] = o;
                }
            } else {
                const float *xr = x + row * D + c0;
                float *zr = z + row * D + c0;
                for (int e = lane; e < cn; e += kWave) {
                    const float xv = xr[e], a = alpha_s[e], b = beta_s[e];
                    zr[e] = DIVIDE ? (xv - b) / a : a * xv + b;
                }
            }
        }
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float ld = DIVIDE ? -red[0] : red[0];
    if (lane == 0)
        for (long long row = (long long)blockIdx.x * rows_per_block + wave; row < N; row += stride)
            logdet[row] = accumulate ? logdet[row] + ld : ld;
}

// ---------------------------------------------------------------------------
// Permutation: z[n, j] = x[n, perm[j]] (permutation.py:19-23).  REVERSE_V4: the
// reversal read as mirrored float4s with swapped components (fully coalesced).
// ---------------------------------------------------------------------------
template <bool REVERSE_V4>
__global__ __launch_bounds__(kBlock) void k_permute(
    const float *__restrict__ x, const int *__restrict__ perm, float *__restrict__ z,
    long long N, int D, int G)
{
    const int lane = threadIdx.x & (G - 1);
    const int rows_per_block = kBlock / G;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row = (long long)blockIdx.x * rows_per_block + threadIdx.x / G; row < N;
         row += stride) {
        if (REVERSE_V4) {
            const int Dv = D >> 2;
            const float4 *xr = reinterpret_cast<const float4 *>(x + row * D);
            float4 *zr = reinterpret_cast<float4 *>(z + row * D);
            for (int v = lane; v < Dv; v += G) {
                const float4 s = TFK_EW_LOAD4(xr + (Dv - 1 - v));
                TFK_EW_STORE4(zr + v, make_float4(s.w, s.z, s.y, s.x));
            }
        } else {
            const float *xr = x + row * D;
            float *zr = z + row * D;
            for (int j = lane; j < D; j += G) zr[j] = xr[perm ? perm[j] : D - 1 - j];
        }
    }
}

// ---------------------------------------------------------------------------
// DiagonalGaussian.log_prob (+ log_det).  gaussian.py:46-54, flows.py:647-648.
// Dynamic LDS: 3*D floats (loc | scale | log_scale).
// ---------------------------------------------------------------------------
// FMA: the rows carry a pending per-column map v = s * raw + t (the deferred ActNorm layers of an image program,
// tfk_glow.hip): tfk_rows_fma_gauss_logprob evaluates the density of the mapped values without writing them -- the flush
// pass (read + write of every row) and the density pass (another read) of Flow.log_prob become ONE read.
template <bool VEC4, bool FMA = false>
__global__ __launch_bounds__(kBlock) void k_diag_gauss(
    const float *__restrict__ z, const float *__restrict__ loc,
    const float *__restrict__ log_scale, const float *logdet_in, float *out,
    long long N, int D, int G, const float2 *__restrict__ st = nullptr)
{
    extern __shared__ float smem[];
    float *loc_s = smem, *scale_s = smem + D, *ls_s = smem + 2 * D;
    float *s_s = smem + 3 * D, *t_s = smem + 4 * D;      // (FMA only)
    for (int e = threadIdx.x; e < D; e += kBlock) {
        loc_s[e] = loc[e];
        ls_s[e] = log_scale[e];
        scale_s[e] = expf(log_scale[e]);              // gaussian.py:37-38
        if (FMA) {
            const float2 m = st[e];
            s_s[e] = m.x, t_s[e] = m.y;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & (G - 1);
    const int rows_per_block = kBlock / G;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row = (long long)blockIdx.x * rows_per_block + threadIdx.x / G; row < N;
         row += stride) {
        float acc = 0.0f;
        auto term = [&](float v, int e) {
            if (FMA) v = fmaf(s_s[e], v, t_s[e]);        // (the flush's own rounding: tfk_rows_fma)
            const float t = (v - loc_s[e]) / scale_s[e];
            float q = 0.5f * (t * t);
            q = q + kHalfLog2Pi;
            q = q + ls_s[e];
            acc += -q;                                   // gaussian.py:53-54
        };
        if (VEC4) {
            const int Dv = D >> 2;
            const float4 *zr = reinterpret_cast<const float4 *>(z + row * D);
            for (int v = lane; v < Dv; v += G) {
                const float4 q = TFK_EW_LOAD4(zr + v);          // last reader of the rows
                term(q.x, 4 * v); term(q.y, 4 * v + 1); term(q.z, 4 * v + 2); term(q.w, 4 * v + 3);
            }
        } else {
            const float *zr = z + row * D;
            for (int e = lane; e < D; e += G) term(zr[e], e);
        }
        acc = group_sum(acc, G);
        if (lane == 0) out[row] = logdet_in ? acc + logdet_in[row] : acc;   // flows.py:648
    }
}


// The same for event sizes whose base parameters do not fit the LDS: column tiles of kColTile, one wavefront per row
// and tile, the row's partial sum carried in `out` from tile to tile (the first tile stores, the others add; the
// log-det is added with the last).
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void k_diag_gauss_tiled(
    const float *__restrict__ z, const float *__restrict__ loc,
    const float *__restrict__ log_scale, const float *logdet_in, float *out,
    long long N, int D)
{
    __shared__ __attribute__((aligned(16))) float loc_s[kColTile];
    __shared__ __attribute__((aligned(16))) float scale_s[kColTile];
    __shared__ __attribute__((aligned(16))) float ls_s[kColTile];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    constexpr int rows_per_block = kBlock / kWave;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (int c0 = 0; c0 < D; c0 += kColTile) {
        const int cn = (D - c0 < kColTile) ? D - c0 : kColTile;
        const bool last = c0 + cn >= D;
        __syncthreads();
        for (int e = threadIdx.x; e < cn; e += kBlock) {
            loc_s[e] = loc[c0 + e];
            ls_s[e] = log_scale[c0 + e];
            scale_s[e] = expf(log_scale[c0 + e]);
        }
        __syncthreads();
        for (long long row = (long long)blockIdx.x * rows_per_block + wave; row < N; row += stride) {
            float acc = 0.0f;
            auto term = [&](float v, int e) {
                const float t = (v - loc_s[e]) / scale_s[e];
                float q = 0.5f * (t * t);
                q = q + kHalfLog2Pi;
                q = q + ls_s[e];
                acc += -q;
            };
            if (VEC4) {
                const float4 *zr = reinterpret_cast<const float4 *>(z + row * D + c0);
                for (int v = lane; v < (cn >> 2); v += kWave) {
                    const float4 q = zr[v];
                    term(q.x, 4 * v); term(q.y, 4 * v + 1); term(q.z, 4 * v + 2); term(q.w, 4 * v + 3);
                }
            } else {
                const float *zr = z + row * D + c0;
                for (int e = lane; e < cn; e += kWave) term(zr[e], e);
            }
            acc = group_sum(acc, kWave);
            if (lane == 0) {
                // (an aliased log-det would be overwritten by the first tile's store: it is added there instead)
                const bool aliased = (logdet_in == out);
                float v = c0 == 0 ? acc : out[row] + acc;
                if (logdet_in && (aliased ? c0 == 0 : last)) v = v + logdet_in[row];
                out[row] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// fp64 sum of an fp32 vector: fixed grid, fixed tree => bitwise reproducible.
// ---------------------------------------------------------------------------
constexpr int kSumBlocks = 1024;

__global__ __launch_bounds__(kBlock) void k_sum_partial(const float *__restrict__ in,
                                                        double *partial, long long N)
{
    __shared__ double red[kBlock];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < N;
         i += (long long)gridDim.x * kBlock)
        acc += (double)in[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(kBlock) void k_sum_final(const double *__restrict__ partial,
                                                      double *out, int n)
{
    __shared__ double red[kBlock];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// one workgroup, no scratch memory (tfk_sum_f32): thread t adds in[t], in[t + 1024], ... then a fixed tree
__global__ __launch_bounds__(1024) void k_sum_single(const float *__restrict__ in, double *out, long long N)
{
    __shared__ double red[1024];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < N; i += 1024) acc += (double)in[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
static int check_coupling_args(const char *fn, const void *x, const void *h, const void *z,
                               const void *logdet, bool need_logdet, int64_t N, int32_t D,
                               int32_t T)
{
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0) return fail(TFK_EINVAL, "%s: D = %d must be positive", fn, D);
    if (T <= 0 || T > D) return fail(TFK_EINVAL, "%s: T = %d must be in [1, D = %d]", fn, T, D);
    if (N == 0) return TFK_OK;
    if (!x || !h || !z) return fail(TFK_EINVAL, "%s: null data pointer", fn);
    if (need_logdet && !logdet) return fail(TFK_EINVAL, "%s: null logdet pointer", fn);
    return TFK_OK;
}

// G lanes per row for `units` per-row work items (power of two, <= 64)
static inline int lanes_per_row(int units) {
    int g = pow2_ceil(units < 1 ? 1 : units);
    return g > kWave ? kWave : g;
}

template <int MODE>
static int launch_coupling_generic(const float *x, const float *h, float *z, float *logdet,
                                   int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                                   int32_t accumulate, hipStream_t s, const char *fn)
{
    const bool inplace = (x == z);
    const int G = lanes_per_row(T > 16 ? T : (D < 16 ? D : 16));
    const size_t lds = (tgt_idx && !inplace) ? (size_t)D : 0;
    if (lds > 64 * 1024) return fail(TFK_EINVAL, "%s: masked path supports D <= 65536, got %d", fn, D);
    const int grid = grid_for(N, kBlock / G);
    hipLaunchKernelGGL((k_coupling_generic<MODE>), dim3(grid), dim3(kBlock), lds, s, x, h, z,
                       logdet, (long long)N, D, tgt_idx, T, G, accumulate, inplace ? 1 : 0);
    return check_launch(fn);
}

template <bool INVERSE>
static int affine_coupling(const float *x, const float *h, float *z, float *logdet, int64_t N,
                           int32_t D, const int32_t *tgt_idx, int32_t T, int32_t accumulate,
                           void *stream, const char *fn)
{
    if (int rc = check_coupling_args(fn, x, h, z, logdet, true, N, D, T)) return rc;
    if (N == 0) return TFK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool inplace = (x == z);
    const bool fast = (tgt_idx == nullptr) && (D == 2 * T) && (T % 4 == 0) && aligned16(x) &&
                      aligned16(h) && aligned16(z);
    if (!fast)
        return launch_coupling_generic<INVERSE ? 1 : 0>(x, h, z, logdet, N, D, tgt_idx, T,
                                                        accumulate, s, fn);
    const int Sv = T / 4;
    const int G = lanes_per_row(Sv);
    const int grid = grid_for(N, kBlock / G);
    auto x4 = reinterpret_cast<const float4 *>(x);
    auto h4 = reinterpret_cast<const float4 *>(h);
    auto z4 = reinterpret_cast<float4 *>(z);
    if (inplace)
        hipLaunchKernelGGL((k_affine_half_v4<INVERSE, true>), dim3(grid), dim3(kBlock), 0, s, x4,
                           h4, z4, logdet, (long long)N, Sv, G, accumulate);
    else
        hipLaunchKernelGGL((k_affine_half_v4<INVERSE, false>), dim3(grid), dim3(kBlock), 0, s, x4,
                           h4, z4, logdet, (long long)N, Sv, G, accumulate);
    return check_launch(fn);
}

template <bool DIVIDE>
static int elementwise_affine(const float *x, const float *value, float *z, float *logdet,
                              int64_t N, int32_t D, int32_t accumulate, void *stream,
                              const char *fn)
{
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0) return fail(TFK_EINVAL, "%s: D = %d must be positive", fn, D);
    if (N == 0) return TFK_OK;
    if (!x || !value || !z || !logdet) return fail(TFK_EINVAL, "%s: null pointer", fn);
    const size_t lds = ((size_t)2 * D + kBlock) * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = (D % 4 == 0) && aligned16(x) && aligned16(z);
    if (lds > 64 * 1024) {                    // parameters through the LDS in column tiles (D > 8064)
        const int grid = grid_for(N, kBlock / kWave);
        if (vec)
            hipLaunchKernelGGL((k_elementwise_affine_tiled<DIVIDE, true>), dim3(grid), dim3(kBlock), 0, s,
                               x, value, z, logdet, (long long)N, D, accumulate);
        else
            hipLaunchKernelGGL((k_elementwise_affine_tiled<DIVIDE, false>), dim3(grid), dim3(kBlock), 0, s,
                               x, value, z, logdet, (long long)N, D, accumulate);
        return check_launch(fn);
    }
    const int G = lanes_per_row(vec ? D / 4 : D);
    const int grid = grid_for(N, kBlock / G);
    if (vec)
        hipLaunchKernelGGL((k_elementwise_affine<DIVIDE, true>), dim3(grid), dim3(kBlock), lds, s,
                           x, value, z, logdet, (long long)N, D, G, accumulate);
    else
        hipLaunchKernelGGL((k_elementwise_affine<DIVIDE, false>), dim3(grid), dim3(kBlock), lds, s,
                           x, value, z, logdet, (long long)N, D, G, accumulate);
    return check_launch(fn);
}

}  // namespace tfk

using namespace tfk;

extern "C" {

int tfk_affine_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N,
                            int32_t D, const int32_t *tgt_idx, int32_t T, int32_t accumulate,
                            void *stream)
{
    return affine_coupling<false>(x, h, z, logdet, N, D, tgt_idx, T, accumulate, stream,
                                  "tfk_affine_coupling_fwd");
}

int tfk_affine_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N,
                            int32_t D, const int32_t *tgt_idx, int32_t T, int32_t accumulate,
                            void *stream)
{
    return affine_coupling<true>(z, h, x, logdet, N, D, tgt_idx, T, accumulate, stream,
                                 "tfk_affine_coupling_inv");
}

int tfk_shift_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N,
                           int32_t D, const int32_t *tgt_idx, int32_t T, int32_t accumulate,
                           void *stream)
{
    const char *fn = "tfk_shift_coupling_fwd";
    if (int rc = check_coupling_args(fn, x, h, z, logdet, !accumulate, N, D, T)) return rc;
    if (N == 0) return TFK_OK;
    return launch_coupling_generic<2>(x, h, z, logdet, N, D, tgt_idx, T, accumulate,
                                      static_cast<hipStream_t>(stream), fn);
}

int tfk_shift_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N,
                           int32_t D, const int32_t *tgt_idx, int32_t T, int32_t accumulate,
                           void *stream)
{
    const char *fn = "tfk_shift_coupling_inv";
    if (int rc = check_coupling_args(fn, z, h, x, logdet, !accumulate, N, D, T)) return rc;
    if (N == 0) return TFK_OK;
    return launch_coupling_generic<3>(z, h, x, logdet, N, D, tgt_idx, T, accumulate,
                                      static_cast<hipStream_t>(stream), fn);
}

int tfk_elementwise_affine_fwd(const float *x, const float *value, float *z, float *logdet,
                               int64_t N, int32_t D, int32_t inverse_affine, int32_t accumulate,
                               void *stream)
{
    const char *fn = "tfk_elementwise_affine_fwd";
    return inverse_affine ? elementwise_affine<true>(x, value, z, logdet, N, D, accumulate, stream, fn)
                          : elementwise_affine<false>(x, value, z, logdet, N, D, accumulate, stream, fn);
}

int tfk_elementwise_affine_inv(const float *z, const float *value, float *x, float *logdet,
                               int64_t N, int32_t D, int32_t inverse_affine, int32_t accumulate,
                               void *stream)
{
    const char *fn = "tfk_elementwise_affine_inv";
    return inverse_affine ? elementwise_affine<false>(z, value, x, logdet, N, D, accumulate, stream, fn)
                          : elementwise_affine<true>(z, value, x, logdet, N, D, accumulate, stream, fn);
}

int tfk_permute(const float *x, const int32_t *perm, float *z, int64_t N, int32_t D, void *stream)
{
    const char *fn = "tfk_permute";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0) return fail(TFK_EINVAL, "%s: D = %d must be positive", fn, D);
    if (N == 0) return TFK_OK;
    if (!x || !z) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (x == z) return fail(TFK_EINVAL, "%s: z must not alias x", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool rev4 = (perm == nullptr) && (D % 4 == 0) && aligned16(x) && aligned16(z);
    const int G = lanes_per_row(rev4 ? D / 4 : D);
    const int grid = grid_for(N, kBlock / G);
    if (rev4)
        hipLaunchKernelGGL((k_permute<true>), dim3(grid), dim3(kBlock), 0, s, x, perm, z,
                           (long long)N, D, G);
    else
        hipLaunchKernelGGL((k_permute<false>), dim3(grid), dim3(kBlock), 0, s, x, perm, z,
                           (long long)N, D, G);
    return check_launch(fn);
}

int tfk_diag_gauss_logprob(const float *z, const float *loc, const float *log_scale,
                           const float *logdet_in, float *out, int64_t N, int32_t D, void *stream)
{
    const char *fn = "tfk_diag_gauss_logprob";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0) return fail(TFK_EINVAL, "%s: D = %d must be positive", fn, D);
    if (N == 0) return TFK_OK;
    if (!z || !loc || !log_scale || !out) return fail(TFK_EINVAL, "%s: null pointer", fn);
    const size_t lds = (size_t)3 * D * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = (D % 4 == 0) && aligned16(z);
    if (lds > 64 * 1024) {                    // base parameters through the LDS in column tiles (D > 5461)
        const int grid = grid_for(N, kBlock / kWave);
        if (vec)
            hipLaunchKernelGGL((k_diag_gauss_tiled<true>), dim3(grid), dim3(kBlock), 0, s, z, loc, log_scale,
                               logdet_in, out, (long long)N, D);
        else
            hipLaunchKernelGGL((k_diag_gauss_tiled<false>), dim3(grid), dim3(kBlock), 0, s, z, loc, log_scale,
                               logdet_in, out, (long long)N, D);
        return check_launch(fn);
    }
    const int G = lanes_per_row(vec ? D / 4 : D);
    const int grid = grid_for(N, kBlock / G);
    if (vec)
        hipLaunchKernelGGL((k_diag_gauss<true>), dim3(grid), dim3(kBlock), lds, s, z, loc, log_scale,
                           logdet_in, out, (long long)N, D, G);
    else
        hipLaunchKernelGGL((k_diag_gauss<false>), dim3(grid), dim3(kBlock), lds, s, z, loc, log_scale,
                           logdet_in, out, (long long)N, D, G);
    return check_launch(fn);
}

int tfk_rows_fma_gauss_logprob(const float *rows, const float *st, const float *loc, const float *log_scale,
                               const float *logdet_in, float *out, int64_t N, int32_t D, void *stream)
{
    const char *fn = "tfk_rows_fma_gauss_logprob";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0) return fail(TFK_EINVAL, "%s: D = %d must be positive", fn, D);
    if (N == 0) return TFK_OK;
    if (!rows || !st || !loc || !log_scale || !out) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (reinterpret_cast<uintptr_t>(st) & 7u) return fail(TFK_EINVAL, "%s: st needs 8-byte alignment", fn);
    const size_t lds = (size_t)5 * D * sizeof(float);
    if (lds > 64 * 1024) return fail(TFK_EINVAL, "%s: D = %d (the base parameters and the pending maps take 20 D bytes of LDS: D <= 3276)", fn, D);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = (D % 4 == 0) && aligned16(rows);
    const int G = lanes_per_row(vec ? D / 4 : D);
    const int grid = grid_for(N, kBlock / G);
    const float2 *st2 = reinterpret_cast<const float2 *>(st);
    if (vec)
        hipLaunchKernelGGL((k_diag_gauss<true, true>), dim3(grid), dim3(kBlock), lds, s, rows, loc, log_scale,
                           logdet_in, out, (long long)N, D, G, st2);
    else
        hipLaunchKernelGGL((k_diag_gauss<false, true>), dim3(grid), dim3(kBlock), lds, s, rows, loc, log_scale,
                           logdet_in, out, (long long)N, D, G, st2);
    return check_launch(fn);
}

int64_t tfk_sum_workspace_bytes(int64_t N)
{
    (void)N;
    return (int64_t)kSumBlocks * (int64_t)sizeof(double);
}

int tfk_sum_f32(const float *in, double *out, int64_t N, void *stream)
{
    const char *fn = "tfk_sum_f32";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (!out) return fail(TFK_EINVAL, "%s: null out", fn);
    if (N > 0 && !in) return fail(TFK_EINVAL, "%s: null input", fn);
    hipLaunchKernelGGL(k_sum_single, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), in, out, (long long)N);
    return check_launch(fn);
}

int tfk_sum_f32_ws(const float *in, double *out, void *workspace, int64_t N, void *stream)
{
    const char *fn = "tfk_sum_f32_ws";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (!out || !workspace) return fail(TFK_EINVAL, "%s: null out/workspace", fn);
    if (N > 0 && !in) return fail(TFK_EINVAL, "%s: null input", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int grid = (int)((N + kBlock - 1) / kBlock);
    if (grid < 1) grid = 1;
    if (grid > kSumBlocks) grid = kSumBlocks;
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL(k_sum_partial, dim3(grid), dim3(kBlock), 0, s, in, partial, (long long)N);
    if (int rc = check_launch(fn)) return rc;
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kBlock), 0, s, partial, out, grid);
    return check_launch(fn);
}

}  // extern "C"
