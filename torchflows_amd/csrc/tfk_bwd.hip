// tfk_bwd.hip -- reverse mode of the layer kernels (training on the HIP path).
//
// The reference has no backward code: its gradients are what torch.autograd derives from
//   Affine.forward / inverse                 transformers/linear/affine.py:36-59
//   MonotonicSpline + RationalQuadratic      transformers/spline/base.py:53-72,
//                                            transformers/spline/rational_quadratic.py:45-200
//   ElementwiseBijection (ElementwiseAffine, ActNorm)     layers_base.py:237-318
//   DiagonalGaussian.log_prob                base_distributions/gaussian.py:46-54
// The kernels below are the hand-derived reverse mode of exactly those graphs (same clip
// sub-gradients, same constant boundary knots, identity outside the spline box), checked
// against oracle/oracle.c:orc_*_bwd, which is pinned to the reference's autograd outputs
// (tests/golden/grads*.npz).
//
// Recompute, not store: a backward kernel receives the layer INPUT rows x and the conditioner
// output h and rebuilds alpha / knots / bins itself; nothing from the forward launch is kept.
// Data movement: `g` is the (N, D) gradient row buffer of the whole composition, updated IN
// PLACE -- on entry the target columns hold dL/d(out), on exit dL/d(x_target); the pass-through
// columns are not touched (their conditioner contribution is added by the caller).  gh (N, T, P)
// is written once, coalesced.  All kernels are HBM-bound streams.
#include "tfk_common.h"
#include "tfk_spline.h"

namespace tfk {

// ---------------------------------------------------------------------------------------------
// affine / shift coupling: one lane per target element
//   bytes per element: x 4 + h 8 + g 4 (+4 written) + gh 8 written (+ gld, shared per row)
// ---------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_affine_coupling_bwd(
    const float *__restrict__ x, const float *__restrict__ h, float *g, const float *__restrict__ gld,
    float *__restrict__ gh, long long N, int D, const int *__restrict__ tgt_idx, int T)
{
    const long long total = N * (long long)T;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (long long)gridDim.x * kBlock) {
        const long long row = e / T;
        const int t = (int)(e - row * T);
        const int idx = tgt_idx ? tgt_idx[t] : D - T + t;
        const float2 hh = reinterpret_cast<const float2 *>(h)[e];
        const float ex = exp_noovf(hh.x * 0.5f + kAffC0);           // affine.py:33-34
        const float alpha = ex + kAffMinScale;
        const float ra = __builtin_amdgcn_rcpf(alpha);
        const float gz = g[row * D + idx];
        const float xv = x[row * D + idx];
        const float gl = gld[row];
        float gx, gbeta, galpha;
        if (!INVERSE) {                 // out = alpha x + beta, ld = +sum log alpha
            gx = gz * alpha;
            gbeta = gz;
            galpha = gz * xv + gl * ra;
        } else {                        // out = (x - beta) / alpha, ld = -sum log alpha
            const float r = gz * ra;
            gx = r;
            gbeta = -r;
            galpha = -r * ((xv - hh.y) * ra) - gl * ra;
        }
        g[row * D + idx] = gx;
        reinterpret_cast<float2 *>(gh)[e] = make_float2(galpha * ex * 0.5f, gbeta);
    }
}

// shift coupling (affine.py:137-159): out = x +/- h, log-det 0: gh = +/- g[:, target]
__global__ __launch_bounds__(kBlock) void k_shift_coupling_bwd(
    const float *__restrict__ g, float *__restrict__ gh, long long N, int D,
    const int *__restrict__ tgt_idx, int T, float sign)
{
    const long long total = N * (long long)T;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (long long)gridDim.x * kBlock) {
        const long long row = e / T;
        const int t = (int)(e - row * T);
        const int idx = tgt_idx ? tgt_idx[t] : D - T + t;
        gh[e] = sign * g[row * D + idx];
    }
}

// ---------------------------------------------------------------------------------------------
// RQ-spline coupling.  F(x, theta) = rqs_forward_1d, L(x, theta) = its log-det.
//   forward direction: out = F(v), ld = L(v); upstream (A, B) on (out, ld).
//   inverse direction: out = X with F(X) = v, ld = -L(X).  Implicit differentiation:
//     dX/dv = 1/F_x, dX/dtheta = -F_theta/F_x, so with G = A - B L_x the parameter gradients
//     are those of the forward graph at x = X for the upstream pair (-G/F_x, -B), gv = G/F_x.
//     (If the root was clipped the reference's graph has xi constant: gv = 0 and A reaches
//     only the knots through out = xi w + x_k.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus20_grad(float t) {        // ATen: z = exp(t); z / (z + 1)
    if (t > 20.0f) return 1.0f;
    const float z = exp_noovf(t);
    return z * __builtin_amdgcn_rcpf(z + 1.0f);
}

// p: the element's P = 3K-1 parameters on entry, its parameter gradients on exit.
template <int KT, bool INVERSE>
__device__ __forceinline__ void rqs_bwd_eval(float (&p)[3 * KT - 1], float v, const RqsConst &C,
                                             float A, float B, float &gv)
{
    float smx[KT], smy[KT];
    float mx = 0.0f, my = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const float ux = p[j];
        const float uy = ux + div_1000(p[KT + j]);                  // rational_quadratic.py:76
        smx[j] = ux;
        smy[j] = uy;
        mx = j ? fmaxf(mx, ux) : ux;
        my = j ? fmaxf(my, uy) : uy;
    }
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        smx[j] = exp_noovf(smx[j] - mx);
        smy[j] = exp_noovf(smy[j] - my);
        sx += smx[j];
        sy += smy[j];
    }
    const float rx = div_fast(1.0f, sx), ry = div_fast(1.0f, sy);
    int k = 0;
    float bxk = C.minimum, bxk1 = C.maximum, byk = C.minimum, byk1 = C.maximum;
    float runx = 0.0f, runy = 0.0f, prevx = C.minimum, prevy = C.minimum;
    bool prev_below = true;
#pragma unroll
    for (int j = 1; j <= KT; ++j) {
        smx[j - 1] = smx[j - 1] * rx;                               // softmax, :46
        smy[j - 1] = smy[j - 1] * ry;
        runx = runx + (kRqsMinBin + C.scale * smx[j - 1]);          // :47-48
        runy = runy + (kRqsMinBin + C.scale * smy[j - 1]);
        const float kx = (j == KT) ? C.maximum : C.span * runx + C.minimum;
        const float ky = (j == KT) ? C.maximum : C.span * runy + C.minimum;
        const bool below = (INVERSE ? ky : kx) < v;                 // searchsorted left, :82 / :147
        const bool sel = prev_below && !below;
        k = sel ? j - 1 : k;
        bxk = sel ? prevx : bxk;
        bxk1 = sel ? kx : bxk1;
        byk = sel ? prevy : byk;
        byk1 = sel ? ky : byk1;
        prev_below = below;
        prevx = kx;
        prevy = ky;
    }
    const float wk = bxk1 - bxk, hk = byk1 - byk;
    float udk = C.c, udk1 = C.c;
#pragma unroll
    for (int j = 0; j < KT - 1; ++j) {
        udk = (k == j + 1) ? p[2 * KT + j] : udk;
        udk1 = (k == j) ? p[2 * KT + j] : udk1;
    }
    const float tdk = C.c + div_1000(udk), tdk1 = C.c + div_1000(udk1);
    const float dk = kRqsMinDelta + softplus20(tdk);                // :77
    const float dk1 = kRqsMinDelta + softplus20(tdk1);
    const float rw = __builtin_amdgcn_rcpf(wk);
    const float s = div_fast(hk, wk);
    const float term1 = dk1 + dk - 2.0f * s;

    float xi_raw;
    if (!INVERSE) {
        xi_raw = div_fast(v - bxk, wk);                             // :99
    } else {                                                        // :164-173
        const float term0 = v - byk;
        const float term2 = hk * dk;
        const float a = (hk * s - term2) + term0 * term1;
        const float b = term2 - term0 * term1;
        const float c = (-s) * term0;
        float r = sqrtf(b * b - (4.0f * a) * c);
        r = r < 0.0f ? 0.0f : r;
        xi_raw = div_fast(2.0f * c, (-b) - r);
    }
    const bool pass = (xi_raw >= 0.0f) && (xi_raw <= 1.0f);         // torch.clip sub-gradient
    const float xi = clip01(xi_raw);
    const float omx = 1.0f - xi;
    const float q = xi * omx;
    const float inner2 = s * (xi * xi) + dk * q;
    const float num0 = hk * inner2;
    const float den0 = s + term1 * q;
    const float rden = __builtin_amdgcn_rcpf(den0);
    const float inner = dk1 * (xi * xi) + (2.0f * s) * q + dk * (omx * omx);
    const float rinner = __builtin_amdgcn_rcpf(inner);
    const float dq = 1.0f - 2.0f * xi;

    float g_wk = 0.0f, g_bx = 0.0f;
    if (!INVERSE) {
        // gv = A F_x + B L_x falls out of the sweep below (through g_xi)
    } else {
        const float F_xi = (hk * (2.0f * s * xi + dk * dq)) * rden - (num0 * rden * rden) * (term1 * dq);
        const float L_xi = (2.0f * dk1 * xi + 2.0f * s * dq - 2.0f * dk * omx) * rinner
                           - 2.0f * (term1 * dq) * rden;
        if (pass) {
            const float F_x = F_xi * rw, L_x = L_xi * rw;
            const float G = A - B * L_x;                            // d loss / d X (ld_inv = -L(X))
            gv = div_fast(G, F_x);
            A = -gv;
        } else {                                                    // xi is a constant of the graph
            gv = 0.0f;
            g_wk = A * xi;                                          // out = xi w + x_k, :178
            g_bx = A;
            A = 0.0f;
        }
        B = -B;
    }

    // reverse sweep of the forward graph (:94-109, :56-63) for upstream (A, B)
    float g_s = 0.0f, g_q = 0.0f, g_xi = 0.0f, g_d0 = 0.0f, g_d1 = 0.0f, g_hk = 0.0f;
    const float g_by = A;
    const float g_num0 = A * rden;
    float g_den0 = -A * num0 * rden * rden;
    g_s += B * 2.0f * __builtin_amdgcn_rcpf(s);
    const float g_inner = B * rinner;
    g_den0 += -2.0f * B * rden;
    g_d1 += g_inner * (xi * xi);
    g_s += g_inner * 2.0f * q;
    g_q += g_inner * 2.0f * s;
    g_d0 += g_inner * (omx * omx);
    g_xi += g_inner * (2.0f * dk1 * xi - 2.0f * dk * omx);
    g_s += g_den0;
    const float g_t1 = g_den0 * q;
    g_q += g_den0 * term1;
    g_hk += g_num0 * inner2;
    const float g_in2 = g_num0 * hk;
    g_s += g_in2 * (xi * xi);
    g_xi += g_in2 * 2.0f * s * xi;
    g_d0 += g_in2 * q;
    g_q += g_in2 * dk;
    g_d1 += g_t1;
    g_d0 += g_t1;
    g_s -= 2.0f * g_t1;
    g_xi += g_q * dq;
    if (pass) {                                               // xi = (x - x_k) / w
        if (!INVERSE) gv = g_xi * rw;
        g_bx -= g_xi * rw;
        g_wk -= g_xi * xi_raw * rw;
    } else if (!INVERSE) {
        gv = 0.0f;
    }
    g_hk += g_s * rw;                                               // s = h / w
    g_wk -= g_s * s * rw;
    // w = x_{k+1} - x_k, h = y_{k+1} - y_k; knots 0 and K are constants (:51-52)
    const float g_x0 = g_bx - g_wk, g_x1 = g_wk;
    const float g_y0 = g_by - g_hk, g_y1 = g_hk;
    const float sc = C.scale * C.span;
    const float gcx0 = (k >= 1) ? sc * g_x0 : 0.0f, gcx1 = (k + 1 <= KT - 1) ? sc * g_x1 : 0.0f;
    const float gcy0 = (k >= 1) ? sc * g_y0 : 0.0f, gcy1 = (k + 1 <= KT - 1) ? sc * g_y1 : 0.0f;
    float dotx = 0.0f, doty = 0.0f;
#pragma unroll
    for (int i = 0; i < KT; ++i) {                                  // cumsum: knot j sums bins i < j
        const float gwx = (i < k ? gcx0 : 0.0f) + (i < k + 1 ? gcx1 : 0.0f);
        const float gwy = (i < k ? gcy0 : 0.0f) + (i < k + 1 ? gcy1 : 0.0f);
        dotx += smx[i] * gwx;
        doty += smy[i] * gwy;
    }
#pragma unroll
    for (int i = 0; i < KT; ++i) {                                  // softmax backward
        const float gwx = (i < k ? gcx0 : 0.0f) + (i < k + 1 ? gcx1 : 0.0f);
        const float gwy = (i < k ? gcy0 : 0.0f) + (i < k + 1 ? gcy1 : 0.0f);
        const float gux = smx[i] * (gwx - dotx);
        const float guy = smy[i] * (gwy - doty);
        p[i] = gux + guy;                                           // u_y enters as u_x + u_y / 1000
        p[KT + i] = div_1000(guy);
    }
    const float gud0 = div_1000(g_d0 * softplus20_grad(tdk));
    const float gud1 = div_1000(g_d1 * softplus20_grad(tdk1));
#pragma unroll
    for (int j = 0; j < KT - 1; ++j)
        p[2 * KT + j] = (k == j + 1) ? gud0 : ((k == j) ? gud1 : 0.0f);
}

// Flat map over the N*T spline elements; a workgroup takes 256 consecutive elements, i.e. 256
// consecutive parameter records (256*P floats, 16-byte aligned): coalesced float4 loads into
// LDS, each lane reads its own record at a P-dword stride (conflict-free for odd P), the
// gradient record goes back through the same LDS slot and out with coalesced float4 stores.
//   bytes per element: h 4P + gh 4P + x 4 + g 8 (+ gld per row)   (K = 8: 196 B)
template <int KT, bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_rqs_coupling_bwd(
    const float *__restrict__ x, const float *__restrict__ h, float *g, const float *__restrict__ gld,
    float *__restrict__ gh, long long N, int D, const int *__restrict__ tgt_idx, int T, RqsConst C)
{
    constexpr int P = 3 * KT - 1;
    __shared__ __attribute__((aligned(16))) float rec[kBlock * P];
    const int tid = threadIdx.x;
    const long long total = N * (long long)T;
    const long long n_tiles = (total + kBlock - 1) / kBlock;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long e0 = tile * kBlock;
        const int E = (int)((total - e0) < (long long)kBlock ? (total - e0) : (long long)kBlock);
        const int nv = (E * P) >> 2, nfl = E * P;
        const float4 *src = reinterpret_cast<const float4 *>(h + e0 * P);
        float4 *dst = reinterpret_cast<float4 *>(rec);
        __syncthreads();                       // previous tile's stores have read rec
        for (int i = tid; i < nv; i += kBlock) dst[i] = nt_load4(src + i);          // h: touched once per pass
        for (int i = (nv << 2) + tid; i < nfl; i += kBlock) rec[i] = nt_load(h + e0 * P + i);
        __syncthreads();
        if (tid < E) {
            const long long e = e0 + tid;
            const long long row = e / T;
            const int t = (int)(e - row * T);
            const int idx = tgt_idx ? tgt_idx[t] : D - T + t;
            const float v = x[row * D + idx];
            const float A = g[row * D + idx];
            float p[P];
#pragma unroll
            for (int j = 0; j < P; ++j) p[j] = rec[tid * P + j];
            float gv = A;                      // identity outside the box, spline/base.py:54-55
            if (v > C.minimum && v < C.maximum) {
                rqs_bwd_eval<KT, INVERSE>(p, v, C, A, gld[row], gv);
            } else {
#pragma unroll
                for (int j = 0; j < P; ++j) p[j] = 0.0f;
            }
            g[row * D + idx] = gv;
#pragma unroll
            for (int j = 0; j < P; ++j) rec[tid * P + j] = p[j];
        }
        __syncthreads();
        float4 *out = reinterpret_cast<float4 *>(gh + e0 * P);
        for (int i = tid; i < nv; i += kBlock) nt_store4(out + i, dst[i]);         // 2.9 KB per row: no cache holds it
        for (int i = (nv << 2) + tid; i < nfl; i += kBlock) gh[e0 * P + i] = rec[i];
    }
}

// ---------------------------------------------------------------------------------------------
// ElementwiseAffine / ActNorm with batch-constant parameters value (D, 2):
//   g <- g * alpha (forward form) or g / alpha (inverse form), in place;
//   gvalue[d] = sum over rows of (d loss / d u_alpha, d loss / d beta)  -- two-stage, deterministic:
//   a workgroup owns a contiguous slab of rows, thread (ty, tx) walks rows ty, ty + RY, ... of
//   the slab at columns tx, tx + CW, ...; partials are reduced over ty through LDS and written to
//   part[block][2 D]; k_colsum_final adds the blocks in index order.
//   bytes per row: g 8 D (+ x 4 D when the parameter gradient is wanted)
// ---------------------------------------------------------------------------------------------
template <int V> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<4> { typedef float4 type; };
__device__ __forceinline__ void unpack(float v, float (&a)[1]) { a[0] = v; }
__device__ __forceinline__ void unpack(float4 v, float (&a)[4]) { a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
__device__ __forceinline__ float pack(const float (&a)[1]) { return a[0]; }
__device__ __forceinline__ float4 pack(const float (&a)[4]) { return make_float4(a[0], a[1], a[2], a[3]); }

// V = 4: a thread owns four consecutive columns (16-byte loads / stores, 1 KiB per
// wave-instruction; D % 4 == 0); V = 1: any D.  CW = column groups walked side by side
// (a power of two <= 256), RY = 256 / CW rows in flight per workgroup.
template <bool INVERSE, bool WANT_PARAM, int V>
__global__ __launch_bounds__(kBlock) void k_elementwise_affine_bwd(
    const float *__restrict__ x, const float *__restrict__ value, float *g,
    const float *__restrict__ gld, float *__restrict__ part, long long N, int D, int CW,
    long long rows_per_block)
{
    typedef typename VecT<V>::type vec;
    __shared__ float red[WANT_PARAM ? 2 * V * kBlock : 1];
    const int tid = threadIdx.x;
    const int tx = tid % CW, ty = tid / CW, RY = kBlock / CW;
    const int G = D / V;                                            // column groups per row
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int c0 = 0; c0 < G; c0 += CW) {
        const int cg = c0 + tx;
        float sa[V], sb[V], ex[V];
#pragma unroll
        for (int i = 0; i < V; ++i) sa[i] = sb[i] = ex[i] = 0.0f;
        if (cg < G) {
            float alpha[V], ra[V], beta[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int c = cg * V + i;
                beta[i] = value[2 * c + 1];
                ex[i] = exp_noovf(value[2 * c] * 0.5f + kAffC0);
                alpha[i] = ex[i] + kAffMinScale;
                ra[i] = __builtin_amdgcn_rcpf(alpha[i]);
            }
            vec *gp = reinterpret_cast<vec *>(g);
            const vec *xp = reinterpret_cast<const vec *>(x);
#pragma unroll 4
            for (long long row = r0 + ty; row < r1; row += RY) {
                float gz[V], xv[V];
                unpack(gp[row * G + cg], gz);
                float gl = 0.0f;
                if (WANT_PARAM) {
                    unpack(xp[row * G + cg], xv);
                    gl = gld[row];
                }
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    if (!INVERSE) {
                        if (WANT_PARAM) {
                            sb[i] += gz[i];
                            sa[i] += gz[i] * xv[i] + gl * ra[i];
                        }
                        gz[i] = gz[i] * alpha[i];
                    } else {
                        const float r = gz[i] * ra[i];
                        if (WANT_PARAM) {
                            sb[i] += -r;
                            sa[i] += -r * ((xv[i] - beta[i]) * ra[i]) - gl * ra[i];
                        }
                        gz[i] = r;
                    }
                }
                gp[row * G + cg] = pack(gz);
            }
        }
        if (WANT_PARAM) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < V; ++i) {
                red[(2 * i) * kBlock + tid] = sa[i] * ex[i] * 0.5f;     // d alpha / d u
                red[(2 * i + 1) * kBlock + tid] = sb[i];
            }
            __syncthreads();
            if (ty == 0 && cg < G) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float ta = 0.0f, tb = 0.0f;
                    for (int j = 0; j < RY; ++j) {
                        ta += red[(2 * i) * kBlock + j * CW + tx];
                        tb += red[(2 * i + 1) * kBlock + j * CW + tx];
                    }
                    const int c = cg * V + i;
                    part[(long long)blockIdx.x * 2 * D + 2 * c] = ta;
                    part[(long long)blockIdx.x * 2 * D + 2 * c + 1] = tb;
                }
            }
        }
    }
}

// one workgroup per column: threads stride over the per-block partials, then a fixed-order
// LDS tree (deterministic)
__global__ __launch_bounds__(kBlock) void k_colsum_final(const float *__restrict__ part,
                                                         float *__restrict__ out, int n_blocks, int M)
{
    __shared__ float red[kBlock];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int b = threadIdx.x; b < n_blocks; b += kBlock) s += part[(long long)b * M + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

// DiagonalGaussian.log_prob backward (gaussian.py:46-54): g[row, d] = -glp[row] (z - loc) / scale^2
__global__ __launch_bounds__(kBlock) void k_diag_gauss_bwd(
    const float *__restrict__ z, const float *__restrict__ loc, const float *__restrict__ log_scale,
    const float *__restrict__ glp, float *__restrict__ g, long long N, int D)
{
    const long long total = N * (long long)D;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (long long)gridDim.x * kBlock) {
        const long long row = e / D;
        const int d = (int)(e - row * D);
        const float sc = exp_noovf(log_scale[d]);
        const float t = div_fast(z[e] - loc[d], sc);
        g[e] = -glp[row] * div_fast(t, sc);
    }
}

static int check_common(const char *fn, int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T)
{
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1) return fail(TFK_EINVAL, "%s: D = %d < 1", fn, D);
    if (T < 1 || T > D) return fail(TFK_EINVAL, "%s: T = %d must be in [1, D = %d]", fn, T, D);
    (void)tgt_idx;
    return TFK_OK;
}

template <int KT>
static int launch_rqs_bwd(const float *x, const float *h, float *g, const float *gld, float *gh,
                          int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T, RqsConst C,
                          int inverse, hipStream_t s, const char *fn)
{
    const int64_t tiles = (N * (int64_t)T + kBlock - 1) / kBlock;
    const int grid = (int)(tiles < max_grid() ? tiles : max_grid());
    if (inverse)
        hipLaunchKernelGGL((k_rqs_coupling_bwd<KT, true>), dim3(grid), dim3(kBlock), 0, s, x, h, g, gld,
                           gh, (long long)N, D, tgt_idx, T, C);
    else
        hipLaunchKernelGGL((k_rqs_coupling_bwd<KT, false>), dim3(grid), dim3(kBlock), 0, s, x, h, g, gld,
                           gh, (long long)N, D, tgt_idx, T, C);
    return check_launch(fn);
}

}  // namespace tfk

using namespace tfk;

extern "C" {

int tfk_affine_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh,
                            int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T, int32_t inverse,
                            void *stream)
{
    const char *fn = "tfk_affine_coupling_bwd";
    if (int rc = check_common(fn, N, D, tgt_idx, T)) return rc;
    if (N == 0) return TFK_OK;
    if (!x || !h || !g || !gld || !gh) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if ((reinterpret_cast<uintptr_t>(h) & 7u) || (reinterpret_cast<uintptr_t>(gh) & 7u))
        return fail(TFK_EINVAL, "%s: h and gh must be 8-byte aligned", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = grid_for(N * (int64_t)T, kBlock);
    if (inverse)
        hipLaunchKernelGGL((k_affine_coupling_bwd<true>), dim3(grid), dim3(kBlock), 0, s, x, h, g, gld, gh,
                           (long long)N, D, tgt_idx, T);
    else
        hipLaunchKernelGGL((k_affine_coupling_bwd<false>), dim3(grid), dim3(kBlock), 0, s, x, h, g, gld, gh,
                           (long long)N, D, tgt_idx, T);
    return check_launch(fn);
}

int tfk_shift_coupling_bwd(const float *g, float *gh, int64_t N, int32_t D, const int32_t *tgt_idx,
                           int32_t T, int32_t inverse, void *stream)
{
    const char *fn = "tfk_shift_coupling_bwd";
    if (int rc = check_common(fn, N, D, tgt_idx, T)) return rc;
    if (N == 0) return TFK_OK;
    if (!g || !gh) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_shift_coupling_bwd, dim3(grid_for(N * (int64_t)T, kBlock)), dim3(kBlock), 0, s, g,
                       gh, (long long)N, D, tgt_idx, T, inverse ? -1.0f : 1.0f);
    return check_launch(fn);
}

int tfk_rqs_coupling_bwd_supported(int32_t n_bins) { return (n_bins == 4 || n_bins == 8 || n_bins == 16) ? 1 : 0; }

int tfk_rqs_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh,
                         int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T, int32_t n_bins,
                         float boundary, int32_t inverse, void *stream)
{
    const char *fn = "tfk_rqs_coupling_bwd";
    if (int rc = check_common(fn, N, D, tgt_idx, T)) return rc;
    if (!tfk_rqs_coupling_bwd_supported(n_bins))
        return fail(TFK_EINVAL, "%s: n_bins = %d (backward kernels exist for 4, 8, 16)", fn, n_bins);
    if (!(boundary > 0.0f)) return fail(TFK_EINVAL, "%s: boundary must be positive", fn);
    if (N == 0) return TFK_OK;
    if (!x || !h || !g || !gld || !gh) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!aligned16(h) || !aligned16(gh)) return fail(TFK_EINVAL, "%s: h and gh must be 16-byte aligned", fn);
    RqsConst C;
    C.minimum = -boundary;
    C.maximum = boundary;
    C.span = (float)((double)boundary + (double)boundary);
    C.scale = (float)(1.0 - 1e-3 * (double)n_bins);
    C.c = (float)log(expm1(1.0 - 1e-5));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_bins == 8) return launch_rqs_bwd<8>(x, h, g, gld, gh, N, D, tgt_idx, T, C, inverse, s, fn);
    if (n_bins == 4) return launch_rqs_bwd<4>(x, h, g, gld, gh, N, D, tgt_idx, T, C, inverse, s, fn);
    return launch_rqs_bwd<16>(x, h, g, gld, gh, N, D, tgt_idx, T, C, inverse, s, fn);
}

int64_t tfk_elementwise_affine_bwd_workspace_bytes(int64_t N, int32_t D)
{
    if (N <= 0 || D <= 0) return 0;
    return (int64_t)max_grid() * 2 * D * (int64_t)sizeof(float);
}

int tfk_elementwise_affine_bwd(const float *x, const float *value, float *g, const float *gld,
                               float *gvalue, float *workspace, int64_t N, int32_t D,
                               int32_t inverse, void *stream)
{
    const char *fn = "tfk_elementwise_affine_bwd";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1) return fail(TFK_EINVAL, "%s: D = %d < 1", fn, D);
    if (!value || (N > 0 && !g)) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (gvalue && N > 0 && (!x || !gld || !workspace))
        return fail(TFK_EINVAL, "%s: the parameter gradient needs x, gld and a workspace", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (gvalue) {
            hipError_t e = hipMemsetAsync(gvalue, 0, (size_t)2 * D * sizeof(float), s);
            if (e != hipSuccess) return fail(TFK_ELAUNCH, "%s: memset: %s", fn, hipGetErrorString(e));
        }
        return TFK_OK;
    }
    const bool vec4 = (D % 4 == 0) && aligned16(g) && (!gvalue || aligned16(x));
    const int G = vec4 ? D / 4 : D;
    int CW = pow2_ceil(G);
    if (CW > kBlock) CW = kBlock;
    const int RY = kBlock / CW;
    // slabs of rows: enough workgroups to fill the chip, at least RY * 8 rows each
    int64_t blocks = max_grid();
    int64_t rpb = (N + blocks - 1) / blocks;
    const int64_t min_rows = (int64_t)RY * 8;
    if (rpb < min_rows) rpb = min_rows;
    blocks = (N + rpb - 1) / rpb;
    const int grid = (int)blocks;
#define TFK_EW_BWD(INV, WANT, V)                                                                          \
    hipLaunchKernelGGL((k_elementwise_affine_bwd<INV, WANT, V>), dim3(grid), dim3(kBlock), 0, s, x, value, g, \
                       gld, workspace, (long long)N, D, CW, (long long)rpb)
    if (gvalue) {
        if (vec4) { if (inverse) TFK_EW_BWD(true, true, 4); else TFK_EW_BWD(false, true, 4); }
        else      { if (inverse) TFK_EW_BWD(true, true, 1); else TFK_EW_BWD(false, true, 1); }
        if (int rc = check_launch(fn)) return rc;
        hipLaunchKernelGGL(k_colsum_final, dim3(2 * D), dim3(kBlock), 0, s, workspace,
                           gvalue, grid, 2 * D);
    } else {
        if (vec4) { if (inverse) TFK_EW_BWD(true, false, 4); else TFK_EW_BWD(false, false, 4); }
        else      { if (inverse) TFK_EW_BWD(true, false, 1); else TFK_EW_BWD(false, false, 1); }
    }
#undef TFK_EW_BWD
    return check_launch(fn);
}

int tfk_diag_gauss_logprob_bwd(const float *z, const float *loc, const float *log_scale,
                               const float *glp, float *g, int64_t N, int32_t D, void *stream)
{
    const char *fn = "tfk_diag_gauss_logprob_bwd";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1) return fail(TFK_EINVAL, "%s: D = %d < 1", fn, D);
    if (N == 0) return TFK_OK;
    if (!z || !loc || !log_scale || !glp || !g) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_diag_gauss_bwd, dim3(grid_for(N * (int64_t)D, kBlock)), dim3(kBlock), 0, s, z, loc,
                       log_scale, glp, g, (long long)N, D);
    return check_launch(fn);
}

}  // extern "C"

// =============================================================================================
// Fused training backward of one affine coupling layer (HalfSplit: source = first half of the
// row, target = second half; FeedForward(Linear, Tanh, Linear) conditioner, hidden width <= 15).
// ONE launch does what the layer-by-layer route does in ~15: re-evaluate the conditioner, reverse
// the transform, reverse the MLP (dL/dx_A) and reduce the weight gradients over the batch rows --
// h, dL/dh and the hidden activations never exist in HBM.
//   bytes per row: x 4D + g 8D + 4 (gld)            (D = 64: 772 B; the split route moves ~3 KB)
// Register layout = the forward flow program's (tfk_flow_mfma.hip): a wave owns 16 rows, lane
// (q, j) holds elements [EPL q, EPL (q+1)) of the source half and of the target half of row j.
//   1. GEMM 1 + tanh, GEMM 2 -> (u, beta) of this lane's target elements       (as the forward)
//   2. transform backward in registers -> dL/dx_B, dL/dh (same D-layout as h)
//   3. dL/dhidden = W2^T dL/dh  : MFMA, A-operand = W2 packed so the result lands on the lane /
//      register that holds the matching hidden unit; times tanh'
//   4. dL/dx_A   = W1^T dL/dpre : MFMA, result lands on the lane that holds the element
//   5. dW2 += dL/dh^T hidden, dW1 += dL/dpre^T x_A: the contraction runs over the 16 ROWS of the
//      wave, i.e. over the lane index that MFMA never contracts -- both operands take one trip
//      through a wave-private LDS tile (written row-major, read transposed), accumulators stay
//      in registers for the whole row loop; db2 rides along as hidden unit 15 == 1.
//   6. per-workgroup partial sums -> part[block][M]; tfk reduces them with k_colsum2d.
// Parameter block (floats, packed by torchflows_amd/autograd.py:_TrainPack):
//   A1[EPL][64] | b1[16] | A2[T2][steps2][64] | b2[T2][16] | A2T[T2][4][64] | A1T[EPL/4][4][64]
// Partial layout (M floats): dW2[T2][64][4] | dW1[EPL/4][64][4] | db1[16]
// =============================================================================================
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int EPL, bool INVFORM>
__global__ __launch_bounds__(kBlock) void k_affine_coupling_train_bwd(
    const float *__restrict__ x, float *g, const float *__restrict__ gld,
    const float *__restrict__ params, int n_params, int steps2, float *__restrict__ part, long long N,
    const float *__restrict__ gscale, int g_reversed)
{
    constexpr int D = 8 * EPL, HALF = 4 * EPL, T2 = EPL / 2, T1 = EPL / 4;
    // wave-private transpose tiles.  Strides are padded so that the transposed ds_read_b32 of a
    // half-wave touch 32 different banks: blocks of 16 rows x 4 floats every 72 floats (bank =
    // 8*block + 4*q + r), x rows every HALF + 16 floats (bank = 16*q + column).
    constexpr int BLK = 72, XROW = HALF + 16;
    // D = 256 (round 4): the dL/dh tile is staged in two halves of T2 / 2 tiles (16 tiles at once would take the LDS to
    // 166 KB with four waves per workgroup)
    constexpr int GH = (EPL == 32) ? 2 : 1, TG = T2 / GH;
    constexpr int SCR_H = 4 * BLK, SCR_G = TG * 4 * BLK, SCR_X = 16 * XROW;
    constexpr int M = T2 * 256 + T1 * 256 + 16;
    constexpr int SCR = (2 * SCR_H + SCR_G + SCR_X) > M ? (2 * SCR_H + SCR_G + SCR_X) : M;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const float4 *src = reinterpret_cast<const float4 *>(params);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < (n_params >> 2); i += kBlock) dst[i] = src[i];
    }
    __syncthreads();
    const float *A1 = lds;
    const float *b1 = A1 + EPL * 64;
    const float *A2 = b1 + 16;
    const float *b2 = A2 + T2 * steps2 * 64;
    const float *A2T = b2 + T2 * 16;
    const float *A1T = A2T + T2 * 4 * 64;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    float *scr = lds + n_params + wave * SCR;
    float *scr_h = scr;                                  // [q][row][r]   hidden (unit 4r+q)
    float *scr_p = scr + SCR_H;                          // [q][row][r]   dL/dpre
    float *scr_g = scr + 2 * SCR_H;                      // [t][q][row][r] dL/dh
    float *scr_x = scr + 2 * SCR_H + SCR_G;              // [row][HALF]   x_A

    f32x4_t accW2[T2], accW1[T1];
#pragma unroll
    for (int t = 0; t < T2; ++t) accW2[t] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < T1; ++t) accW1[t] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
    float sb1[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    constexpr int rows_per_block = (kBlock / 64) * 16;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row0 = (long long)blockIdx.x * rows_per_block + wave * 16; row0 < N; row0 += stride) {
        const long long row = row0 + j;
        const bool valid = row < N;
        const long long rr = valid ? row : N - 1;
        float xa[EPL], xb[EPL], ga[EPL], gb[EPL];
        {
            const float4 *pa = reinterpret_cast<const float4 *>(x + rr * D + EPL * q);
            const float4 *pb = reinterpret_cast<const float4 *>(x + rr * D + HALF + EPL * q);
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                const float4 va = pa[i], vb = pb[i];
                xa[4 * i] = va.x; xa[4 * i + 1] = va.y; xa[4 * i + 2] = va.z; xa[4 * i + 3] = va.w;
                xb[4 * i] = vb.x; xb[4 * i + 1] = vb.y; xb[4 * i + 2] = vb.z; xb[4 * i + 3] = vb.w;
            }
            if (!g_reversed) {
                const float4 *qa = reinterpret_cast<const float4 *>(g + rr * D + EPL * q);
                const float4 *qb = reinterpret_cast<const float4 *>(g + rr * D + HALF + EPL * q);
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    const float4 wa = qa[i], wb = qb[i];
                    ga[4 * i] = wa.x; ga[4 * i + 1] = wa.y; ga[4 * i + 2] = wa.z; ga[4 * i + 3] = wa.w;
                    gb[4 * i] = wb.x; gb[4 * i + 1] = wb.y; gb[4 * i + 2] = wb.z; gb[4 * i + 3] = wb.w;
                }
            } else {        // column c of the logical row sits at D-1-c (a reversal followed the layer)
                const float4 *qa = reinterpret_cast<const float4 *>(g + rr * D + D - EPL * (q + 1));
                const float4 *qb = reinterpret_cast<const float4 *>(g + rr * D + HALF - EPL * (q + 1));
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    const float4 wa = qa[i], wb = qb[i];
                    ga[EPL - 1 - 4 * i] = wa.x; ga[EPL - 2 - 4 * i] = wa.y; ga[EPL - 3 - 4 * i] = wa.z; ga[EPL - 4 - 4 * i] = wa.w;
                    gb[EPL - 1 - 4 * i] = wb.x; gb[EPL - 2 - 4 * i] = wb.y; gb[EPL - 3 - 4 * i] = wb.z; gb[EPL - 4 - 4 * i] = wb.w;
                }
            }
            if (gscale) {   // reverse mode of a fixed elementwise scale that followed the layer
                const float4 *sa = reinterpret_cast<const float4 *>(gscale + EPL * q);
                const float4 *sb = reinterpret_cast<const float4 *>(gscale + HALF + EPL * q);
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    const float4 ua = sa[i], ub = sb[i];
                    ga[4 * i] *= ua.x; ga[4 * i + 1] *= ua.y; ga[4 * i + 2] *= ua.z; ga[4 * i + 3] *= ua.w;
                    gb[4 * i] *= ub.x; gb[4 * i + 1] *= ub.y; gb[4 * i + 2] *= ub.z; gb[4 * i + 3] *= ub.w;
                }
            }
        }
        float gl = gld[rr];
        if (!valid) {                       // padding rows of the last tile contribute nothing
            gl = 0.0f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) ga[e] = gb[e] = 0.0f;
        }

        // 1. conditioner forward (transforms.py:293-304)
        f32x4_t acc = *reinterpret_cast<const f32x4_t *>(b1 + 4 * q);
#pragma unroll
        for (int s = 0; s < EPL; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[s * 64 + lane], xa[s], acc, 0, 0, 0);
        float hid[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[r] = tanh_fast(acc[r]);

        // 2. h tile by tile, transform backward (affine.py:36-59)
        f32x4_t ghv[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            f32x4_t o = *reinterpret_cast<const f32x4_t *>(b2 + (t * 4 + q) * 4);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2) * 64 + lane], hid[0], o, 0, 0, 0);
            if (steps2 > 1) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2 + 1) * 64 + lane], hid[1], o, 0, 0, 0);
            if (steps2 > 2) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2 + 2) * 64 + lane], hid[2], o, 0, 0, 0);
            if (steps2 > 3) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2 + 3) * 64 + lane], hid[3], o, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 2 * t + i;
                const float beta = o[2 * i + 1];
                const float ex = exp_noovf(o[2 * i] * 0.5f + kAffC0);
                const float alpha = ex + kAffMinScale;
                const float ra = __builtin_amdgcn_rcpf(alpha);
                const float gz = gb[e];
                float gx, gbeta, galpha;
                if (!INVFORM) {
                    gx = gz * alpha;
                    gbeta = gz;
                    galpha = gz * xb[e] + gl * ra;
                } else {
                    const float r_ = gz * ra;
                    gx = r_;
                    gbeta = -r_;
                    galpha = -r_ * ((xb[e] - beta) * ra) - gl * ra;
                }
                gb[e] = gx;
                ghv[t][2 * i] = galpha * ex * 0.5f;
                ghv[t][2 * i + 1] = gbeta;
            }
        }

        // 3. dL/dhidden (unit 4r+q lands in register r), times tanh'
        f32x4_t gha = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gha = __builtin_amdgcn_mfma_f32_16x16x4f32(A2T[(t * 4 + r) * 64 + lane], ghv[t][r], gha, 0, 0, 0);
        float gpre[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gpre[r] = gha[r] * (1.0f - hid[r] * hid[r]);
            sb1[r] += gpre[r];
        }

        // 4. dL/dx_A: element EPL q + 4 t + r' lands in register r' of tile t
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            f32x4_t d = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(A1T[(t * 4 + r) * 64 + lane], gpre[r], d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[4 * t + r] += d[r];
        }
        if (valid) {
            float4 *qa = reinterpret_cast<float4 *>(g + row * D + EPL * q);
            float4 *qb = reinterpret_cast<float4 *>(g + row * D + HALF + EPL * q);
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                qa[i] = make_float4(ga[4 * i], ga[4 * i + 1], ga[4 * i + 2], ga[4 * i + 3]);
                qb[i] = make_float4(gb[4 * i], gb[4 * i + 1], gb[4 * i + 2], gb[4 * i + 3]);
            }
        }

        // 5. weight gradients: both operands through the wave-private LDS tile (row-major in,
        //    transposed out); hidden unit 15 (r = 3 of lane-group q = 3) is the constant 1 -> db2
        wave_lds_sync();                    // the previous iteration's reads are done
        *reinterpret_cast<float4 *>(scr_h + q * BLK + j * 4) =
            make_float4(hid[0], hid[1], hid[2], q == 3 ? 1.0f : hid[3]);
        *reinterpret_cast<float4 *>(scr_p + q * BLK + j * 4) = make_float4(gpre[0], gpre[1], gpre[2], gpre[3]);
#pragma unroll
        for (int t = 0; t < TG; ++t)
            *reinterpret_cast<float4 *>(scr_g + (t * 4 + q) * BLK + j * 4) =
                make_float4(ghv[t][0], ghv[t][1], ghv[t][2], ghv[t][3]);
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i)
            *reinterpret_cast<float4 *>(scr_x + j * XROW + EPL * q + 4 * i) =
                make_float4(xa[4 * i], xa[4 * i + 1], xa[4 * i + 2], xa[4 * i + 3]);
        wave_lds_sync();
        float bh[4], bp[4];                 // B-operands: [k = row 4s+q][column = unit j]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bh[s] = scr_h[(j & 3) * BLK + (4 * s + q) * 4 + (j >> 2)];
            bp[s] = scr_p[(j & 3) * BLK + (4 * s + q) * 4 + (j >> 2)];
        }
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            if (h > 0) {                    // the second half of the dL/dh tiles takes the first one's place
                wave_lds_sync();
#pragma unroll
                for (int t = 0; t < TG; ++t)
                    *reinterpret_cast<float4 *>(scr_g + (t * 4 + q) * BLK + j * 4) =
                        make_float4(ghv[h * TG + t][0], ghv[h * TG + t][1], ghv[h * TG + t][2], ghv[h * TG + t][3]);
                wave_lds_sync();
            }
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)     // A-operand: [D-row j of tile t][k = row 4s+q]
                    accW2[h * TG + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        scr_g[(t * 4 + (j >> 2)) * BLK + (4 * s + q) * 4 + (j & 3)], bh[s], accW2[h * TG + t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s)     // A-operand: [input 16 t + j][k = row 4s+q]
                accW1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(scr_x[(4 * s + q) * XROW + 16 * t + j], bp[s],
                                                                accW1[t], 0, 0, 0);
    }

    // 6. wave results -> LDS, summed over the workgroup's waves -> part[block][M]
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < T2; ++t)
        *reinterpret_cast<float4 *>(scr + (t * 64 + lane) * 4) =
            make_float4(accW2[t][0], accW2[t][1], accW2[t][2], accW2[t][3]);
#pragma unroll
    for (int t = 0; t < T1; ++t)
        *reinterpret_cast<float4 *>(scr + T2 * 256 + (t * 64 + lane) * 4) =
            make_float4(accW1[t][0], accW1[t][1], accW1[t][2], accW1[t][3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = sb1[r];
        v += __shfl_xor(v, 1, kWave);
        v += __shfl_xor(v, 2, kWave);
        v += __shfl_xor(v, 4, kWave);
        v += __shfl_xor(v, 8, kWave);
        if (j == 0) scr[T2 * 256 + T1 * 256 + 4 * q + r] = v;
    }
    __syncthreads();
    const float *all = lds + n_params;
    for (int c = threadIdx.x; c < M; c += kBlock) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += all[w * SCR + c];
        part[(long long)blockIdx.x * M + c] = s;
    }
}

// column sums of part[n_blocks][M] -> out[M]: 64 columns x 16 row slices per 1024-thread
// workgroup, coalesced along the columns, fixed-order tree (deterministic)
__global__ __launch_bounds__(1024) void k_colsum2d(const float *__restrict__ part, float *__restrict__ out,
                                                   int n_blocks, int M)
{
    __shared__ float red[1024];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    float s = 0.0f;
    if (c < M) {
#pragma unroll 4
        for (int b = slice; b < n_blocks; b += 16) s += part[(long long)b * M + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o >= 64; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (slice == 0 && c < M) out[c] = red[threadIdx.x];
}

template <int EPL>
static int launch_train_bwd(const float *x, float *g, const float *gld, const float *params, int n_params,
                            int steps2, float *out, float *workspace, int64_t N, int inverse_form,
                            const float *gscale, int g_reversed, hipStream_t s, const char *fn)
{
    constexpr int M = (EPL / 2) * 256 + (EPL / 4) * 256 + 16;
    constexpr int TILES = 2 * 4 * 72 + ((EPL / 2) / (EPL == 32 ? 2 : 1)) * 4 * 72 + 16 * (4 * EPL + 16);
    constexpr int SCR = TILES > M ? TILES : M;                    // as in the kernel
    const size_t lds = ((size_t)n_params + (kBlock / 64) * SCR) * sizeof(float);
    if (lds > 160 * 1024) return fail(TFK_EINVAL, "%s: %zu bytes of LDS needed", fn, lds);
    const void *kern = inverse_form ? reinterpret_cast<const void *>(&k_affine_coupling_train_bwd<EPL, true>)
                                    : reinterpret_cast<const void *>(&k_affine_coupling_train_bwd<EPL, false>);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);
    }
    int per_cu = 0;
    hipError_t e = inverse_form
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_affine_coupling_train_bwd<EPL, true>, kBlock, lds)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_affine_coupling_train_bwd<EPL, false>, kBlock, lds);
    if (e != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    if (per_cu > 8) per_cu = 8;                          // the workspace holds cu_count() * 8 partial rows
    constexpr int rows_per_block = (kBlock / 64) * 16;  // (tfk_coupling_train_bwd_workspace_bytes)
    int64_t grid = (N + rows_per_block - 1) / rows_per_block;
    const int64_t cap = (int64_t)cu_count() * per_cu;          // one resident set: few partial rows to add
    if (grid > cap) grid = cap;
    if (inverse_form)
        hipLaunchKernelGGL((k_affine_coupling_train_bwd<EPL, true>), dim3((int)grid), dim3(kBlock), lds, s, x, g,
                           gld, params, n_params, steps2, workspace, (long long)N, gscale, g_reversed);
    else
        hipLaunchKernelGGL((k_affine_coupling_train_bwd<EPL, false>), dim3((int)grid), dim3(kBlock), lds, s, x, g,
                           gld, params, n_params, steps2, workspace, (long long)N, gscale, g_reversed);
    if (int rc = check_launch(fn)) return rc;
    hipLaunchKernelGGL(k_colsum2d, dim3((M + 63) / 64), dim3(1024), 0, s, workspace, out, (int)grid, M);
    return check_launch(fn);
}

extern "C" {

int tfk_coupling_train_bwd_supported(int32_t D) { return (D == 64 || D == 128 || D == 256) ? 1 : 0; }

int64_t tfk_coupling_train_bwd_out_floats(int32_t D)
{
    if (!tfk_coupling_train_bwd_supported(D)) return 0;
    const int EPL = D / 8;
    return (int64_t)(EPL / 2) * 256 + (EPL / 4) * 256 + 16;
}

int64_t tfk_coupling_train_bwd_workspace_bytes(int32_t D)
{
    return tfk_coupling_train_bwd_out_floats(D) * (int64_t)cu_count() * 8 * (int64_t)sizeof(float);
}

int tfk_affine_coupling_train_bwd(const float *x, float *g, const float *gld, const float *params,
                                  int64_t n_params, int32_t gemm2_steps, float *out, float *workspace,
                                  int64_t N, int32_t D, int32_t inverse_form, const float *gscale,
                                  int32_t g_reversed, void *stream)
{
    const char *fn = "tfk_affine_coupling_train_bwd";
    if (N < 1) return fail(TFK_EINVAL, "%s: N = %lld < 1", fn, (long long)N);
    if (!tfk_coupling_train_bwd_supported(D)) return fail(TFK_EINVAL, "%s: D = %d must be 64, 128 or 256", fn, D);
    if (gemm2_steps < 1 || gemm2_steps > 4) return fail(TFK_EINVAL, "%s: GEMM-2 steps %d not in [1, 4]", fn, gemm2_steps);
    const int EPL = D / 8, T2 = EPL / 2, T1 = EPL / 4;
    const int64_t need = (int64_t)EPL * 64 + 16 + (int64_t)T2 * gemm2_steps * 64 + T2 * 16 + T2 * 4 * 64 + T1 * 4 * 64;
    if (n_params != need) return fail(TFK_EINVAL, "%s: parameter block has %lld floats, expected %lld", fn,
                                      (long long)n_params, (long long)need);
    if (!x || !g || !gld || !params || !out || !workspace) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!aligned16(x) || !aligned16(g) || !aligned16(params) || (gscale && !aligned16(gscale)))
        return fail(TFK_EINVAL, "%s: x, g, params and gscale must be 16-byte aligned", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (EPL == 8)
        return launch_train_bwd<8>(x, g, gld, params, (int)n_params, gemm2_steps, out, workspace, N, inverse_form,
                                   gscale, g_reversed ? 1 : 0, s, fn);
    if (EPL == 32)
        return launch_train_bwd<32>(x, g, gld, params, (int)n_params, gemm2_steps, out, workspace, N, inverse_form,
                                    gscale, g_reversed ? 1 : 0, s, fn);
    return launch_train_bwd<16>(x, g, gld, params, (int)n_params, gemm2_steps, out, workspace, N, inverse_form,
                                gscale, g_reversed ? 1 : 0, s, fn);
}

}  // extern "C"

// =============================================================================================
// Training backward of one RQ-spline coupling layer (HalfSplit, FeedForward(Linear, Tanh, Linear)
// conditioner, hidden width <= 16, D = 64, 8 bins) with the conditioner re-evaluated in the
// kernel.  The layer-by-layer route moves h (2.9 KB per row) through HBM five times (conditioner
// re-evaluation, backward kernel, dL/dh for three GEMMs); here h never exists and dL/dh is written
// once, for the one product that contracts over the batch rows (dW2 = dL/dh^T hidden, done as a
// split-K batched GEMM by the caller):
//   1. GEMM 1 + tanh                                  (as the forward flow program)
//   2. per target element e of the lane (run-time loop): its 24 (23 + pad) spline parameters by
//      6 MFMA tiles, rqs_bwd_eval in registers -> dL/dx_B[e] and the element's dL/dh record,
//      24 MFMAs fold the record into dL/dhidden (A-operand = W2 packed so that the result lands
//      on the register that holds the unit), and the record goes to HBM in ACCUMULATOR order:
//      gh_perm[row][(6 e + c) * 16 + 4 q + r] = dL/d(parameter 4c + r of element 8q + e)
//      (one float4 per tile and lane; the caller un-permutes the 768 rows of dW2 instead);
//   3. dL/dpre = dL/dhidden * tanh', written as gpre_perm[row][4 q + r] (unit 4 r + q);
//   4. dL/dx_A = W1^T dL/dpre by MFMA, added to the source half of g.
// Parameter block (floats): A1[8][64] | b1[16] | A2[48][steps2][64] | b2[48][16] | A2T[48][4][64] |
// A1T[2][4][64]  (tile index = 6 e + c) -- 105 KB for hidden 14: one 512-thread workgroup per CU.
// =============================================================================================
template <bool INVERSE>
__global__ __launch_bounds__(512) void k_rqs_coupling_train_bwd(
    const float *__restrict__ x, float *g, const float *__restrict__ gld, const float *__restrict__ params,
    int n_params, int steps2, float *__restrict__ gh_perm, float *__restrict__ gpre_perm, long long N,
    RqsConst C, const float *__restrict__ gscale, int g_reversed, float *__restrict__ hid_perm)
{
    constexpr int EPL = 8, D = 64, HALF = 32, T2 = 48, T1 = 2, BLOCK = 512;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const float4 *src = reinterpret_cast<const float4 *>(params);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < (n_params >> 2); i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const float *A1 = lds;
    const float *b1 = A1 + EPL * 64;
    const float *A2 = b1 + 16;
    const float *b2 = A2 + T2 * steps2 * 64;
    const float *A2T = b2 + T2 * 16;
    const float *A1T = A2T + T2 * 4 * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    constexpr int rows_per_block = (BLOCK / 64) * 16;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row0 = (long long)blockIdx.x * rows_per_block + wave * 16; row0 < N; row0 += stride) {
        const long long row = row0 + j;
        const bool valid = row < N;
        const long long rr = valid ? row : N - 1;
        float xa[EPL], xb[EPL], ga[EPL], gb[EPL];
        {
            const float4 *pa = reinterpret_cast<const float4 *>(x + rr * D + EPL * q);
            const float4 *pb = reinterpret_cast<const float4 *>(x + rr * D + HALF + EPL * q);
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                const float4 va = pa[i], vb = pb[i];
                xa[4 * i] = va.x; xa[4 * i + 1] = va.y; xa[4 * i + 2] = va.z; xa[4 * i + 3] = va.w;
                xb[4 * i] = vb.x; xb[4 * i + 1] = vb.y; xb[4 * i + 2] = vb.z; xb[4 * i + 3] = vb.w;
            }
            if (!g_reversed) {
                const float4 *qa = reinterpret_cast<const float4 *>(g + rr * D + EPL * q);
                const float4 *qb = reinterpret_cast<const float4 *>(g + rr * D + HALF + EPL * q);
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    const float4 wa = qa[i], wb = qb[i];
                    ga[4 * i] = wa.x; ga[4 * i + 1] = wa.y; ga[4 * i + 2] = wa.z; ga[4 * i + 3] = wa.w;
                    gb[4 * i] = wb.x; gb[4 * i + 1] = wb.y; gb[4 * i + 2] = wb.z; gb[4 * i + 3] = wb.w;
                }
            } else {
                const float4 *qa = reinterpret_cast<const float4 *>(g + rr * D + D - EPL * (q + 1));
                const float4 *qb = reinterpret_cast<const float4 *>(g + rr * D + HALF - EPL * (q + 1));
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    const float4 wa = qa[i], wb = qb[i];
                    ga[EPL - 1 - 4 * i] = wa.x; ga[EPL - 2 - 4 * i] = wa.y; ga[EPL - 3 - 4 * i] = wa.z; ga[EPL - 4 - 4 * i] = wa.w;
                    gb[EPL - 1 - 4 * i] = wb.x; gb[EPL - 2 - 4 * i] = wb.y; gb[EPL - 3 - 4 * i] = wb.z; gb[EPL - 4 - 4 * i] = wb.w;
                }
            }
            if (gscale) {
                const float4 *sa = reinterpret_cast<const float4 *>(gscale + EPL * q);
                const float4 *sb = reinterpret_cast<const float4 *>(gscale + HALF + EPL * q);
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    const float4 ua = sa[i], ub = sb[i];
                    ga[4 * i] *= ua.x; ga[4 * i + 1] *= ua.y; ga[4 * i + 2] *= ua.z; ga[4 * i + 3] *= ua.w;
                    gb[4 * i] *= ub.x; gb[4 * i + 1] *= ub.y; gb[4 * i + 2] *= ub.z; gb[4 * i + 3] *= ub.w;
                }
            }
        }
        float gl = gld[rr];
        if (!valid) {
            gl = 0.0f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) ga[e] = gb[e] = 0.0f;
        }

        // 1. conditioner forward
        f32x4_t acc = *reinterpret_cast<const f32x4_t *>(b1 + 4 * q);
#pragma unroll
        for (int s = 0; s < EPL; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[s * 64 + lane], xa[s], acc, 0, 0, 0);
        float hid[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[r] = tanh_fast(acc[r]);

        // 2. element by element
        f32x4_t gha = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
        float *out_row = gh_perm + rr * (long long)(T2 * 16) + 4 * q;
        for (int e = 0; e < EPL; ++e) {
            float p[24];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int t = e * 6 + c;
                f32x4_t o = *reinterpret_cast<const f32x4_t *>(b2 + (t * 4 + q) * 4);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2) * 64 + lane], hid[0], o, 0, 0, 0);
                if (steps2 > 1) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2 + 1) * 64 + lane], hid[1], o, 0, 0, 0);
                if (steps2 > 2) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2 + 2) * 64 + lane], hid[2], o, 0, 0, 0);
                if (steps2 > 3) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * steps2 + 3) * 64 + lane], hid[3], o, 0, 0, 0);
                p[4 * c] = o[0]; p[4 * c + 1] = o[1]; p[4 * c + 2] = o[2]; p[4 * c + 3] = o[3];
            }
            float v = xb[0], A = gb[0];
#pragma unroll
            for (int i = 1; i < EPL; ++i) {
                v = (e == i) ? xb[i] : v;
                A = (e == i) ? gb[i] : A;
            }
            float p23[23];
#pragma unroll
            for (int i = 0; i < 23; ++i) p23[i] = p[i];
            float gv = A;                                           // identity outside the box
            if (v > C.minimum && v < C.maximum) {
                rqs_bwd_eval<8, INVERSE>(p23, v, C, A, gl, gv);
            } else {
#pragma unroll
                for (int i = 0; i < 23; ++i) p23[i] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < EPL; ++i) gb[i] = (e == i) ? gv : gb[i];
            // one accumulator per element (24 k-steps), combined afterwards: shorter fp32 chains
            f32x4_t ghe = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int t = e * 6 + c;
                const float r0 = p23[4 * c], r1 = p23[4 * c + 1], r2 = p23[4 * c + 2];
                const float r3 = (c == 5) ? 0.0f : p23[c == 5 ? 0 : 4 * c + 3];
                ghe = __builtin_amdgcn_mfma_f32_16x16x4f32(A2T[(t * 4 + 0) * 64 + lane], r0, ghe, 0, 0, 0);
                ghe = __builtin_amdgcn_mfma_f32_16x16x4f32(A2T[(t * 4 + 1) * 64 + lane], r1, ghe, 0, 0, 0);
                ghe = __builtin_amdgcn_mfma_f32_16x16x4f32(A2T[(t * 4 + 2) * 64 + lane], r2, ghe, 0, 0, 0);
                ghe = __builtin_amdgcn_mfma_f32_16x16x4f32(A2T[(t * 4 + 3) * 64 + lane], r3, ghe, 0, 0, 0);
                if (valid) *reinterpret_cast<float4 *>(out_row + t * 16) = make_float4(r0, r1, r2, r3);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gha[r] += ghe[r];
        }

        // 3. dL/dpre
        float gpre[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gpre[r] = gha[r] * (1.0f - hid[r] * hid[r]);
        if (valid)
            *reinterpret_cast<float4 *>(gpre_perm + row * 16 + 4 * q) = make_float4(gpre[0], gpre[1], gpre[2], gpre[3]);
        // the hidden activations in the same order, slot 15 (unit 15: hidden width <= 15) == 1 for the bias column of the
        // weight-gradient products that contract over the rows (tfk_rows_outer)
        if (valid && hid_perm)
            *reinterpret_cast<float4 *>(hid_perm + row * 16 + 4 * q) = make_float4(hid[0], hid[1], hid[2], q == 3 ? 1.0f : hid[3]);

        // 4. dL/dx_A
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            f32x4_t d = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(A1T[(t * 4 + r) * 64 + lane], gpre[r], d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[4 * t + r] += d[r];
        }
        if (valid) {
            float4 *qa = reinterpret_cast<float4 *>(g + row * D + EPL * q);
            float4 *qb = reinterpret_cast<float4 *>(g + row * D + HALF + EPL * q);
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                qa[i] = make_float4(ga[4 * i], ga[4 * i + 1], ga[4 * i + 2], ga[4 * i + 3]);
                qb[i] = make_float4(gb[4 * i], gb[4 * i + 1], gb[4 * i + 2], gb[4 * i + 3]);
            }
        }
    }
}

extern "C" {

int tfk_rqs_coupling_train_bwd_supported(int32_t D, int32_t n_bins) { return (D == 64 && n_bins == 8) ? 1 : 0; }

static int rqs_train_bwd_impl(const float *x, float *g, const float *gld, const float *params,
                              int64_t n_params, int32_t gemm2_steps, float *gh_perm, float *gpre_perm, float *hid_perm,
                              int64_t N, int32_t D, int32_t n_bins, float boundary, int32_t inverse,
                              const float *gscale, int32_t g_reversed, void *stream, const char *fn)
{
    if (N < 1) return fail(TFK_EINVAL, "%s: N = %lld < 1", fn, (long long)N);
    if (!tfk_rqs_coupling_train_bwd_supported(D, n_bins))
        return fail(TFK_EINVAL, "%s: D = %d, n_bins = %d (the kernel exists for D = 64, 8 bins)", fn, D, n_bins);
    if (gemm2_steps < 1 || gemm2_steps > 4) return fail(TFK_EINVAL, "%s: GEMM-2 steps %d not in [1, 4]", fn, gemm2_steps);
    if (!(boundary > 0.0f)) return fail(TFK_EINVAL, "%s: boundary must be positive", fn);
    const int64_t need = 8 * 64 + 16 + (int64_t)48 * gemm2_steps * 64 + 48 * 16 + 48 * 4 * 64 + 2 * 4 * 64;
    if (n_params != need) return fail(TFK_EINVAL, "%s: parameter block has %lld floats, expected %lld", fn,
                                      (long long)n_params, (long long)need);
    if (!x || !g || !gld || !params || !gh_perm || !gpre_perm) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!aligned16(x) || !aligned16(g) || !aligned16(params) || !aligned16(gh_perm) || !aligned16(gpre_perm) ||
        (gscale && !aligned16(gscale)))
        return fail(TFK_EINVAL, "%s: buffers must be 16-byte aligned", fn);
    RqsConst C;
    C.minimum = -boundary;
    C.maximum = boundary;
    C.span = (float)((double)boundary + (double)boundary);
    C.scale = (float)(1.0 - 1e-3 * 8.0);
    C.c = (float)log(expm1(1.0 - 1e-5));
    const size_t lds = (size_t)n_params * sizeof(float);
    if (lds > 160 * 1024) return fail(TFK_EINVAL, "%s: %zu bytes of LDS needed", fn, lds);
    const void *kern = inverse ? reinterpret_cast<const void *>(&k_rqs_coupling_train_bwd<true>)
                               : reinterpret_cast<const void *>(&k_rqs_coupling_train_bwd<false>);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);
    }
    constexpr int rows_per_block = (512 / 64) * 16;
    int64_t grid = (N + rows_per_block - 1) / rows_per_block;
    const int64_t cap = (int64_t)cu_count() * kGridOversubscribe;      // one workgroup per CU resident
    if (grid > cap) grid = cap;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (inverse)
        hipLaunchKernelGGL((k_rqs_coupling_train_bwd<true>), dim3((int)grid), dim3(512), lds, s, x, g, gld, params,
                           (int)n_params, gemm2_steps, gh_perm, gpre_perm, (long long)N, C, gscale, g_reversed ? 1 : 0, hid_perm);
    else
        hipLaunchKernelGGL((k_rqs_coupling_train_bwd<false>), dim3((int)grid), dim3(512), lds, s, x, g, gld, params,
                           (int)n_params, gemm2_steps, gh_perm, gpre_perm, (long long)N, C, gscale, g_reversed ? 1 : 0, hid_perm);
    return check_launch(fn);
}

int tfk_rqs_coupling_train_bwd(const float *x, float *g, const float *gld, const float *params,
                               int64_t n_params, int32_t gemm2_steps, float *gh_perm, float *gpre_perm,
                               int64_t N, int32_t D, int32_t n_bins, float boundary, int32_t inverse,
                               const float *gscale, int32_t g_reversed, void *stream)
{
    return rqs_train_bwd_impl(x, g, gld, params, n_params, gemm2_steps, gh_perm, gpre_perm, nullptr, N, D, n_bins,
                              boundary, inverse, gscale, g_reversed, stream, "tfk_rqs_coupling_train_bwd");
}

int tfk_rqs_coupling_train_bwd_hid(const float *x, float *g, const float *gld, const float *params,
                                   int64_t n_params, int32_t gemm2_steps, float *gh_perm, float *gpre_perm,
                                   float *hid_perm, int64_t N, int32_t D, int32_t n_bins, float boundary,
                                   int32_t inverse, const float *gscale, int32_t g_reversed, void *stream)
{
    const char *fn = "tfk_rqs_coupling_train_bwd_hid";
    if (!hid_perm || !aligned16(hid_perm)) return fail(TFK_EINVAL, "%s: hid_perm must be a 16-byte aligned (N, 16) buffer", fn);
    return rqs_train_bwd_impl(x, g, gld, params, n_params, gemm2_steps, gh_perm, gpre_perm, hid_perm, N, D, n_bins,
                              boundary, inverse, gscale, g_reversed, stream, fn);
}

}  // extern "C"

// =============================================================================================
// out[m][k] = sum_n A[n][m] * B[n][k]  (m < M, k < 16): the weight-gradient products of a training step that contract
// over the batch ROWS -- dW2 = dL/dh^T hidden (M = 768), dW1^T = x_A^T dL/dpre (M = 32), db1 (M = 16) -- without a
// GEMM-library call (those invalidate a hipGraph capture on this stack, and run a 768 x 17 output as ONE tile).
// v_mfma_f32_16x16x4_f32 contracts over k = 4 consecutive rows: lane (q, i) supplies A[row 4 s + q][column of tile t,
// M-index i] and B[row 4 s + q][i] -- both straight out of the row-major buffers, no transposes.  A lane reads its four
// A-values of four tiles as ONE float4 (columns 64 T + 4 i + (t & 3) of tile t = 4 T + (t & 3)), so a quarter-wave reads
// 256 contiguous bytes of a row.  Rows are strided over the waves of the grid; a workgroup adds its four waves' tiles in
// the LDS and writes ONE partial block; k_colsum2d adds the blocks in a fixed order (deterministic).
//   blockIdx.y = chunk of 256 columns (MT = 16 tiles) of A, or the single chunk of MT = 1 / 2 tiles (M = 16 / 32).
//   Output order (floats): tile t, lane (q, j), register r  ->  [(t * 64 + 16 q + j) * 4 + r] = out[col(t, 4 q + r)][j],
//   col(t, i) = 64 (t >> 2) + 4 i + (t & 3) for MT >= 4, 16 t + i otherwise.
// =============================================================================================
template <int MT>
__global__ __launch_bounds__(256) void k_rows_outer(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                    float *__restrict__ part, long long N, int M_total)
{
    __shared__ __attribute__((aligned(16))) float red[3 * MT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, i = lane & 15;
    const int chunk = blockIdx.y;                      // 16 MT columns each
    const float *Ac = A + chunk * 16 * MT;
    f32x4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
    const long long n_steps = (N + 3) >> 2;
    const long long stride = (long long)gridDim.x * 4;
    if constexpr (MT >= 4) {
        // the next step's operands are in flight while this step's MFMAs issue (a wave's loads would otherwise wait
        // behind 16 MFMAs of 32 cycles each: 3.7 -> 5.0 TB/s on 2^18 x 768, measured)
        float4 a[MT / 4], an[MT / 4];
        float b = 0.0f, bn = 0.0f;
        auto fetch = [&](long long s, float4 (&dst)[MT / 4], float &bd) {
            const long long row = 4 * s + q;
            const bool ok = row < N;
            const long long rr = ok ? row : N - 1;
            bd = ok ? B[rr * 16 + i] : 0.0f;
#pragma unroll
            for (int T = 0; T < MT / 4; ++T) dst[T] = *reinterpret_cast<const float4 *>(Ac + rr * lda + 64 * T + 4 * i);
        };
        long long s = (long long)blockIdx.x * 4 + wave;
        if (s < n_steps) fetch(s, a, b);
        for (; s < n_steps; s += stride) {
            const bool more = s + stride < n_steps;
            if (more) fetch(s + stride, an, bn);
#pragma unroll
            for (int T = 0; T < MT / 4; ++T) {
                acc[4 * T + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[T].x, b, acc[4 * T + 0], 0, 0, 0);
                acc[4 * T + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[T].y, b, acc[4 * T + 1], 0, 0, 0);
                acc[4 * T + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[T].z, b, acc[4 * T + 2], 0, 0, 0);
                acc[4 * T + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[T].w, b, acc[4 * T + 3], 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int T = 0; T < MT / 4; ++T) a[T] = an[T];
                b = bn;
            }
        }
    } else {
        for (long long s = (long long)blockIdx.x * 4 + wave; s < n_steps; s += stride) {
            const long long row = 4 * s + q;
            const bool ok = row < N;
            const long long rr = ok ? row : N - 1;
            const float b = ok ? B[rr * 16 + i] : 0.0f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[rr * lda + 16 * t + i], b, acc[t], 0, 0, 0);
        }
    }
    // waves 1..3 -> LDS, wave 0 adds them in a fixed order and writes the workgroup's partial block
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
            *reinterpret_cast<f32x4_t *>(red + (((wave - 1) * MT + t) * 64 + lane) * 4) = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
        float *dst = part + ((long long)blockIdx.x * (M_total / 16) + chunk * MT) * 256;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f32x4_t v = acc[t];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const f32x4_t o = *reinterpret_cast<const f32x4_t *>(red + ((w * MT + t) * 64 + lane) * 4);
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            *reinterpret_cast<f32x4_t *>(dst + (t * 64 + lane) * 4) = v;
        }
    }
}

extern "C" {

int64_t tfk_rows_outer_workspace_bytes(int32_t M)
{
    return (int64_t)cu_count() * 2 * (int64_t)M * 16 * (int64_t)sizeof(float);
}

int tfk_rows_outer(const float *A, int32_t lda, int32_t M, const float *B, float *out, float *workspace,
                   int64_t N, void *stream)
{
    const char *fn = "tfk_rows_outer";
    if (N < 1) return fail(TFK_EINVAL, "%s: N = %lld < 1", fn, (long long)N);
    if (!(M == 16 || M == 32 || (M >= 256 && M % 256 == 0)) || M > 1024)
        return fail(TFK_EINVAL, "%s: M = %d must be 16, 32 or a multiple of 256 up to 1024", fn, M);
    if (lda < M || (lda & 3)) return fail(TFK_EINVAL, "%s: lda = %d must be a multiple of 4 and >= M = %d", fn, lda, M);
    if (!A || !B || !out || !workspace) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!aligned16(A) || !aligned16(B) || !aligned16(out) || !aligned16(workspace))
        return fail(TFK_EINVAL, "%s: buffers must be 16-byte aligned", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t steps = (N + 3) / 4;
    int64_t grid = (steps + 3) / 4;
    const int64_t cap = (int64_t)cu_count() * 2;              // (tfk_rows_outer_workspace_bytes)
    if (grid > cap) grid = cap;
    if (M == 16)
        hipLaunchKernelGGL((k_rows_outer<1>), dim3((int)grid, 1), dim3(256), 0, s, A, lda, B, workspace, (long long)N, M);
    else if (M == 32)
        hipLaunchKernelGGL((k_rows_outer<2>), dim3((int)grid, 1), dim3(256), 0, s, A, lda, B, workspace, (long long)N, M);
    else
        hipLaunchKernelGGL((k_rows_outer<16>), dim3((int)grid, M / 256), dim3(256), 0, s, A, lda, B, workspace, (long long)N, M);
    if (int rc = check_launch(fn)) return rc;
    hipLaunchKernelGGL(k_colsum2d, dim3((M * 16 + 63) / 64), dim3(1024), 0, s, workspace, out, (int)grid, M * 16);
    return check_launch(fn);
}

}  // extern "C"
