// tfk_spline.h -- rational-quadratic spline element (device), shared by the standalone RQS
// coupling kernels (tfk_rqs.hip) and the fused flow programs (tfk_flow.hip).
// Follows RationalQuadratic.rqs_forward_1d / rqs_inverse_1d and compute_bins
// (transformers/spline/rational_quadratic.py:45-200) in the reference's fp32 op order,
// including ATen's CPU softmax form e * (1 / sum).
#pragma once

#include <type_traits>

#include "tfk_common.h"

namespace tfk {

struct RqsConst {
    float minimum;   // -boundary
    float maximum;   // +boundary
    float span;      // maximum - minimum (python double, cast once)
    float scale;     // 1 - min_bin_size * n_bins (python double, cast once)
    float c;         // boundary_u_delta = log(expm1(1 - min_delta))
};

// F.softplus, beta = 1, threshold = 20: log1p(exp(v)) below the threshold
__device__ __forceinline__ float softplus20(float v) {
    return v > 20.0f ? v : log1p_pos(exp_noovf(fminf(v, 20.0f)));
}
// ... with the lean exp / log of the datapath-bound flow programs (tfk_common.h: within 1 ulp each)
__device__ __forceinline__ float softplus20_lean(float v) {
    const float y = exp_lean(fminf(v, 20.0f));
    const float u = 1.0f + y;
    const float r = log_lean(u) + (y - (u - 1.0f)) * __builtin_amdgcn_rcpf(u);
    return v > 20.0f ? v : r;
}

// rational_quadratic.py:56-63
template <bool LEAN = false>
__device__ __forceinline__ float rqs_log_det(float s, float dk, float dk1, float xi, float q,
                                             float term1)
{
    const float omx = 1.0f - xi;
    const float inner = dk1 * (xi * xi) + (2.0f * s) * q + dk * (omx * omx);
    const float den = s + term1 * q;
    const float log_num = 2.0f * (LEAN ? log_lean(s) : log_normal(s)) + (LEAN ? log_lean(inner) : log_normal(inner));
    const float log_den = 2.0f * (LEAN ? log_lean(den) : log_normal(den));
    return log_num - log_den;
}

// torch.clip: NaN passes through
__device__ __forceinline__ float clip01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// One in-box element.  p = its P = 3K-1 parameters: a pointer into LDS, or (REGS) a register
// array.  KT > 0: compile-time K, everything in registers; KT == 0: run-time K, streamed from
// LDS (no local arrays).
template <int KT, bool INVERSE, bool REGS = false, typename PT = const float *>
__device__ __forceinline__ void rqs_eval(const PT &p, int Krt, float v, const RqsConst &C,
                                         float &out, float &ld)
{
    const int K = KT > 0 ? KT : Krt;
    int k = 0;
    float bxk = C.minimum, bxk1 = C.maximum, byk = C.minimum, byk1 = C.maximum;

    if constexpr (KT > 0) {
        float ex[KT], ey[KT];
        float mx = 0.0f, my = 0.0f;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const float ux = p[j];
            // rational_quadratic.py:76; in the flow programs u_y / 1000 as one multiplication: its rounding
            // (1 ulp of a term ~1e-3 of u_x) disappears in the sum
            const float uy = ux + (REGS ? p[KT + j] * 1e-3f : div_1000(p[KT + j]));
            ex[j] = ux;
            ey[j] = uy;
            mx = j ? fmaxf(mx, ux) : ux;
            my = j ? fmaxf(my, uy) : uy;
        }
        float sx = 0.0f, sy = 0.0f;
#pragma unroll
        for (int j = 0; j < KT; ++j) {                     // softmax numerators, :46
            // argument <= 0; the flow programs (REGS) are VALU-issue bound: 6-op exp, within 1 ulp
            ex[j] = REGS ? exp_lean(ex[j] - mx) : exp_noovf(ex[j] - mx);
            ey[j] = REGS ? exp_lean(ey[j] - my) : exp_noovf(ey[j] - my);
            sx += ex[j];
            sy += ey[j];
        }
        const float rx = div_fast(1.0f, sx), ry = div_fast(1.0f, sy);
        // torch.cumsum on the CPU -- the reference's path -- accumulates fp32 inputs in DOUBLE and rounds every prefix once
        // (ATen ReduceOpsKernel.cpp: acc_type<float, false>); the per-layer kernels (HBM-bound) do the same, the
        // VALU-bound flow programs (REGS) keep the fp32 running sum
        typedef typename std::conditional<REGS, float, double>::type run_t;
        run_t runx = 0, runy = 0;
        float prevx = C.minimum, prevy = C.minimum;
        bool prev_below = true;                            // knot 0 = minimum < v (in box)
#pragma unroll
        for (int j = 1; j <= KT; ++j) {
            runx = runx + (run_t)(kRqsMinBin + C.scale * (ex[j - 1] * rx));   // :47-48
            runy = runy + (run_t)(kRqsMinBin + C.scale * (ey[j - 1] * ry));
            const float kx = (j == KT) ? C.maximum : C.span * (float)runx + C.minimum;   // :50-52
            const float ky = (j == KT) ? C.maximum : C.span * (float)runy + C.minimum;
            // searchsorted(knots, v) - 1, right=False: last knot strictly below v (:82/:147)
            const bool below = (INVERSE ? ky : kx) < v;
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
    } else {
        float mx = p[0], my = p[0] + div_1000(p[K]);
        for (int j = 1; j < K; ++j) {
            mx = fmaxf(mx, p[j]);
            my = fmaxf(my, p[j] + div_1000(p[K + j]));
        }
        float sx = 0.0f, sy = 0.0f;
        for (int j = 0; j < K; ++j) {
            sx += exp_noovf(p[j] - mx);
            sy += exp_noovf((p[j] + div_1000(p[K + j])) - my);
        }
        const float rx = div_fast(1.0f, sx), ry = div_fast(1.0f, sy);
        double runx = 0.0, runy = 0.0;                      // (double accumulator: see above)
        float prevx = C.minimum, prevy = C.minimum;
        bool prev_below = true;
        for (int j = 1; j <= K; ++j) {
            const float e_x = exp_noovf(p[j - 1] - mx);
            const float e_y = exp_noovf((p[j - 1] + div_1000(p[K + j - 1])) - my);
            runx = runx + (double)(kRqsMinBin + C.scale * (e_x * rx));
            runy = runy + (double)(kRqsMinBin + C.scale * (e_y * ry));
            const float kx = (j == K) ? C.maximum : C.span * (float)runx + C.minimum;
            const float ky = (j == K) ? C.maximum : C.span * (float)runy + C.minimum;
            const bool below = (INVERSE ? ky : kx) < v;
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
    }

    const float wk = bxk1 - bxk;                           // bin_sizes = bins[1:] - bins[:-1], :53
    const float hk = byk1 - byk;
    // u_d = pad(h[2K:], (1, 1), value = c) (:127); only delta_k and delta_k+1 are used.
    // (the discarded LDS reads at k == 0 / k == K-1 stay inside the padded tile)
    float udk = C.c, udk1 = C.c;
    if constexpr (REGS) {           // registers cannot be indexed by k: select instead
#pragma unroll
        for (int j = 0; j < KT - 1; ++j) {
            udk = (k == j + 1) ? p[2 * KT + j] : udk;
            udk1 = (k == j) ? p[2 * KT + j] : udk1;
        }
    } else {
        udk = (k == 0) ? C.c : p[2 * K + k - 1];
        udk1 = (k == K - 1) ? C.c : p[2 * K + k];
    }
    const float tk = C.c + (REGS ? udk * 1e-3f : div_1000(udk)), tk1 = C.c + (REGS ? udk1 * 1e-3f : div_1000(udk1));
    const float dk = kRqsMinDelta + (REGS ? softplus20_lean(tk) : softplus20(tk));      // :77
    const float dk1 = kRqsMinDelta + (REGS ? softplus20_lean(tk1) : softplus20(tk1));
    const float s = div_fast(hk, wk);                      // :94 / :159
    const float term1 = dk1 + dk - 2.0f * s;               // :97 / :162

    if (!INVERSE) {
        float xi = div_fast(v - bxk, wk);                  // :99
        xi = clip01(xi);                                   // :100
        const float q = xi * (1.0f - xi);                  // :101
        const float num0 = hk * (s * (xi * xi) + dk * q);  // :104
        const float den0 = s + term1 * q;                  // :105
        out = byk + div_fast(num0, den0);                  // :106
        ld = rqs_log_det<REGS>(s, dk, dk1, xi, q, term1);  // :109
    } else {
        const float term0 = v - byk;                       // :164
        const float term2 = hk * dk;                       // :165
        const float a = (hk * s - term2) + term0 * term1;  // :167
        const float b = term2 - term0 * term1;             // :168
        const float c = (-s) * term0;                      // :169
        float r = sqrtf(b * b - (4.0f * a) * c);           // :171
        r = r < 0.0f ? 0.0f : r;
        float xi = div_fast(2.0f * c, (-b) - r);           // :173
        xi = clip01(xi);                                   // :174
        const float q = xi * (1.0f - xi);                  // :175
        out = xi * wk + bxk;                               // :178
        ld = -rqs_log_det<REGS>(s, dk, dk1, xi, q, term1); // :181
    }
}

}  // namespace tfk
