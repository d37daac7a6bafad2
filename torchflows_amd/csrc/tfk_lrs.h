// tfk_lrs.h -- linear rational spline element (device), shared by the coupling kernels (tfk_lrs.hip) and the
// sequential MADE walk (tfk_made.hip).  Follows LinearRational (spline/linear_rational.py:9-182).
#pragma once

#include "tfk_common.h"
#include "tfk_spline.h"

namespace tfk {

constexpr float kLrsMinBin = 1e-2f;
constexpr float kLrsMinD = 1e-5f;
constexpr float kLrsEps = 5e-10f;

struct LrsConst {
    float minimum, maximum, span;
    float scale;     // 1 - 1e-2 * K
    float c;         // log(exp(1 - 1e-5) - 1)
};

__device__ __forceinline__ float lrs_deriv(float u, float c) {       // linear_rational.py:79, :89
    return softplus20(c + u / 100.0f) + kLrsMinD;
}

template <int KT, bool INVERSE>
__device__ __forceinline__ void lrs_eval(const float *p, float v, const LrsConst &C, float &out, float &ld)
{
    float ex[KT], ey[KT];
    float mx = 0.0f, my = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const float ux = p[j];
        const float uy = ux + p[KT + j] / 100.0f;                   // :87
        ex[j] = ux;
        ey[j] = uy;
        mx = j ? fmaxf(mx, ux) : ux;
        my = j ? fmaxf(my, uy) : uy;
    }
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        ex[j] = exp_noovf(ex[j] - mx);
        ey[j] = exp_noovf(ey[j] - my);
        sx += ex[j];
        sy += ey[j];
    }
    const float rx = 1.0f / sx, ry = 1.0f / sy;
    int k = 0;
    float xk = C.minimum, xk1 = C.maximum, yk = C.minimum, yk1 = C.maximum;
    // (torch.cumsum on the CPU accumulates fp32 inputs in double and rounds every prefix once: tfk_spline.h)
    double runx = 0.0, runy = 0.0;
    float prevx = C.minimum, prevy = C.minimum;
    bool prev_below = true;
#pragma unroll
    for (int j = 1; j <= KT; ++j) {
        runx = runx + (double)(kLrsMinBin + C.scale * (ex[j - 1] * rx));    // :69-71
        runy = runy + (double)(kLrsMinBin + C.scale * (ey[j - 1] * ry));
        const float kx = (j == KT) ? C.maximum : C.span * (float)runx + C.minimum;
        const float ky = (j == KT) ? C.maximum : C.span * (float)runy + C.minimum;
        const bool below = (INVERSE ? ky : kx) < v;                 // searchsorted left, :105 / :150
        const bool sel = prev_below && !below;
        k = sel ? j - 1 : k;
        xk = sel ? prevx : xk;
        xk1 = sel ? kx : xk1;
        yk = sel ? prevy : yk;
        yk1 = sel ? ky : yk1;
        prev_below = below;
        prevx = kx;
        prevy = ky;
    }
    const float lam = 1.0f / (1.0f + exp_noovf(-p[2 * KT + k]));    // sigmoid, :88
    // boundary derivatives are exactly 1 (pad value, :80); the discarded reads stay in the record
    const float dk = (k == 0) ? 1.0f : lrs_deriv(p[3 * KT + k - 1], C.c);
    const float dk1 = (k == KT - 1) ? 1.0f : lrs_deriv(p[3 * KT + k], C.c);
    const float w0 = softplus20(p[4 * KT - 1]);                     // :41
    const float wk = w0 * sqrtf(1.0f / dk);                         // :42 (d_0 = 1)
    const float wk1 = w0 * sqrtf(1.0f / dk1);
    const float one_m = 1.0f - lam;
    const float ym = (one_m * wk * yk + lam * wk1 * yk1) / (one_m * wk + lam * wk1);       // :58-61
    const float dx = xk1 - xk;
    const float wm = (lam * wk * dk + one_m * wk1 * dk1) * (dx / (yk1 - yk));              // :62-67
    if (!INVERSE) {
        const float phi = (v - xk) / dx;                            // :110
        if (!(phi > lam)) {                                         // :113-121
            const float den = wk * (lam - phi) + wm * phi;
            out = (wk * yk * (lam - phi) + wm * ym * phi) / den;
            ld = log_normal(lam * wk * wm * (ym - yk)) - log_normal(den * den + kLrsEps) - log_normal(dx);
        } else {                                                    // :123-131
            const float den = wm * (1.0f - phi) + wk1 * (phi - lam);
            out = (wm * ym * (1.0f - phi) + wk1 * yk1 * (phi - lam)) / den;
            ld = log_normal(one_m * wm * wk1 * (yk1 - ym)) - log_normal(den * den + kLrsEps) - log_normal(dx);
        }
    } else {
        if (!(v > ym)) {                                            // :157-166
            const float den = wk * (yk - v) + wm * (v - ym);
            out = (lam * wk * (yk - v)) / den * dx + xk;
            ld = log_normal(lam * wk * wm * (ym - yk)) - log_normal(den * den + kLrsEps) + log_normal(dx);
        } else {                                                    // :168-176
            const float den = wk1 * (yk1 - v) + wm * (v - ym);
            out = (lam * wk1 * (yk1 - v) + wm * (v - ym)) / den * dx + xk;
            ld = log_normal(one_m * wm * wk1 * (yk1 - ym)) - log_normal(den * den + kLrsEps) + log_normal(dx);
        }
    }
}

}  // namespace tfk
