// tfk_flow_rqs_chain.h -- a whole chain of rational-quadratic-spline couplings in ONE launch
// (templates; instantiated per row width in tfk_flow_rqs_chain_{8,16,32}.hip, dispatched from tfk_flow_run_mfma).
//
// A spline coupling needs 23 parameters per target element: GEMM 2's A-operands are 49 KB per 8 target elements per
// lane-group, far more than a whole chain's worth fits the LDS (the interpreter of tfk_flow_mfma.h therefore ran one
// launch per coupling, and the rows crossed HBM 2 x 9 times for config 3).  Here the rows STAY in registers for the
// whole chain and the operands come to them: the workgroup walks the layers together and stages, per layer, the
// conditioner's first GEMM and then one CHUNK of 8 target elements' GEMM-2 operands at a time from global memory
// (L2-resident: every workgroup reads the same 55 KB per layer) into LDS.  Rows are read once and, for log_prob,
// never written: 4 D + 4 bytes per row instead of 17 x 4 D.
//
// The spline itself is evaluated in a form that costs ~170 vector instructions per element instead of ~420
// (rocprofv3: SQ_INSTS_VALU 220 M -> see profiles/r02/):
//   * the packer (fused.py) folds the reference's  u_x + u_y / 1000,  c + u_d / 1000  (rational_quadratic.py:76-77)
//     and the factor log2(e) into W2 / b2, so the kernel receives softmax logits for exp2 directly;
//   * knots are  minimum + j * span * min_bin + (span * scale / sum) * partial sums  (one fma + one add per knot);
//   * the bin is found by a 3-level binary search that narrows knots, opposite knots and derivative logits together
//     (30 selects instead of ~56), strict '<' as searchsorted(right=False) (an input on a knot goes left);
//   * softplus as ln 2 * log2(1 + exp2(.)), divisions as v_rcp_f32 (1 ulp), the log-det as ONE logarithm of
//     (s / den)^2 * inner, accumulated in base 2;
//   * elementwise layers between the couplings are deferred exactly as in tfk_flow_chain.h (pre-affine of the target
//     plane, folded into W1 / b1 for the source plane, TFK_OP_EW_FMA at the end).
//
// Parameter block of a lean RQS op (floats, GLOBAL memory), NC = EPL / 8 chunks:
//   head   A1[EPL/4][64][4] | b1[4][4] | pre_s[HALF] | pre_t[HALF]
//   chunk  A2[48][64][4] | b2[48][4][4]        tile 6 e + c of a chunk = parameters 4 c .. 4 c + 3 of the lane-group's
//                                              target element 8 chunk + e; A2 lane-major over the (<= 4) k-steps
//   parameters per element: [0, 8) width logits, [8, 16) height logits (u_x + u_y / 1000), [16, 23) derivative
//   logits (c + u_d / 1000), all times log2(e); 23 = padding.
#pragma once
#include "tfk_common.h"
#ifndef TFK_CHAIN_OVERSUB
#define TFK_CHAIN_OVERSUB kGridOversubscribe   // resident sets of workgroups a launch is cut into (tuning: tools/variants.sh)
#endif
#include <type_traits>
#include "tfk_flow_chain.h"

namespace tfk {

struct RqsLean {
    float minimum, maximum;   // -boundary, +boundary
    float g;                  // span * (1 - min_bin * K)
    float cmin;               // span * min_bin
    float d_edge;             // (c + c / 1000) * log2(e): the padded derivative logits (rational_quadratic.py:127)
    float knot_c[7];          // minimum + j * span * min_bin, j = 1..7: the constant part of inner knot j
};

struct RqsChainProg {
    int n_layers;
    int first_src;
    int ew_offset;            // TFK_OP_EW_FMA (global offset), -1: none
    int layer_stride;         // floats between consecutive layers' blocks
    int offset0;              // first layer's block
    int ctx_steps;            // k-steps of context in GEMM 1 (0: none; the head then ends with A1c[HT][64][4])
    int made;                 // MADE-based spline layers (TFK_OP_MADE_*_LEAN spline kinds): both planes in, both planes out
    int side_floats;          // (context programs) floats of the elementwise ops' blocks, staged in the LDS behind the chunk
    // elementwise ops in front of the couplings / behind the closing TFK_OP_EW_FMA (see ChainProg): kind, offset in params
    int pre_kind[kChainSideOps], pre_off[kChainSideOps], pre_lds[kChainSideOps];      // (..._lds: float offset in the LDS)
    int post_kind[kChainSideOps], post_off[kChainSideOps], post_lds[kChainSideOps];
    RqsLean C;
    double *sum_ws;           // tfk_flow_run_mfma_sum (see ChainProg)
    double *sum_out;
    unsigned long long move_mask;   // odd event sizes: layer l first takes the middle element over (ChainProg::move_mask)
};

constexpr int kRqsChunkFloats = 48 * 256 + 48 * 16;
// bf16x3 operand format (FMT = 1): a chunk holds 4 target elements = 24 tiles; per tile and lane two 16-byte A-operands
// [W_hi | W_mid] and [W_lo | W_hi] (4 bf16 each = the lane-group's 4 hidden units), then b2[24][4][4]
// (no b2: the bias rides in the GEMM as hidden unit 15 = 1, hidden width <= 15)
constexpr int kRqsChunk3Dwords = 24 * 2 * 64 * 4;

typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef int ci32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp_f(float v) { return __builtin_amdgcn_rcpf(v); }
// v_rcp_f32 + one Newton step: the reciprocal to ~0.5 ulp instead of 1 ulp.  For the three reciprocals whose error is
// multiplied by a knot position / bin size of up to 100 (the normaliser of each softmax, the bin width, the denominator of
// the rational function) -- 8 fmas per spline element; -DTFK_RQS_NEWTON=0 restores the 1-ulp forms.
#ifndef TFK_RQS_NEWTON
#define TFK_RQS_NEWTON 1
#endif
__device__ __forceinline__ float rcp_n(float v)
{
    const float r = __builtin_amdgcn_rcpf(v);
#if TFK_RQS_NEWTON
    return fmaf(fmaf(-v, r, 1.0f), r, r);
#else
    return r;
#endif
}
__device__ __forceinline__ float exp2_f(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ float log2_f(float v) { return __builtin_amdgcn_logf(v); }

// The 7 inner knots of one axis from its 8 pre-scaled logits (rational_quadratic.py:46-52, linear_rational.py:44-47):
//   knot_j = minimum + span * sum_{i<j} (min_bin + scale * softmax_i) = knot_c[j-1] + (g / sum) * prefix_j,
// prefix sums of the softmax numerators (the last one is the normaliser), then one fma per knot.
// FAST: no maximum is subtracted -- exact to rounding while the logits stay within +-kSoftmaxFastMax (the sums cannot
// leave fp32's range, terms 2^-128 below the largest are lost either way); the lane keeps the largest |logit| it has
// seen in `amax` and the kernel re-runs the 16 rows with FAST = false if any lane of the workgroup saw a larger one
// (22 of an element's ~150 vector instructions).  -DTFK_SOFTMAX_FAST=0: always subtract the maximum.
#ifndef TFK_SOFTMAX_FAST
#define TFK_SOFTMAX_FAST 1
#endif
constexpr float kSoftmaxFastMax = 64.0f;
template <bool FAST>
__device__ __forceinline__ void spline_knots(const float *u, const RqsLean &C, float (&K)[9], float &amax)
{
    float pre[8];
    if constexpr (FAST) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) amax = fmaxf(fmaxf(amax, __builtin_fabsf(u[j])), __builtin_fabsf(u[j + 1]));
        pre[0] = exp2_f(u[0]);
#pragma unroll
        for (int j = 1; j < 8; ++j) pre[j] = pre[j - 1] + exp2_f(u[j]);
    } else {
        // (plain fmaxf: the compiler folds the chain into v_max3_f32; hand-placed v_max3_f32 through inline asm is NOT an
        // option on matrix-core results -- the hazard recognizer inserts no wait states for asm operands, measured as
        // last-bit differences between runs)
        float m = u[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) m = fmaxf(m, u[j]);
        pre[0] = exp2_f(u[0] - m);
#pragma unroll
        for (int j = 1; j < 8; ++j) pre[j] = pre[j - 1] + exp2_f(u[j] - m);
    }
    const float g = C.g * rcp_n(pre[7]);
    K[0] = C.minimum;
    K[8] = C.maximum;
#pragma unroll
    for (int j = 1; j < 8; ++j) K[j] = fmaf(pre[j - 1], g, C.knot_c[j - 1]);
}

// One in-box element.  p: its 24 pre-scaled parameters (registers).  l2 = log2 |dz/dx| (forward) or of dx/dz (inverse).
template <bool INVERSE, bool FAST = false>
__device__ __forceinline__ void rqs_eval_lean(const float (&p)[24], float v, const RqsLean &C, float &out, float &l2, float &amax)
{
    float X[9], Y[9], Dl[9];
    spline_knots<FAST>(p, C, X, amax);
    spline_knots<FAST>(p + 8, C, Y, amax);
    Dl[0] = C.d_edge; Dl[8] = C.d_edge;
#pragma unroll
    for (int j = 1; j < 8; ++j) Dl[j] = p[15 + j];
    // bin = number of knots strictly below v, minus one (:82 / :147); three halvings that carry the searched knots,
    // the opposite knots and the derivative logits along
    float S5[5], O5[5], D5[5];
    {
        const bool up = (INVERSE ? Y[4] : X[4]) < v;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            S5[i] = up ? (INVERSE ? Y[4 + i] : X[4 + i]) : (INVERSE ? Y[i] : X[i]);
            O5[i] = up ? (INVERSE ? X[4 + i] : Y[4 + i]) : (INVERSE ? X[i] : Y[i]);
            D5[i] = up ? Dl[4 + i] : Dl[i];
        }
    }
    float S3[3], O3[3], D3[3];
    {
        const bool up = S5[2] < v;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            S3[i] = up ? S5[2 + i] : S5[i];
            O3[i] = up ? O5[2 + i] : O5[i];
            D3[i] = up ? D5[2 + i] : D5[i];
        }
    }
    const bool up = S3[1] < v;
    const float s0 = up ? S3[1] : S3[0], s1 = up ? S3[2] : S3[1];
    const float o0 = up ? O3[1] : O3[0], o1 = up ? O3[2] : O3[1];
    const float t0 = up ? D3[1] : D3[0], t1 = up ? D3[2] : D3[1];
    const float bxk = INVERSE ? o0 : s0, bxk1 = INVERSE ? o1 : s1;
    const float byk = INVERSE ? s0 : o0, byk1 = INVERSE ? s1 : o1;

    const float wk = bxk1 - bxk;                              // :53
    const float hk = byk1 - byk;
    // delta = 1e-5 + softplus(.) = 1e-5 + ln 2 * log2(1 + exp2(.)), :77 (the logit is clamped so that exp2 stays
    // finite; softplus's own threshold of 20 only swaps in an approximation that this form does not need)
    const float LN2 = __int_as_float(0x3f317218);
    const float dk = fmaf(LN2, log2_f(1.0f + exp2_f(fminf(t0, 126.0f))), kRqsMinDelta);
    const float dk1 = fmaf(LN2, log2_f(1.0f + exp2_f(fminf(t1, 126.0f))), kRqsMinDelta);
    const float rw = rcp_n(wk);
    const float s = hk * rw;                                  // :94 / :159
    const float term1 = fmaf(-2.0f, s, dk1 + dk);            // :97 / :162
    float xi;
    if constexpr (!INVERSE) {
        xi = (v - bxk) * rw;                                  // :99
        xi = fminf(fmaxf(xi, 0.0f), 1.0f);                    // :100
    } else {
        const float term0 = v - byk;                          // :164
        const float term2 = hk * dk;                          // :165
        const float tt = term0 * term1;
        const float a = (hk * s - term2) + tt;                // :167
        const float b = term2 - tt;                           // :168
        const float c = (-s) * term0;                         // :169
        float r = fmaf(b, b, -(4.0f * a) * c);                // :171
        r = __builtin_sqrtf(fmaxf(r, 0.0f));
        xi = (2.0f * c) * rcp_f((-b) - r);                    // :173
        xi = fminf(fmaxf(xi, 0.0f), 1.0f);                    // :174
    }
    const float omx = 1.0f - xi;
    const float q = xi * omx;                                 // :101 / :175
    const float xi2 = xi * xi;
    const float den = fmaf(term1, q, s);                      // :105
    const float rden = rcp_n(den);
    if constexpr (!INVERSE) {
        const float num = hk * fmaf(dk, q, s * xi2);          // :104
        out = fmaf(num, rden, byk);                           // :106
    } else {
        out = fmaf(xi, wk, bxk);                              // :178
    }
    // :56-63  2 log s + log(inner) - 2 log(den) = log((s / den)^2 inner)
    const float inner = fmaf(dk1, xi2, fmaf(s + s, q, dk * (omx * omx)));
    const float r = s * rden;
    const float l = log2_f((r * r) * inner);
    l2 = INVERSE ? -l : l;
}


// Linear rational spline element (LinearRational, spline/linear_rational.py:9-182) in the same lean form.  p: 32
// pre-scaled parameters -- [0, 8) width logits, [8, 16) height logits (u_x + u_y / 100), [16, 24) MINUS the lambda logits,
// [24, 31) derivative logits (c + u_d / 100), [31] the w0 logit, all times log2(e).  C.d_edge = c log2(e) makes the
// boundary derivative 1e-5 + softplus(c) = 1 (the reference pads exactly 1).  l2 = log2 of the map's own log-det factor
// (the reference's inverse_1d returns the inverse's log-det directly: no negation here).
template <bool INVERSE, bool FAST = false>
__device__ __forceinline__ void lrs_eval_lean(const float (&p)[32], float v, const RqsLean &C, float &out, float &l2, float &amax)
{
    float X[9], Y[9], Dl[9];
    spline_knots<FAST>(p, C, X, amax);
    spline_knots<FAST>(p + 8, C, Y, amax);
    Dl[0] = C.d_edge; Dl[8] = C.d_edge;
#pragma unroll
    for (int j = 1; j < 8; ++j) Dl[j] = p[23 + j];
    // bin search (:53, searchsorted left), carrying knots, opposite knots, derivative logits and the bins' lambda logits
    float S5[5], O5[5], D5[5], L4[4];
    {
        const bool up = (INVERSE ? Y[4] : X[4]) < v;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            S5[i] = up ? (INVERSE ? Y[4 + i] : X[4 + i]) : (INVERSE ? Y[i] : X[i]);
            O5[i] = up ? (INVERSE ? X[4 + i] : Y[4 + i]) : (INVERSE ? X[i] : Y[i]);
            D5[i] = up ? Dl[4 + i] : Dl[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) L4[i] = up ? p[20 + i] : p[16 + i];
    }
    float S3[3], O3[3], D3[3], L2[2];
    {
        const bool up = S5[2] < v;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            S3[i] = up ? S5[2 + i] : S5[i];
            O3[i] = up ? O5[2 + i] : O5[i];
            D3[i] = up ? D5[2 + i] : D5[i];
        }
        L2[0] = up ? L4[2] : L4[0];
        L2[1] = up ? L4[3] : L4[1];
    }
    const bool up = S3[1] < v;
    const float s0 = up ? S3[1] : S3[0], s1 = up ? S3[2] : S3[1];
    const float o0 = up ? O3[1] : O3[0], o1 = up ? O3[2] : O3[1];
    const float t0 = up ? D3[1] : D3[0], t1 = up ? D3[2] : D3[1];
    const float ls = up ? L2[1] : L2[0];
    const float xk = INVERSE ? o0 : s0, xk1 = INVERSE ? o1 : s1;
    const float yk = INVERSE ? s0 : o0, yk1 = INVERSE ? s1 : o1;

    const float LN2 = __int_as_float(0x3f317218);
    const float lam = rcp_f(1.0f + exp2_f(ls));                                     // sigmoid, :57
    const float dk = fmaf(LN2, log2_f(1.0f + exp2_f(fminf(t0, 126.0f))), kRqsMinDelta);   // :56
    const float dk1 = fmaf(LN2, log2_f(1.0f + exp2_f(fminf(t1, 126.0f))), kRqsMinDelta);
    const float w0 = LN2 * log2_f(1.0f + exp2_f(fminf(p[31], 126.0f)));             // softplus(u_w0), :58
    const float wk = w0 * __builtin_amdgcn_rsqf(dk), wk1 = w0 * __builtin_amdgcn_rsqf(dk1);   // w0 sqrt(d_0 / d_j), d_0 = 1
    const float one_m = 1.0f - lam;
    const float a1 = one_m * wk, a2 = lam * wk1;
    const float ym = fmaf(a1, yk, a2 * yk1) * rcp_f(a1 + a2);                       // :68
    const float dx = xk1 - xk;
    const float wm = fmaf(lam * wk, dk, (one_m * wk1) * dk1) * (dx * rcp_f(yk1 - yk));      // :69
    const float n_lo = ((lam * wk) * wm) * (ym - yk), n_hi = ((one_m * wm) * wk1) * (yk1 - ym);
    if constexpr (!INVERSE) {
        const float phi = (v - xk) * rcp_f(dx);                                     // :74 (not clipped)
        const bool upper = phi > lam;
        const float wa = upper ? wm : wk, ya = upper ? ym : yk, ta = upper ? 1.0f - phi : lam - phi;
        const float wb = upper ? wk1 : wm, yb = upper ? yk1 : ym, tb = upper ? phi - lam : phi;
        const float den = fmaf(wa, ta, wb * tb);
        out = fmaf(wa * ya, ta, (wb * yb) * tb) * rcp_f(den);                       // :76 / :80
        l2 = log2_f((upper ? n_hi : n_lo) * rcp_f(fmaf(den, den, 5e-10f) * dx));    // :77 / :81-82
    } else {
        const bool upper = v > ym;
        const float wa = upper ? wk1 : wk, yv = upper ? yk1 - v : yk - v;
        const float vm = v - ym;
        const float den = fmaf(wa, yv, wm * vm);
        const float num = upper ? fmaf(lam * wa, yv, wm * vm) : (lam * wa) * yv;
        out = fmaf(num * rcp_f(den), dx, xk);                                       // :89 / :93
        l2 = log2_f(((upper ? n_hi : n_lo) * dx) * rcp_f(fmaf(den, den, 5e-10f)));  // :90 / :94-95
    }
}

// one layer with the roles of the planes fixed: src feeds the conditioner, tgt is transformed
template <int EPL, int BLOCK, int STEPS2, bool INVERSE>
__device__ __forceinline__ void rqs_layer(const float *__restrict__ gprm, float *stage, int lane, int q,
                                          const RqsLean &C, const float (&src)[EPL], float (&tgt)[EPL], float &ld2)
{
    constexpr int HALF = 4 * EPL;
    constexpr int HEAD = EPL * 64 + 16 + 2 * HALF;
    constexpr int NC = EPL / 8;
    float *head_s = stage;                                    // [HEAD]
    float *chunk_s = stage + HEAD;                            // [kRqsChunkFloats]
    float hid[4];
#pragma unroll 1
    for (int ch = 0; ch < NC; ++ch) {
        __syncthreads();                                      // every wave is done with the previous stage
        {
            const float4 *g4 = reinterpret_cast<const float4 *>(gprm + HEAD + (size_t)ch * kRqsChunkFloats);
            float4 *d4 = reinterpret_cast<float4 *>(chunk_s);
            for (int i = threadIdx.x; i < kRqsChunkFloats / 4; i += BLOCK) d4[i] = g4[i];
            if (ch == 0) {
                const float4 *h4 = reinterpret_cast<const float4 *>(gprm);
                float4 *e4 = reinterpret_cast<float4 *>(head_s);
                for (int i = threadIdx.x; i < HEAD / 4; i += BLOCK) e4[i] = h4[i];
            }
        }
        __syncthreads();
        if (ch == 0) {
            const cf32x4 *A1 = reinterpret_cast<const cf32x4 *>(head_s);
            const float *b1 = head_s + EPL * 64;
            const float *pre = b1 + 16;
            cf32x4 acc = *reinterpret_cast<const cf32x4 *>(b1 + 4 * q);
#pragma unroll
            for (int g = 0; g < EPL / 4; ++g) {
                const cf32x4 w = A1[g * 64 + lane];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], src[4 * g + k], acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {                // pending elementwise layers of the target plane
                const cf32x4 s = *reinterpret_cast<const cf32x4 *>(pre + EPL * q + 4 * i);
                const cf32x4 t = *reinterpret_cast<const cf32x4 *>(pre + HALF + EPL * q + 4 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) tgt[4 * i + k] = fmaf(s[k], tgt[4 * i + k], t[k]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) hid[r] = fmaf(-2.0f, rcp_f(exp2_f(acc[r]) + 1.0f), 1.0f);
        }
        const cf32x4 *A2 = reinterpret_cast<const cf32x4 *>(chunk_s);
        const float *b2 = chunk_s + 48 * 256;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float p[24];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int t = e * 6 + c;
                cf32x4 o = *reinterpret_cast<const cf32x4 *>(b2 + (t * 4 + q) * 4);
                const cf32x4 w = A2[t * 64 + lane];
#pragma unroll
                for (int k = 0; k < STEPS2; ++k)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], hid[k], o, 0, 0, 0);
                p[4 * c] = o[0]; p[4 * c + 1] = o[1]; p[4 * c + 2] = o[2]; p[4 * c + 3] = o[3];
            }
            // the element this (chunk, e) addresses: static when there is one chunk, else a select over the chunks
            float v = tgt[e];
            if constexpr (NC > 1) {
#pragma unroll
                for (int c2 = 1; c2 < NC; ++c2) v = (ch == c2) ? tgt[8 * c2 + e] : v;
            }
            float out = v, l = 0.0f;                           // spline/base.py:54-55
            if (v > C.minimum && v < C.maximum)                // strict box, base.py:29-33
                { float unused = 0.0f; rqs_eval_lean<INVERSE, false>(p, v, C, out, l, unused); }
            ld2 += l;
            if constexpr (NC > 1) {
#pragma unroll
                for (int c2 = 0; c2 < NC; ++c2) tgt[8 * c2 + e] = (ch == c2) ? out : tgt[8 * c2 + e];
            } else {
                tgt[e] = out;
            }
        }
    }
}


// The same layer with GEMM 2 on the bf16 matrix pipe at fp32 accuracy ("bf16 x 3"): every fp32 operand is the sum of
// three bf16 pieces of 8 mantissa bits each (truncation: hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = the
// rest -- exact), and of the nine piece products the six that matter (down to 2^-23 of |w h|) are kept:
//   W_hi h_hi + W_mid h_hi | W_hi h_mid + W_mid h_mid | W_lo h_hi + W_hi h_lo
// -- three v_mfma_f32_16x16x32_bf16 per tile (K = 32 = two 16-deep products side by side: slots 0..3 of a lane-group
// carry the first product's 4 hidden units, slots 4..7 the second's), ~17.7 cycles each, instead of four
// v_mfma_f32_16x16x4_f32 of 32 cycles: 319 instead of 768 matrix cycles per spline element.  The weights are split by
// the packer; the 4 hidden activations of a lane are split once per layer (16 + 6 vector instructions).
// HT = 16-unit tiles of the hidden layer (hidden width <= 16 HT - 1: the last unit carries b2); a chunk holds
// 4 / HT target elements, so its size does not depend on HT.
// Block: head A1[EPL/4][HT][64][4] | b1[HT][4][4] | pre_s | pre_t, then chunks A[4/HT][6][HT][2][64][4 dwords].
// LRS: linear rational spline elements (32 parameters = 8 tiles per element) instead of rational-quadratic ones (24 = 6).
// CTX: the conditioner also reads the context (Concatenation, conditioning/context.py:38-64: [x_A | context]): its columns
// of W1 are up to 4 further k-steps of GEMM 1 -- A1c[HT][64][4] behind the head, the lane's context elements 4 s + q in cx.
template <int EPL, int BLOCK, int HT, bool INVERSE, bool LRS, bool FAST, bool CTX = false>
__device__ __forceinline__ void rqs_layer3(const float *__restrict__ gprm, float *stage, int lane, int q,
                                           const RqsLean &C, const float (&src)[EPL], float (&tgt)[EPL], float &ld2,
                                           float &amax, const float (&cx)[4] = {0.0f, 0.0f, 0.0f, 0.0f}, int cs = 0)
{
    constexpr int HALF = 4 * EPL;
    constexpr int HEAD = EPL * HT * 64 + HT * 16 + 2 * HALF + (CTX ? HT * 256 : 0);
    constexpr int TPE = LRS ? 8 : 6;                          // tiles of 4 parameters per element
    constexpr int GRP = LRS ? 4 : 3;                          // tiles side by side (TPE / 2)
    constexpr int CHUNKD = (LRS ? 4 : 3) * 4096;              // = (4 / HT) * TPE * HT * 2 * 64 * 4 dwords
    constexpr int ELEMS = 4 / HT;
    constexpr int NC = EPL / ELEMS;
    constexpr bool STATIC_CH = NC <= 8;                       // chunk loop unrolled: the element index is static
    float *head_s = stage;
    float *chunk_s = stage + HEAD;
    ci32x4 B1[HT], B2[HT], B3[HT];                            // [h_hi | h_hi], [h_mid | h_mid], [h_hi | h_lo] per hidden tile

    auto stage_in = [&](int ch) {
        __syncthreads();
        const float4 *g4 = reinterpret_cast<const float4 *>(gprm + HEAD + (size_t)ch * CHUNKD);
        float4 *d4 = reinterpret_cast<float4 *>(chunk_s);
        for (int i = threadIdx.x; i < CHUNKD / 4; i += BLOCK) d4[i] = g4[i];
        if (ch == 0) {
            const float4 *h4 = reinterpret_cast<const float4 *>(gprm);
            float4 *e4 = reinterpret_cast<float4 *>(head_s);
            for (int i = threadIdx.x; i < HEAD / 4; i += BLOCK) e4[i] = h4[i];
        }
        __syncthreads();
    };
    auto gemm1 = [&]() {
        const cf32x4 *A1 = reinterpret_cast<const cf32x4 *>(head_s);
        const float *b1 = head_s + EPL * HT * 64;
        const float *pre = b1 + HT * 16;
        cf32x4 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = *reinterpret_cast<const cf32x4 *>(b1 + t * 16 + 4 * q);
#pragma unroll
        for (int g = 0; g < EPL / 4; ++g)
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const cf32x4 w = A1[(g * HT + t) * 64 + lane];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], src[4 * g + k], acc[t], 0, 0, 0);
            }
        if constexpr (CTX) {
            const cf32x4 *A1c = reinterpret_cast<const cf32x4 *>(pre + 2 * HALF);
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const cf32x4 w = A1c[t * 64 + lane];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cs) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], cx[k], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const cf32x4 s = *reinterpret_cast<const cf32x4 *>(pre + EPL * q + 4 * i);
            const cf32x4 t = *reinterpret_cast<const cf32x4 *>(pre + HALF + EPL * q + 4 * i);
#pragma unroll
            for (int k = 0; k < 4; ++k) tgt[4 * i + k] = fmaf(s[k], tgt[4 * i + k], t[k]);
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            int hi[4], mid[4], lo[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h = fmaf(-2.0f, rcp_f(exp2_f(acc[t][r]) + 1.0f), 1.0f);
                // the last hidden unit (tile HT - 1, lane-group 3, register 3) is the constant 1 that carries b2 through
                // GEMM 2: no bias reads
                if (t == HT - 1 && r == 3) h = (q == 3) ? 1.0f : h;
                const int hb = __float_as_int(h) & (int)0xffff0000;
                const float r1 = h - __int_as_float(hb);                      // exact
                const int mb = __float_as_int(r1) & (int)0xffff0000;
                const float r2 = r1 - __int_as_float(mb);                     // exact; its top 16 bits are the third piece
                hi[r] = hb; mid[r] = mb; lo[r] = __float_as_int(r2);
            }
            // two bf16 (= the upper halves) per dword: bytes {src1[2], src1[3], src0[2], src0[3]}
            const int hh01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302), hh23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302);
            const int mm01 = __builtin_amdgcn_perm(mid[1], mid[0], 0x07060302), mm23 = __builtin_amdgcn_perm(mid[3], mid[2], 0x07060302);
            const int ll01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302), ll23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302);
            B1[t] = ci32x4{hh01, hh23, hh01, hh23};
            B2[t] = ci32x4{mm01, mm23, mm01, mm23};
            B3[t] = ci32x4{hh01, hh23, ll01, ll23};
        }
    };
    auto chunk = [&](int ch) {
        const ci32x4 *A = reinterpret_cast<const ci32x4 *>(chunk_s);          // [elem][6][HT][2][64]
#pragma unroll
        for (int e = 0; e < ELEMS; ++e) {
            // the tiles of an element side by side, product by product: a bf16 MFMA issues in ~16 cycles but its result
            // takes longer, so the MFMAs of ONE tile must not follow each other (back to back they ran at ~36 cycles
            // each: measured, the first version of this loop); groups of three tiles = 36 operand / accumulator registers
            float p[4 * TPE];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                cf32x4 o[GRP];
#pragma unroll
                for (int c = 0; c < GRP; ++c) o[c] = cf32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    ci32x4 a1[GRP];
#pragma unroll
                    for (int c = 0; c < GRP; ++c) a1[c] = A[(((e * TPE + GRP * g + c) * HT + t) * 2) * 64 + lane];
#pragma unroll
                    for (int c = 0; c < GRP; ++c)
                        o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8, a1[c]), __builtin_bit_cast(cbf16x8, B1[t]), o[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < GRP; ++c)
                        o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8, a1[c]), __builtin_bit_cast(cbf16x8, B2[t]), o[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < GRP; ++c) {
                        const ci32x4 a2 = A[(((e * TPE + GRP * g + c) * HT + t) * 2 + 1) * 64 + lane];
                        o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8, a2), __builtin_bit_cast(cbf16x8, B3[t]), o[c], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int c = 0; c < GRP; ++c) {
                    p[4 * GRP * g + 4 * c] = o[c][0]; p[4 * GRP * g + 4 * c + 1] = o[c][1];
                    p[4 * GRP * g + 4 * c + 2] = o[c][2]; p[4 * GRP * g + 4 * c + 3] = o[c][3];
                }
            }
            float v;
            if constexpr (STATIC_CH) {
                v = tgt[ELEMS * ch + e];
            } else {                                           // run-time chunk index: a select over the chunks
                v = tgt[e];
#pragma unroll
                for (int c2 = 1; c2 < NC; ++c2) v = (ch == c2) ? tgt[ELEMS * c2 + e] : v;
            }
            float out = v, l = 0.0f;
            if (v > C.minimum && v < C.maximum) {
                if constexpr (LRS) lrs_eval_lean<INVERSE, FAST>(p, v, C, out, l, amax);
                else rqs_eval_lean<INVERSE, FAST>(p, v, C, out, l, amax);
            }
            ld2 += l;
            if constexpr (STATIC_CH) {
                tgt[ELEMS * ch + e] = out;
            } else {
#pragma unroll
                for (int c2 = 0; c2 < NC; ++c2) tgt[ELEMS * c2 + e] = (ch == c2) ? out : tgt[ELEMS * c2 + e];
            }
        }
    };
    if constexpr (STATIC_CH) {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            stage_in(ch);
            if (ch == 0) gemm1();
            chunk(ch);
        }
    } else {
#pragma unroll 1
        for (int ch = 0; ch < NC; ++ch) {
            stage_in(ch);
            if (ch == 0) gemm1();
            chunk(ch);
        }
    }
}

// MADE-based spline layer, parallel map (MaskedAutoregressiveBijection.forward with a spline transformer,
// layers_base.py:201-206; MADE = two masked linear layers, transforms.py:184-267, masks folded into the packed weights):
// GEMM 1 reads BOTH planes as they are, every element of both planes then takes its pending elementwise layers and is
// transformed with parameters that depend on the preceding elements only.  Head: A1[2 EPL / 4][HT][64][4] (plane A's
// k-steps, then plane B's) | b1[HT][4][4] | pre_s[D] | pre_t[D]; then 2 EPL HT / 4 chunks, plane A's elements first.
template <int EPL, int BLOCK, int HT, bool INVERSE, bool LRS, bool FAST>
__device__ __forceinline__ void rqs_made_layer3(const float *__restrict__ gprm, float *stage, int lane, int q,
                                                const RqsLean &C, float (&pa)[EPL], float (&pb)[EPL], float &ld2,
                                                float &amax)
{
    constexpr int HALF = 4 * EPL, D = 8 * EPL;
    constexpr int HEAD = 2 * EPL * HT * 64 + HT * 16 + 2 * D;
    constexpr int TPE = LRS ? 8 : 6;                          // tiles of 4 parameters per element
    constexpr int GRP = LRS ? 4 : 3;                          // tiles side by side (TPE / 2)
    constexpr int CHUNKD = (LRS ? 4 : 3) * 4096;              // = (4 / HT) * TPE * HT * 2 * 64 * 4 dwords
    constexpr int ELEMS = 4 / HT;
    constexpr int NC = EPL / ELEMS;
    constexpr bool STATIC_CH = NC <= 8;                       // chunk loop unrolled: the element index is static
    float *head_s = stage;
    float *chunk_s = stage + HEAD;
    ci32x4 B1[HT], B2[HT], B3[HT];                            // [h_hi | h_hi], [h_mid | h_mid], [h_hi | h_lo] per hidden tile

    auto stage_in = [&](int ch) {
        __syncthreads();
        const float4 *g4 = reinterpret_cast<const float4 *>(gprm + HEAD + (size_t)ch * CHUNKD);
        float4 *d4 = reinterpret_cast<float4 *>(chunk_s);
        for (int i = threadIdx.x; i < CHUNKD / 4; i += BLOCK) d4[i] = g4[i];
        if (ch == 0) {
            const float4 *h4 = reinterpret_cast<const float4 *>(gprm);
            float4 *e4 = reinterpret_cast<float4 *>(head_s);
            for (int i = threadIdx.x; i < HEAD / 4; i += BLOCK) e4[i] = h4[i];
        }
        __syncthreads();
    };
    auto gemm1 = [&]() {
        const cf32x4 *A1 = reinterpret_cast<const cf32x4 *>(head_s);
        const float *b1 = head_s + 2 * EPL * HT * 64;
        const float *pre = b1 + HT * 16;
        cf32x4 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = *reinterpret_cast<const cf32x4 *>(b1 + t * 16 + 4 * q);
#pragma unroll
        for (int g = 0; g < EPL / 4; ++g)
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const cf32x4 wa = A1[(g * HT + t) * 64 + lane], wb = A1[((EPL / 4 + g) * HT + t) * 64 + lane];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k], pa[4 * g + k], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[k], pb[4 * g + k], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const cf32x4 sa = *reinterpret_cast<const cf32x4 *>(pre + EPL * q + 4 * i);
            const cf32x4 sb = *reinterpret_cast<const cf32x4 *>(pre + HALF + EPL * q + 4 * i);
            const cf32x4 ta = *reinterpret_cast<const cf32x4 *>(pre + D + EPL * q + 4 * i);
            const cf32x4 tb = *reinterpret_cast<const cf32x4 *>(pre + D + HALF + EPL * q + 4 * i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pa[4 * i + k] = fmaf(sa[k], pa[4 * i + k], ta[k]);
                pb[4 * i + k] = fmaf(sb[k], pb[4 * i + k], tb[k]);
            }
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            int hi[4], mid[4], lo[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h = fmaf(-2.0f, rcp_f(exp2_f(acc[t][r]) + 1.0f), 1.0f);
                // the last hidden unit (tile HT - 1, lane-group 3, register 3) is the constant 1 that carries b2 through
                // GEMM 2: no bias reads
                if (t == HT - 1 && r == 3) h = (q == 3) ? 1.0f : h;
                const int hb = __float_as_int(h) & (int)0xffff0000;
                const float r1 = h - __int_as_float(hb);                      // exact
                const int mb = __float_as_int(r1) & (int)0xffff0000;
                const float r2 = r1 - __int_as_float(mb);                     // exact; its top 16 bits are the third piece
                hi[r] = hb; mid[r] = mb; lo[r] = __float_as_int(r2);
            }
            // two bf16 (= the upper halves) per dword: bytes {src1[2], src1[3], src0[2], src0[3]}
            const int hh01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302), hh23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302);
            const int mm01 = __builtin_amdgcn_perm(mid[1], mid[0], 0x07060302), mm23 = __builtin_amdgcn_perm(mid[3], mid[2], 0x07060302);
            const int ll01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302), ll23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302);
            B1[t] = ci32x4{hh01, hh23, hh01, hh23};
            B2[t] = ci32x4{mm01, mm23, mm01, mm23};
            B3[t] = ci32x4{hh01, hh23, ll01, ll23};
        }
    };
    auto chunk = [&](float (&tgt)[EPL], int ch) {
        const ci32x4 *A = reinterpret_cast<const ci32x4 *>(chunk_s);          // [elem][6][HT][2][64]
#pragma unroll
        for (int e = 0; e < ELEMS; ++e) {
            // the tiles of an element side by side, product by product: a bf16 MFMA issues in ~16 cycles but its result
            // takes longer, so the MFMAs of ONE tile must not follow each other (back to back they ran at ~36 cycles
            // each: measured, the first version of this loop); groups of three tiles = 36 operand / accumulator registers
            float p[4 * TPE];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                cf32x4 o[GRP];
#pragma unroll
                for (int c = 0; c < GRP; ++c) o[c] = cf32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    ci32x4 a1[GRP];
#pragma unroll
                    for (int c = 0; c < GRP; ++c) a1[c] = A[(((e * TPE + GRP * g + c) * HT + t) * 2) * 64 + lane];
#pragma unroll
                    for (int c = 0; c < GRP; ++c)
                        o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8, a1[c]), __builtin_bit_cast(cbf16x8, B1[t]), o[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < GRP; ++c)
                        o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8, a1[c]), __builtin_bit_cast(cbf16x8, B2[t]), o[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < GRP; ++c) {
                        const ci32x4 a2 = A[(((e * TPE + GRP * g + c) * HT + t) * 2 + 1) * 64 + lane];
                        o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8, a2), __builtin_bit_cast(cbf16x8, B3[t]), o[c], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int c = 0; c < GRP; ++c) {
                    p[4 * GRP * g + 4 * c] = o[c][0]; p[4 * GRP * g + 4 * c + 1] = o[c][1];
                    p[4 * GRP * g + 4 * c + 2] = o[c][2]; p[4 * GRP * g + 4 * c + 3] = o[c][3];
                }
            }
            float v;
            if constexpr (STATIC_CH) {
                v = tgt[ELEMS * ch + e];
            } else {                                           // run-time chunk index: a select over the chunks
                v = tgt[e];
#pragma unroll
                for (int c2 = 1; c2 < NC; ++c2) v = (ch == c2) ? tgt[ELEMS * c2 + e] : v;
            }
            float out = v, l = 0.0f;
            if (v > C.minimum && v < C.maximum) {
                if constexpr (LRS) lrs_eval_lean<INVERSE, FAST>(p, v, C, out, l, amax);
                else rqs_eval_lean<INVERSE, FAST>(p, v, C, out, l, amax);
            }
            ld2 += l;
            if constexpr (STATIC_CH) {
                tgt[ELEMS * ch + e] = out;
            } else {
#pragma unroll
                for (int c2 = 0; c2 < NC; ++c2) tgt[ELEMS * c2 + e] = (ch == c2) ? out : tgt[ELEMS * c2 + e];
            }
        }
    };
    if constexpr (STATIC_CH) {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            stage_in(ch);
            if (ch == 0) gemm1();
            chunk(pa, ch);
        }
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            stage_in(NC + ch);
            chunk(pb, ch);
        }
    } else {
#pragma unroll 1
        for (int ch = 0; ch < NC; ++ch) {
            stage_in(ch);
            if (ch == 0) gemm1();
            chunk(pa, ch);
        }
#pragma unroll 1
        for (int ch = 0; ch < NC; ++ch) {
            stage_in(NC + ch);
            chunk(pb, ch);
        }
    }
}

#ifndef TFK_RQS3_WAVES
#define TFK_RQS3_WAVES 4
#endif
template <int EPL, int BLOCK, int STEPS2, bool INVERSE, bool CTX = false, bool MADE = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((STEPS2 == 0 || STEPS2 >= 8) ? (EPL == 32 ? 2 : TFK_RQS3_WAVES) : 1)))
void k_flow_rqs_chain(
    const float *__restrict__ x, float *z, float *logdet, const float *__restrict__ gauss_loc,
    const float *__restrict__ gauss_log_scale, float *logprob, long long N,
    const float *__restrict__ params, RqsChainProg prog, int flags, int xw,
    const float *__restrict__ context = nullptr, int Cn = 0)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = 8 * EPL, HALF = 4 * EPL;
    constexpr bool F3 = STEPS2 == 0 || STEPS2 >= 8;
    constexpr bool LRS = STEPS2 >= 16;
    constexpr int HT3 = (STEPS2 == 8 || STEPS2 == 24) ? 2 : 1;
    static_assert(!CTX || (STEPS2 == 0 || STEPS2 >= 8), "context-conditioned spline chains: bf16 x 3 operand format");
    static_assert(!MADE || ((STEPS2 == 0 || STEPS2 >= 8) && !CTX), "MADE spline layers: bf16 x 3 operand format, no context");
    constexpr int HEAD = MADE ? 2 * EPL * HT3 * 64 + HT3 * 16 + 2 * D
                              : (F3 ? EPL * HT3 * 64 + HT3 * 16 + 2 * HALF : EPL * 64 + 16 + 2 * HALF) + (CTX ? HT3 * 256 : 0);
    const int accumulate = flags & 1;
    const bool reverse_out = (flags & 2) != 0;
    const bool base_of_input = (flags & 4) != 0;
    constexpr int CHUNK = F3 ? (LRS ? 16384 : kRqsChunk3Dwords) : kRqsChunkFloats;
    float *stage = lds;                                      // [HEAD + chunk]
    float *side_s = lds + HEAD + CHUNK;                      // (context programs) the elementwise ops' blocks, in op order
    float *ew_s = side_s + (CTX ? prog.side_floats : 0);     // s[D] | t[D] | ldc, pad[3]
    float *base_s = ew_s + 2 * D + 4;                        // loc[D] | 1/scale[D] | const
    constexpr bool SIDE = CTX && EPL <= 16;                  // elementwise ops inside the launch (D <= 128)
    if constexpr (SIDE) {
#pragma unroll 1
        for (int i = 0; i < 2 * kChainSideOps; ++i) {
            const int kind = i < kChainSideOps ? prog.pre_kind[i] : prog.post_kind[i - kChainSideOps];
            const int off = i < kChainSideOps ? prog.pre_off[i] : prog.post_off[i - kChainSideOps];
            const int at = i < kChainSideOps ? prog.pre_lds[i] : prog.post_lds[i - kChainSideOps];
            const int len = kind == 0 ? 0 : (kind == 1 ? 2 * D + 4 : EPL * prog.ctx_steps * 64 + EPL * 16);
            for (int e = threadIdx.x; e < len; e += BLOCK) side_s[at + e] = params[off + e];
        }
    }
    if (prog.ew_offset >= 0)
        for (int e = threadIdx.x; e < 2 * D + 4; e += BLOCK) ew_s[e] = params[prog.ew_offset + e];
    if (logprob) {
        for (int e = threadIdx.x; e < D; e += BLOCK) {
            base_s[e] = gauss_loc[e];
            base_s[D + e] = expf(-gauss_log_scale[e]);
        }
        if (threadIdx.x < 64) {
            float c = 0.0f;
            for (int e = threadIdx.x; e < D; e += 64) c += gauss_log_scale[e] + kHalfLog2Pi;
            c = group_sum(c, 64);
            if (threadIdx.x == 0) base_s[2 * D] = c;
        }
    }
    __syncthreads();
    const RqsLean C = prog.C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    constexpr int rows_per_block = (BLOCK / 64) * 16;
    const long long stride = (long long)gridDim.x * rows_per_block;
    const float base_const = logprob ? base_s[2 * D] : 0.0f;
    // (tfk_flow_run_mfma_sum) fp64 sum of the log-probabilities this thread wrote, kept in the LDS (see tfk_flow_chain.h)
    double *lp_slot = reinterpret_cast<double *>(base_s + 2 * D + 4) + threadIdx.x;
    if (prog.sum_ws) *lp_slot = 0.0;
    // the trip count is the same for every wave of a workgroup (barriers inside): loop on the workgroup's first row
    for (long long blk0 = (long long)blockIdx.x * rows_per_block; blk0 < N; blk0 += stride) {
        const long long row = blk0 + wave * 16 + j;
        const long long rr = row < N ? row : N - 1;
        float a[EPL], b[EPL];
        auto load_rows = [&]() {
        if (xw == D) {
            const float4 *pa = reinterpret_cast<const float4 *>(x + rr * D + EPL * q);
            const float4 *pb = reinterpret_cast<const float4 *>(x + rr * D + HALF + EPL * q);
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                const float4 va = pa[i], vb = pb[i];
                a[4 * i] = va.x; a[4 * i + 1] = va.y; a[4 * i + 2] = va.z; a[4 * i + 3] = va.w;
                b[4 * i] = vb.x; b[4 * i + 1] = vb.y; b[4 * i + 2] = vb.z; b[4 * i + 3] = vb.w;
            }
        } else {
            // rows narrower than the kernel's planes (event sizes that are not 64 / 128 / 256): the caller's rows are
            // read as they are -- first half into the head of plane A, second half into the head of plane B, zeros
            // behind them (the padding is an exact identity by construction of the weights, fused.py)
            // (an ODD width: hl sources, the middle element -- into plane B's last column --, hl targets; tfk_flow_chain.h)
            const int hl = xw >> 1, odd = xw & 1;
            const float *xr = x + rr * xw;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int col = EPL * q + e;
                const bool ok = col < hl;
                a[e] = ok ? xr[col] : 0.0f;
                b[e] = ok ? xr[hl + odd + col] : 0.0f;
            }
            if (odd && q == 3) b[EPL - 1] = xr[hl];
        }
        };
        load_rows();
        float cx[4] = {0.0f, 0.0f, 0.0f, 0.0f};                      // this lane's context elements 4 s + q of row j
        if constexpr (CTX) {
#pragma unroll
            for (int s = 0; s < 4; ++s) cx[s] = (4 * s + q < Cn) ? context[rr * Cn + 4 * s + q] : 0.0f;
        }
        float ld = (q == 0 && logdet && accumulate) ? logdet[rr] : 0.0f;
        float sq = 0.0f;
        auto base_terms = [&]() {
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                const cf32x4 la = *reinterpret_cast<const cf32x4 *>(base_s + EPL * q + 4 * i);
                const cf32x4 lb = *reinterpret_cast<const cf32x4 *>(base_s + HALF + EPL * q + 4 * i);
                const cf32x4 ia = *reinterpret_cast<const cf32x4 *>(base_s + D + EPL * q + 4 * i);
                const cf32x4 ib = *reinterpret_cast<const cf32x4 *>(base_s + D + HALF + EPL * q + 4 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ta = (a[4 * i + k] - la[k]) * ia[k];
                    const float tb = (b[4 * i + k] - lb[k]) * ib[k];
                    sq = fmaf(ta, ta, sq);
                    sq = fmaf(tb, tb, sq);
                }
            }
        };
        if (logprob && base_of_input) base_terms();
        float ld2 = 0.0f;
        float amax = 0.0f;                                            // largest |softmax logit| seen (spline_knots)
        auto layers = [&](auto fast) {
            constexpr bool FAST = decltype(fast)::value;
#pragma unroll 1
            for (int l = 0; l < prog.n_layers; ++l) {
                const float *gprm = params + prog.offset0 + (size_t)l * prog.layer_stride;
                if constexpr (MADE) {
                    rqs_made_layer3<EPL, BLOCK, HT3, INVERSE, LRS, FAST>(gprm, stage, lane, q, C, a, b, ld2, amax);
                } else if constexpr (F3) {
                    if (((prog.first_src + l) & 1) == 0) {
                        if ((prog.move_mask >> l) & 1ull) move_middle<EPL>(q, b, a);      // (odd event sizes)
                        rqs_layer3<EPL, BLOCK, HT3, INVERSE, LRS, FAST, CTX>(gprm, stage, lane, q, C, a, b, ld2, amax, cx, prog.ctx_steps);
                    } else {
                        if ((prog.move_mask >> l) & 1ull) move_middle<EPL>(q, a, b);
                        rqs_layer3<EPL, BLOCK, HT3, INVERSE, LRS, FAST, CTX>(gprm, stage, lane, q, C, b, a, ld2, amax, cx, prog.ctx_steps);
                    }
                } else {
                    if (((prog.first_src + l) & 1) == 0) {
                        if ((prog.move_mask >> l) & 1ull) move_middle<EPL>(q, b, a);
                        rqs_layer<EPL, BLOCK, STEPS2, INVERSE>(gprm, stage, lane, q, C, a, b, ld2);
                    } else {
                        if ((prog.move_mask >> l) & 1ull) move_middle<EPL>(q, a, b);
                        rqs_layer<EPL, BLOCK, STEPS2, INVERSE>(gprm, stage, lane, q, C, b, a, ld2);
                    }
                }
            }
        };
        float ld_pre = 0.0f;                                          // (context programs) log-det of the ops in front
        auto pre_ops = [&]() {
            if constexpr (SIDE) {
                ld_pre = 0.0f;
#pragma unroll 1
                for (int i = 0; i < kChainSideOps; ++i)
                    if (prog.pre_kind[i]) side_op<EPL>(prog.pre_kind[i], side_s + prog.pre_lds[i], prog.ctx_steps, lane, q, a, b, ld_pre, cx);
            }
        };
        pre_ops();
        if constexpr (F3 && TFK_SOFTMAX_FAST) {
            layers(std::true_type{});
            if (__syncthreads_or(!(amax <= kSoftmaxFastMax))) {       // (also catches a NaN logit)
                load_rows();                                          // x is intact: z is stored below
                ld2 = 0.0f;
                pre_ops();
                layers(std::false_type{});
            }
        } else {
            layers(std::false_type{});
        }
        ld = fmaf(ld2, __int_as_float(0x3f317218), ld);             // ln 2
        if constexpr (SIDE) ld += ld_pre;
        if (prog.ew_offset >= 0) {
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                const cf32x4 sa = *reinterpret_cast<const cf32x4 *>(ew_s + EPL * q + 4 * i);
                const cf32x4 sb = *reinterpret_cast<const cf32x4 *>(ew_s + HALF + EPL * q + 4 * i);
                const cf32x4 ta = *reinterpret_cast<const cf32x4 *>(ew_s + D + EPL * q + 4 * i);
                const cf32x4 tb = *reinterpret_cast<const cf32x4 *>(ew_s + D + HALF + EPL * q + 4 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[4 * i + k] = fmaf(sa[k], a[4 * i + k], ta[k]);
                    b[4 * i + k] = fmaf(sb[k], b[4 * i + k], tb[k]);
                }
            }
            if (q == 0) ld = ld + ew_s[2 * D];
        }
        if constexpr (SIDE) {                                         // the elementwise layers behind the couplings
#pragma unroll 1
            for (int i = 0; i < kChainSideOps; ++i)
                if (prog.post_kind[i])
                    side_op<EPL>(prog.post_kind[i], side_s + prog.post_lds[i], prog.ctx_steps, lane, q, a, b, ld, cx);
        }
        if (logprob && !base_of_input) base_terms();
        ld += __shfl_xor(ld, 16, kWave);
        ld += __shfl_xor(ld, 32, kWave);
        if (logprob) {
            sq += __shfl_xor(sq, 16, kWave);
            sq += __shfl_xor(sq, 32, kWave);
        }
        if (row < N) {
            if (z && !reverse_out) {
                float4 *qa = reinterpret_cast<float4 *>(z + row * D + EPL * q);
                float4 *qb = reinterpret_cast<float4 *>(z + row * D + HALF + EPL * q);
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    qa[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
                    qb[i] = make_float4(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]);
                }
            } else if (z) {
                float4 *qa = reinterpret_cast<float4 *>(z + row * D + D - EPL * (q + 1));
                float4 *qb = reinterpret_cast<float4 *>(z + row * D + HALF - EPL * (q + 1));
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    qa[i] = make_float4(a[EPL - 1 - 4 * i], a[EPL - 2 - 4 * i], a[EPL - 3 - 4 * i], a[EPL - 4 - 4 * i]);
                    qb[i] = make_float4(b[EPL - 1 - 4 * i], b[EPL - 2 - 4 * i], b[EPL - 3 - 4 * i], b[EPL - 4 - 4 * i]);
                }
            }
            if (q == 0) {
                if (logdet) logdet[row] = ld;
                if (logprob) {
                    const float lp = (fmaf(-0.5f, sq, -base_const)) + ld;
                    logprob[row] = lp;
                    if (prog.sum_ws) *lp_slot += (double)lp;
                }
            }
        }
    }
    if (prog.sum_ws) {
        const double mine = (q == 0) ? *lp_slot : 0.0;
        finish_sum_f64<BLOCK>(mine, reinterpret_cast<double *>(lds), prog.sum_ws, prog.sum_out);
    }
}

template <int EPL, int BLOCK, int STEPS2, bool INVERSE, bool CTX = false, bool MADE = false>
static int launch_rqs_chain_b(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                              float *logprob, int64_t N, const float *params, const RqsChainProg &prog, int flags,
                              int xw, hipStream_t s, const char *fn, const float *context = nullptr, int Cn = 0)
{
    constexpr int D = 8 * EPL, HALF = 4 * EPL;
    constexpr bool F3 = STEPS2 == 0 || STEPS2 >= 8;
    constexpr bool LRS = STEPS2 >= 16;
    constexpr int HT3 = (STEPS2 == 8 || STEPS2 == 24) ? 2 : 1;
    constexpr int HEAD = MADE ? 2 * EPL * HT3 * 64 + HT3 * 16 + 2 * D
                              : (F3 ? EPL * HT3 * 64 + HT3 * 16 + 2 * HALF : EPL * 64 + 16 + 2 * HALF) + (CTX ? HT3 * 256 : 0);
    const size_t lds = ((size_t)HEAD + (F3 ? (LRS ? 16384 : kRqsChunk3Dwords) : kRqsChunkFloats) + 2 * (2 * D + 4)
                        + (CTX ? prog.side_floats : 0)) * sizeof(float)
                       + (prog.sum_ws ? (size_t)BLOCK * sizeof(double) : 0);
    auto kern = &k_flow_rqs_chain<EPL, BLOCK, STEPS2, INVERSE, CTX, MADE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", fn, lds, hipGetErrorString(e));
        }
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    if (per_cu > 8) per_cu = 8;                              // (tfk_flow_sum_workspace_bytes counts on it)
    constexpr int rows_per_block = (BLOCK / 64) * 16;
    const int64_t want = (N + rows_per_block - 1) / rows_per_block;
    const int64_t cap = (int64_t)cu_count() * per_cu * TFK_CHAIN_OVERSUB;
    const int grid = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, s, x, z, logdet, loc, log_scale, logprob, (long long)N,
                       params, prog, flags, xw, context, Cn);
    return check_launch(fn);
}

template <int EPL>
static int launch_rqs_chain(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                            float *logprob, int64_t N, const float *params, const RqsChainProg &prog, int inverse,
                            int steps2, int flags, int xw, hipStream_t s, const char *fn,
                            const float *context = nullptr, int Cn = 0)
{
    if (prog.made) {                                         // MADE spline layers: bf16 x 3 operands, hidden <= 15, D = 64 / 128
        if constexpr (EPL == 8 || EPL == 16) {
#define TFK_RCM(ST_) \
    (inverse ? launch_rqs_chain_b<EPL, 512, ST_, true, false, true>(x, z, logdet, loc, log_scale, logprob, N, params, prog, flags, xw, s, fn) \
             : launch_rqs_chain_b<EPL, 512, ST_, false, false, true>(x, z, logdet, loc, log_scale, logprob, N, params, prog, flags, xw, s, fn))
            switch (steps2) {
            case 0: return TFK_RCM(0);
            case 16: return TFK_RCM(16);
            default: break;
            }
#undef TFK_RCM
        }
        return fail(TFK_EINVAL, "%s: lean MADE spline layers need D = 64 or 128 and hidden width <= 15", fn);
    }
    if (context) {                                           // context-conditioned chains: bf16 x 3 operands, D >= 64
        if constexpr (EPL >= 8) {
#define TFK_RCC(ST_) \
    (inverse ? launch_rqs_chain_b<EPL, 512, ST_, true, true>(x, z, logdet, loc, log_scale, logprob, N, params, prog, flags, xw, s, fn, context, Cn) \
             : launch_rqs_chain_b<EPL, 512, ST_, false, true>(x, z, logdet, loc, log_scale, logprob, N, params, prog, flags, xw, s, fn, context, Cn))
            switch (steps2) {
            case 0: return TFK_RCC(0);
            case 8: return TFK_RCC(8);
            case 16: return TFK_RCC(16);
            case 24: return TFK_RCC(24);
            default: break;
            }
#undef TFK_RCC
        }
        return fail(TFK_EINVAL, "%s: context-conditioned lean spline chains need D >= 64 and the bf16 x 3 operand format", fn);
    }
#define TFK_RC(ST_) \
    (inverse ? launch_rqs_chain_b<EPL, 512, ST_, true>(x, z, logdet, loc, log_scale, logprob, N, params, prog, flags, xw, s, fn) \
             : launch_rqs_chain_b<EPL, 512, ST_, false>(x, z, logdet, loc, log_scale, logprob, N, params, prog, flags, xw, s, fn))
    switch (steps2) {
    case 0: return TFK_RC(0);                                // bf16 x 3 operands, hidden width <= 15
    case 8: return TFK_RC(8);                                // bf16 x 3 operands, hidden width <= 31
    case 16: return TFK_RC(16);                              // linear rational spline, hidden width <= 15
    case 24: return TFK_RC(24);                              // linear rational spline, hidden width <= 31
    default: break;
    }
    if constexpr (EPL >= 8) {                                // fp32 operands: chunks of 8 elements per lane group
        switch (steps2) {
        case 1: return TFK_RC(1);
        case 2: return TFK_RC(2);
        case 3: return TFK_RC(3);
        case 4: return TFK_RC(4);
        default: break;
        }
    }
    return fail(TFK_EINVAL, "%s: lean spline couplings: %d GEMM-2 steps (fp32 operands: 1..4, D >= 64)", fn, steps2);
#undef TFK_RC
}

}  // namespace tfk
