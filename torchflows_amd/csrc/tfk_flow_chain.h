// tfk_flow_chain.h -- straight-line matrix-core kernel for CHAINS of affine / shift couplings
// (templates; instantiated per row width in tfk_flow_chain_{8,16,32}.hip, dispatched from tfk_flow_run_mfma).
//
// The interpreter of tfk_flow_mfma.h spends most of its vector instructions on things that are not the
// transform (rocprofv3 + the ISA of k_flow_run_mfma<8,512,1,false>, profiles/r02/isa/): per wave and layer of
// RealNVP-64, ~100 for the transform's exp / log in their range-reduced forms, 32 for the ActNorm that follows
// every coupling, ~25 register moves where the two planes change roles, ~10 v_cndmask / v_cmp for the run-time
// GEMM-2 step count, plus the dispatch.  This kernel removes them for the programs that are nothing but a chain
// of couplings of ONE kind whose source plane alternates (every RealNVP / NICE preset after the reversals have
// been folded into the weight order):
//
//   * the elementwise layers between the couplings are DEFERRED by the host compiler (fused.py): an element keeps
//     a pending affine map (s, t) that is folded into W1 / b1 where the element is a conditioner input and applied
//     as ONE fma where it is transformed ("pre-affine of the target plane"); whatever is still pending at the end
//     is one fma per element (TFK_OP_EW_FMA) and all the constant log-dets are one number;
//   * W1 / b1 arrive multiplied by 2 log2(e) and the scale-logit rows of W2 / b2 by log2(e) / 2 (+ c0 log2(e)), so
//     tanh is 1 - 2 / (exp2(.) + 1) and alpha = exp2(.) + 1e-10 with no multiplications in the kernel; the
//     log-det is accumulated in base 2 and scaled by ln 2 once per row;
//   * the number of GEMM-2 steps, the kind and the roles of the planes are compile-time constants: no dispatch,
//     no moves, no selects; A-operands are stored lane-major so that four k-steps arrive per ds_read_b128.
//
// Parameter block of a lean coupling op (floats), EPL source k-steps, T2 tiles of GEMM 2:
//   A1[EPL/4][64][4] | b1[4][4] | A2[nA2/4][64][4] | b2[T2][4][4] | pre_s[HALF] | pre_t[HALF]
//   nA2 = T2 * STEPS2 rounded up to a multiple of 4; lane l's value for (tile t, step k) is entry t * STEPS2 + k.
// TFK_OP_EW_FMA: s[D] | t[D] | logdet_const | pad[3]      z = fma(s, x, t)
#pragma once
#include "tfk_common.h"
#ifndef TFK_CHAIN_OVERSUB
#define TFK_CHAIN_OVERSUB kGridOversubscribe   // resident sets of workgroups a launch is cut into (tuning: tools/variants.sh)
#endif

namespace tfk {

constexpr int kMaxChainOps = 64;
constexpr int kChainSideOps = 3;

struct ChainProg {
    int n_c;          // couplings
    int first_src;    // source plane of the first coupling (they alternate)
    int ew_offset;    // TFK_OP_EW_FMA that ends the program (-1: none)
    int pad;
    int offset[kMaxChainOps];
    double *sum_ws;   // tfk_flow_run_mfma_sum: counter + one partial per workgroup (finish_sum_f64), or null
    double *sum_out;  // ... and the fp64 sum of the launch's log-probabilities
    const float *context;   // (N, ctx_n) rows of context, or null: the couplings' conditioners read [x_A | context]
    int ctx_n, ctx_steps;   // context elements; k-steps of 4 they take in GEMM 1 (A1c[64][4] ends each coupling's block)
    // (context programs) elementwise ops in front of the couplings and behind the closing TFK_OP_EW_FMA:
    // kind 0 none, 1 constant x -> s x + t (an EW_FMA block), 2 / 3 context-conditioned multiply-add / subtract-divide
    int pre_kind[kChainSideOps], pre_off[kChainSideOps];
    int post_kind[kChainSideOps], post_off[kChainSideOps];
    // ODD event sizes (HalfSplit: one target more than sources; with reversals the MIDDLE element is a target of every
    // coupling, i.e. it must sit in whichever plane is being transformed): both planes reserve their LAST slot for it, and
    // bit o says that coupling o first takes it over from the other plane (one select + one clear, lanes q == 3 only)
    unsigned long long move_mask;
};

typedef float cf32x4 __attribute__((ext_vector_type(4)));
typedef float cf32x2 __attribute__((ext_vector_type(2)));

// EPL = 2 (16-wide rows, event sizes <= 16): a lane's two contiguous row elements / parameters are ONE 8-byte access where
// the wider kernels take EPL / 4 16-byte ones.  The wider kernels keep their own code (register allocation at their
// VGPR caps is not to be disturbed); every place that differs is an `if constexpr (EPL >= 4) ... else ...`.
__device__ __forceinline__ void load2(const float *p, float &v0, float &v1)
{
    const cf32x2 w = *reinterpret_cast<const cf32x2 *>(p);
    v0 = w[0];
    v1 = w[1];
}

// log2 of the two scales of one GEMM-2 tile, alpha = exp2(u) + 1e-10 (affine.py:33-42):
//   log2 alpha = u + log2(1 + 1e-10 * 2^-u),
// and for u >= -8 (alpha >= 0.0039) the second term is below 3.7e-8 -- under the rounding of v_log_f32's own result
// there.  FAST: take u, and keep the smallest logit this lane has seen (one v_min3_f32 per tile); the kernel checks it
// ONCE per 16 rows after the chain and, if any lane of the wave saw a logit below -8, re-runs the chain on those rows
// with the logarithms (a guard per tile costs more than the 8 quarter-rate v_log_f32 per wave-layer it would save:
// 267 us per launch against 255 without the shortcut and 241 unguarded, RealNVP-64).
// -DTFK_LOG_SHORTCUT=0: always take the logarithm.
#ifndef TFK_LOG_SHORTCUT
#define TFK_LOG_SHORTCUT 1
#endif
constexpr float kLogShortcutMin = -8.0f;
template <bool FAST>
__device__ __forceinline__ float log2_scales(float u0, float u1, float al0, float al1, float &umin)
{
    if constexpr (FAST) {
        umin = fminf(fminf(umin, u0), u1);
        return u0 + u1;
    } else {
        return __builtin_amdgcn_logf(al0) + __builtin_amdgcn_logf(al1);
    }
}

// KIND: 0 affine fwd, 1 affine inv, 2 shift fwd, 3 shift inv
template <int EPL, int STEPS2, int KIND, bool FAST, bool CTX = false>
__device__ __forceinline__ void couple_lean(const float *prm, int lane, int q, const float (&src)[EPL],
                                            float (&tgt)[EPL], float &ld2, float &umin,
                                            const float (&cx)[4] = {0.0f, 0.0f, 0.0f, 0.0f}, int cs = 0)
{
    constexpr bool affine = KIND < 2;
    constexpr int HALF = 4 * EPL;
    constexpr int T2 = affine ? EPL / 2 : (EPL + 3) / 4;     // (EPL = 2, shift: one tile, rows r >= 2 of every group unused)
    constexpr int NA2 = (T2 * STEPS2 + 3) & ~3;
    const cf32x4 *A1 = reinterpret_cast<const cf32x4 *>(prm);
    const float *b1 = prm + EPL * 64;
    const cf32x4 *A2 = reinterpret_cast<const cf32x4 *>(b1 + 16);
    const float *b2 = b1 + 16 + NA2 * 64;
    const float *pre = b2 + T2 * 16;

    // GEMM 1 (weights pre-scaled by 2 log2 e): exp2(acc) = exp(2 * pre-activation)
    cf32x4 acc = *reinterpret_cast<const cf32x4 *>(b1 + 4 * q);
    if constexpr (EPL >= 4) {
#pragma unroll
    for (int g = 0; g < EPL / 4; ++g) {
        const cf32x4 w = A1[g * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], src[4 * g + k], acc, 0, 0, 0);
    }
    } else {                                                 // A1[64][2]
        float w0, w1;
        load2(prm + 2 * lane, w0, w1);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, src[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, src[1], acc, 0, 0, 0);
    }
    if constexpr (CTX) {                                     // [x_A | context] (conditioning/context.py:38-64)
        const cf32x4 w = *reinterpret_cast<const cf32x4 *>(pre + 2 * HALF + 4 * lane);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < cs) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], cx[k], acc, 0, 0, 0);
    }
    // the elements about to be transformed take their pending elementwise layers now (one fma)
    if constexpr (EPL >= 4) {
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const cf32x4 s = *reinterpret_cast<const cf32x4 *>(pre + EPL * q + 4 * i);
        const cf32x4 t = *reinterpret_cast<const cf32x4 *>(pre + HALF + EPL * q + 4 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) tgt[4 * i + k] = fmaf(s[k], tgt[4 * i + k], t[k]);
    }
    } else {
        float s0, s1, t0, t1;
        load2(pre + EPL * q, s0, s1);
        load2(pre + HALF + EPL * q, t0, t1);
        tgt[0] = fmaf(s0, tgt[0], t0);
        tgt[1] = fmaf(s1, tgt[1], t1);
    }
    float hid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)                              // tanh, transforms.py:293-304
        hid[r] = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[r]) + 1.0f), 1.0f);

    // GEMM 2 in groups of GT tiles whose A-operands fill whole ds_read_b128s (GT * STEPS2 is a multiple of 4): at most 12
    // operand registers are live, whatever the row width (all T2 * STEPS2 = 48 of them at D = 256 cost the occupancy)
    constexpr int GT = (STEPS2 == 4) ? 1 : ((STEPS2 == 2) ? 2 : 4);
    static_assert(T2 % GT == 0 || T2 < GT, "tile groups");
    constexpr int GTE = T2 < GT ? T2 : GT;                    // (shift couplings at D = 64: T2 = 2)
    constexpr int GREG = (GTE * STEPS2 + 3) & ~3;
#pragma unroll
    for (int t0 = 0; t0 < T2; t0 += GTE) {
        float a2[GREG];
#pragma unroll
        for (int g = 0; g < GREG / 4; ++g) {
            const cf32x4 w = A2[((t0 * STEPS2) / 4 + g) * 64 + lane];
            a2[4 * g] = w[0]; a2[4 * g + 1] = w[1]; a2[4 * g + 2] = w[2]; a2[4 * g + 3] = w[3];
        }
#pragma unroll
      for (int tt = 0; tt < GTE; ++tt) {
        const int t = t0 + tt;
        cf32x4 o = *reinterpret_cast<const cf32x4 *>(b2 + (t * 4 + q) * 4);
#pragma unroll
        for (int k = 0; k < STEPS2; ++k)
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tt * STEPS2 + k], hid[k], o, 0, 0, 0);
        if constexpr (affine) {
            // affine.py:33-34 with the logit row pre-scaled: o = (u / 2 + c0) log2 e
            if constexpr (FAST && KIND == 1) {               // 1 / alpha = 2^-u to 2.6e-8 relative while u >= -8 (log2_scales)
                ld2 += log2_scales<true>(o[0], o[2], 0.0f, 0.0f, umin);
#pragma unroll
                for (int i = 0; i < 2; ++i) tgt[2 * t + i] = (tgt[2 * t + i] - o[2 * i + 1]) * __builtin_amdgcn_exp2f(-o[2 * i]);
            } else {
            const float al[2] = {__builtin_amdgcn_exp2f(o[0]) + kAffMinScale, __builtin_amdgcn_exp2f(o[2]) + kAffMinScale};
            ld2 += log2_scales<FAST>(o[0], o[2], al[0], al[1], umin);               // log2 alpha; affine.py:42
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 2 * t + i;
                if constexpr (KIND == 0) tgt[e] = fmaf(al[i], tgt[e], o[2 * i + 1]);    // affine.py:48
                else tgt[e] = (tgt[e] - o[2 * i + 1]) * __builtin_amdgcn_rcpf(al[i]);   // affine.py:59
            }
            }
        } else {
#pragma unroll
            for (int i = 0; i < (EPL < 4 ? EPL : 4); ++i) {
                const int e = 4 * t + i;
                if constexpr (KIND == 2) tgt[e] = tgt[e] + o[i];                     // affine.py:150
                else tgt[e] = tgt[e] - o[i];                                         // affine.py:158
            }
        }
      }
    }
}


typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
typedef int li32x4 __attribute__((ext_vector_type(4)));

// The same coupling with GEMM 2 on the bf16 matrix pipe at fp32 accuracy ("bf16 x 3", see tfk_flow_rqs_chain.h): the
// weights arrive split into three bf16 pieces, the lane's 4 hidden activations are split here (22 vector
// instructions), and each tile takes three v_mfma_f32_16x16x32_bf16 (~16 cycles each) instead of STEPS2
// v_mfma_f32_16x16x4_f32 (32 cycles each); b2 rides as the weight of hidden unit 15 = 1 (hidden width <= 15).
// Block: A1[EPL/4][64][4] | b1[4][4] | A23[T2][2][64][4 dwords] | pre_s[HALF] | pre_t[HALF]
//   A23[t][0] = [W_hi | W_mid], A23[t][1] = [W_lo | W_hi]: 4 bf16 each = hidden units 4 i + (lane >> 4).
template <int EPL, int KIND, bool FAST>
__device__ __forceinline__ void couple_lean3(const float *prm, int lane, int q, const float (&src)[EPL],
                                             float (&tgt)[EPL], float &ld2, float &umin)
{
    constexpr bool affine = KIND < 2;
    constexpr int HALF = 4 * EPL;
    constexpr int T2 = affine ? EPL / 2 : EPL / 4;
    const cf32x4 *A1 = reinterpret_cast<const cf32x4 *>(prm);
    const float *b1 = prm + EPL * 64;
    const li32x4 *A23 = reinterpret_cast<const li32x4 *>(b1 + 16);
    const float *pre = b1 + 16 + T2 * 2 * 64 * 4;

    cf32x4 acc = *reinterpret_cast<const cf32x4 *>(b1 + 4 * q);
#pragma unroll
    for (int g = 0; g < EPL / 4; ++g) {
        const cf32x4 w = A1[g * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], src[4 * g + k], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const cf32x4 s = *reinterpret_cast<const cf32x4 *>(pre + EPL * q + 4 * i);
        const cf32x4 t = *reinterpret_cast<const cf32x4 *>(pre + HALF + EPL * q + 4 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) tgt[4 * i + k] = fmaf(s[k], tgt[4 * i + k], t[k]);
    }
    int hi[4], mid[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float h = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[r]) + 1.0f), 1.0f);
        if (r == 3) h = (q == 3) ? 1.0f : h;                  // hidden unit 15 = 1 carries b2
        const int hb = __float_as_int(h) & (int)0xffff0000;
        const float r1 = h - __int_as_float(hb);
        const int mb = __float_as_int(r1) & (int)0xffff0000;
        hi[r] = hb; mid[r] = mb; lo[r] = __float_as_int(r1 - __int_as_float(mb));
    }
    const int hh01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302), hh23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302);
    const int mm01 = __builtin_amdgcn_perm(mid[1], mid[0], 0x07060302), mm23 = __builtin_amdgcn_perm(mid[3], mid[2], 0x07060302);
    const int ll01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302), ll23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302);
    const li32x4 B1 = {hh01, hh23, hh01, hh23}, B2 = {mm01, mm23, mm01, mm23}, B3 = {hh01, hh23, ll01, ll23};

    // the tiles side by side, product by product (a tile's three MFMAs must not follow each other back to back), in groups
    // of four: 32 operand / accumulator registers whatever the row width (D = 256 has 16 tiles per coupling)
    constexpr int GT3 = T2 < 4 ? T2 : 4;
    static_assert(T2 % GT3 == 0, "tile groups");
#pragma unroll
  for (int g0 = 0; g0 < T2; g0 += GT3) {
    cf32x4 o[T2];
    li32x4 a1[GT3];
#pragma unroll
    for (int tt = 0; tt < GT3; ++tt) {
        o[g0 + tt] = cf32x4{0.0f, 0.0f, 0.0f, 0.0f};
        a1[tt] = A23[((g0 + tt) * 2) * 64 + lane];
    }
#pragma unroll
    for (int tt = 0; tt < GT3; ++tt)
        o[g0 + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8, a1[tt]), __builtin_bit_cast(lbf16x8, B1), o[g0 + tt], 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < GT3; ++tt)
        o[g0 + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8, a1[tt]), __builtin_bit_cast(lbf16x8, B2), o[g0 + tt], 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < GT3; ++tt) {
        const li32x4 a2 = A23[((g0 + tt) * 2 + 1) * 64 + lane];
        o[g0 + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8, a2), __builtin_bit_cast(lbf16x8, B3), o[g0 + tt], 0, 0, 0);
    }
#pragma unroll
    for (int t = g0; t < g0 + GT3; ++t) {
        if constexpr (affine) {
            if constexpr (FAST && KIND == 1) {
                ld2 += log2_scales<true>(o[t][0], o[t][2], 0.0f, 0.0f, umin);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    tgt[2 * t + i] = (tgt[2 * t + i] - o[t][2 * i + 1]) * __builtin_amdgcn_exp2f(-o[t][2 * i]);
            } else {
            const float al[2] = {__builtin_amdgcn_exp2f(o[t][0]) + kAffMinScale, __builtin_amdgcn_exp2f(o[t][2]) + kAffMinScale};
            ld2 += log2_scales<FAST>(o[t][0], o[t][2], al[0], al[1], umin);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 2 * t + i;
                if constexpr (KIND == 0) tgt[e] = fmaf(al[i], tgt[e], o[t][2 * i + 1]);
                else tgt[e] = (tgt[e] - o[t][2 * i + 1]) * __builtin_amdgcn_rcpf(al[i]);
            }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * t + i;
                if constexpr (KIND == 2) tgt[e] = tgt[e] + o[t][i];
                else tgt[e] = tgt[e] - o[t][i];
            }
        }
    }
  }
}


// MADE-based affine layer, parallel map (MaskedAutoregressiveBijection.forward, layers_base.py:201-206; MADE = two
// masked linear layers, transforms.py:184-267, masks folded into the packed weights), in the lean form: the
// conditioner reads BOTH planes as they are (pending elementwise layers folded into W1 / b1), every element of both
// planes then takes its pending layers (one fma) and is transformed with parameters that depend on the preceding
// elements only.  KIND 4: alpha x + beta, 5: (x - beta) / alpha.
// Block: A1[2 EPL / 4][64][4] (plane A's k-steps, then plane B's) | b1[4][4] | A2[nA2 / 4][64][4] | b2[EPL][4][4] |
//        pre_s[D] | pre_t[D];  tile t < EPL / 2: this lane's elements 2 t, 2 t + 1 of plane A, else of plane B.
template <int EPL, int STEPS2, int KIND, bool FAST>
__device__ __forceinline__ void made_lean(const float *prm, int lane, int q, float (&a)[EPL], float (&b)[EPL], float &ld2,
                                          float &umin)
{
    constexpr int D = 8 * EPL, HALF = 4 * EPL;
    constexpr int T2 = EPL;
    constexpr int NA2 = (T2 * STEPS2 + 3) & ~3;
    const cf32x4 *A1 = reinterpret_cast<const cf32x4 *>(prm);
    const float *b1 = prm + 2 * EPL * 64;
    const cf32x4 *A2 = reinterpret_cast<const cf32x4 *>(b1 + 16);
    const float *b2 = b1 + 16 + NA2 * 64;
    const float *pre = b2 + T2 * 16;

    cf32x4 acc = *reinterpret_cast<const cf32x4 *>(b1 + 4 * q);
#pragma unroll
    for (int g = 0; g < EPL / 4; ++g) {
        const cf32x4 wa = A1[g * 64 + lane], wb = A1[(EPL / 4 + g) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k], a[4 * g + k], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[k], b[4 * g + k], acc, 0, 0, 0);
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
            a[4 * i + k] = fmaf(sa[k], a[4 * i + k], ta[k]);
            b[4 * i + k] = fmaf(sb[k], b[4 * i + k], tb[k]);
        }
    }
    float hid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        hid[r] = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[r]) + 1.0f), 1.0f);

    constexpr int GT = (STEPS2 == 4) ? 1 : ((STEPS2 == 2) ? 2 : 4);
    constexpr int GREG = GT * STEPS2;
#pragma unroll
    for (int t0 = 0; t0 < T2; t0 += GT) {
        float a2[GREG];
#pragma unroll
        for (int g = 0; g < GREG / 4; ++g) {
            const cf32x4 w = A2[((t0 * STEPS2) / 4 + g) * 64 + lane];
            a2[4 * g] = w[0]; a2[4 * g + 1] = w[1]; a2[4 * g + 2] = w[2]; a2[4 * g + 3] = w[3];
        }
#pragma unroll
        for (int tt = 0; tt < GT; ++tt) {
            const int t = t0 + tt;
            cf32x4 o = *reinterpret_cast<const cf32x4 *>(b2 + (t * 4 + q) * 4);
#pragma unroll
            for (int k = 0; k < STEPS2; ++k)
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tt * STEPS2 + k], hid[k], o, 0, 0, 0);
            if constexpr (FAST && KIND == 5) {
                ld2 += log2_scales<true>(o[0], o[2], 0.0f, 0.0f, umin);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float &v = (t < EPL / 2) ? a[2 * t + i] : b[2 * (t - EPL / 2) + i];
                    v = (v - o[2 * i + 1]) * __builtin_amdgcn_exp2f(-o[2 * i]);
                }
            } else {
            const float al[2] = {__builtin_amdgcn_exp2f(o[0]) + kAffMinScale, __builtin_amdgcn_exp2f(o[2]) + kAffMinScale};
            ld2 += log2_scales<FAST>(o[0], o[2], al[0], al[1], umin);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float &v = (t < EPL / 2) ? a[2 * t + i] : b[2 * (t - EPL / 2) + i];
                if constexpr (KIND == 4) v = fmaf(al[i], v, o[2 * i + 1]);
                else v = (v - o[2 * i + 1]) * __builtin_amdgcn_rcpf(al[i]);
            }
            }
        }
    }
}

// one coupling in whichever operand format the kernel was instantiated for (STEPS2 = 0: bf16 x 3)
template <int EPL, int STEPS2, int KIND, bool FAST, bool CTX = false>
__device__ __forceinline__ void couple_fmt(const float *prm, int lane, int q, const float (&src)[EPL],
                                           float (&tgt)[EPL], float &ld2, float &umin,
                                           const float (&cx)[4] = {0.0f, 0.0f, 0.0f, 0.0f}, int cs = 0)
{
    if constexpr (STEPS2 == 0) couple_lean3<EPL, KIND, FAST>(prm, lane, q, src, tgt, ld2, umin);
    else couple_lean<EPL, STEPS2, KIND, FAST, CTX>(prm, lane, q, src, tgt, ld2, umin, cx, cs);
}

// the middle element of an odd event size changes planes: slot EPL - 1 of lane group 3 (= column HALF - 1 of the plane)
template <int EPL>
__device__ __forceinline__ void move_middle(int q, float (&to)[EPL], float (&from)[EPL])
{
    to[EPL - 1] = (q == 3) ? from[EPL - 1] : to[EPL - 1];
    from[EPL - 1] = (q == 3) ? 0.0f : from[EPL - 1];          // (a padding column again: it must read 0 from here on)
}

// the couplings (or MADE layers) of the program on the 16 rows a wave holds
template <int EPL, int STEPS2, int KIND, bool FAST, bool CTX = false, bool ODD = false>
__device__ __forceinline__ void chain_layers(const float *lds, const ChainProg &prog, int lane, int q, float (&a)[EPL],
                                             float (&b)[EPL], float &ld2, float &umin,
                                             const float (&cx)[4] = {0.0f, 0.0f, 0.0f, 0.0f})
{
    const int cs = CTX ? prog.ctx_steps : 0;
    int o = 0;
    if constexpr (KIND >= 4) {                                    // MADE layers: both planes in, both planes out
#pragma unroll 1
        for (; o < prog.n_c; ++o)
            made_lean<EPL, STEPS2 == 0 ? 1 : STEPS2, KIND, FAST>(lds + prog.offset[o], lane, q, a, b, ld2, umin);
    } else {
        // (ODD: its own instantiation -- as uniform run-time branches the moves cost the even-size programs 2 %: 248.1 / 246.1
        // against 242.8 / 241.3 us per RealNVP-64 step on the same box)
        const unsigned long long mv = ODD ? prog.move_mask : 0ull;
        if (prog.first_src == 1 && prog.n_c > 0) {
            if (mv & 1ull) move_middle<EPL>(q, a, b);
            couple_fmt<EPL, STEPS2, KIND, FAST, CTX>(lds + prog.offset[0], lane, q, b, a, ld2, umin, cx, cs);
            o = 1;
        }
        for (; o + 1 < prog.n_c; o += 2) {
            if ((mv >> o) & 1ull) move_middle<EPL>(q, b, a);
            couple_fmt<EPL, STEPS2, KIND, FAST, CTX>(lds + prog.offset[o], lane, q, a, b, ld2, umin, cx, cs);
            if ((mv >> (o + 1)) & 1ull) move_middle<EPL>(q, a, b);
            couple_fmt<EPL, STEPS2, KIND, FAST, CTX>(lds + prog.offset[o + 1], lane, q, b, a, ld2, umin, cx, cs);
        }
        if (o < prog.n_c) {
            if ((mv >> o) & 1ull) move_middle<EPL>(q, b, a);
            couple_fmt<EPL, STEPS2, KIND, FAST, CTX>(lds + prog.offset[o], lane, q, a, b, ld2, umin, cx, cs);
        }
    }
}

// x -> s x + t on this lane's elements of both planes (block: s[D] | t[D] | constant log-det, pad[3])
template <int EPL>
__device__ __forceinline__ void ew_fma_apply(const float *ew, int q, float (&a)[EPL], float (&b)[EPL], float &ld)
{
    constexpr int D = 8 * EPL, HALF = 4 * EPL;
    if constexpr (EPL >= 4) {
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const cf32x4 sa = *reinterpret_cast<const cf32x4 *>(ew + EPL * q + 4 * i);
        const cf32x4 sb = *reinterpret_cast<const cf32x4 *>(ew + HALF + EPL * q + 4 * i);
        const cf32x4 ta = *reinterpret_cast<const cf32x4 *>(ew + D + EPL * q + 4 * i);
        const cf32x4 tb = *reinterpret_cast<const cf32x4 *>(ew + D + HALF + EPL * q + 4 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[4 * i + k] = fmaf(sa[k], a[4 * i + k], ta[k]);
            b[4 * i + k] = fmaf(sb[k], b[4 * i + k], tb[k]);
        }
    }
    } else {
        float sa[2], sb[2], ta[2], tb[2];
        load2(ew + EPL * q, sa[0], sa[1]);
        load2(ew + HALF + EPL * q, sb[0], sb[1]);
        load2(ew + D + EPL * q, ta[0], ta[1]);
        load2(ew + D + HALF + EPL * q, tb[0], tb[1]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            a[k] = fmaf(sa[k], a[k], ta[k]);
            b[k] = fmaf(sb[k], b[k], tb[k]);
        }
    }
    if (q == 0) ld = ld + ew[2 * D];
}

// An elementwise affine layer whose parameters are a Linear map of the CONTEXT (ElementwiseBijection with a
// context_shape, layers_base.py:300-318) inside a lean context program: one GEMM context -> (D, 2), then the affine
// transform of every element of both planes -- the interpreter's TFK_OP_EWC_* op (tfk_flow_mfma.h: ewc_m), same block
// layout: Ac[EPL][cs][64] | bc[EPL][4][4]; tile t < EPL / 2: this lane's elements 2 t, 2 t + 1 of plane A, else of plane
// B -- but with the scale-logit rows pre-multiplied by log2(e) / 2 (bias += log(1 - 1e-10) log2 e) as in the lean couplings.
template <int EPL, bool DIVIDE>
__device__ __forceinline__ void ewc_lean(const float *prm, int cs, int lane, int q, float (&a)[EPL], float (&b)[EPL],
                                         float &ld, const float (&cx)[4])
{
    const float *Ac = prm;
    const float *bc = prm + EPL * cs * 64;
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        cf32x4 o = *reinterpret_cast<const cf32x4 *>(bc + (t * 4 + q) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < cs) o = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[(t * cs + k) * 64 + lane], cx[k], o, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // the logit rows arrive pre-scaled like a lean coupling's: o = (u / 2 + c0) log2 e (fused.py: lean=True)
            const float al = __builtin_amdgcn_exp2f(o[2 * i]) + kAffMinScale;
            const float be = o[2 * i + 1];
            part += __builtin_amdgcn_logf(al);                              // log2 alpha
            float &v = (t < EPL / 2) ? a[2 * t + i] : b[2 * (t - EPL / 2) + i];
            if (!DIVIDE) v = fmaf(al, v, be);                               // affine.py:48
            else v = (v - be) * __builtin_amdgcn_rcpf(al);                  // affine.py:59
        }
    }
    ld = fmaf(part, DIVIDE ? -__int_as_float(0x3f317218) : __int_as_float(0x3f317218), ld);        // ln 2
}

template <int EPL>
__device__ __forceinline__ void side_op(int kind, const float *prm, int cs, int lane, int q, float (&a)[EPL],
                                        float (&b)[EPL], float &ld, const float (&cx)[4])
{
    if (kind == 1) ew_fma_apply<EPL>(prm, q, a, b, ld);
    else if (kind == 2) ewc_lean<EPL, false>(prm, cs, lane, q, a, b, ld, cx);
    else if (kind == 3) ewc_lean<EPL, true>(prm, cs, lane, q, a, b, ld, cx);
}

// floats of one lean coupling's parameter block (fp32 operand format), see couple_lean
template <int EPL, int STEPS2, int KIND>
constexpr int chain_block_floats()
{
    constexpr int T2 = KIND < 2 ? EPL / 2 : (EPL + 3) / 4;
    if constexpr (STEPS2 == 0)                               // bf16 x 3 operands: A1 | b1 | A23[T2][2][64][4 dwords] | pre_s | pre_t
        return EPL * 64 + 16 + T2 * 2 * 64 * 4 + 8 * EPL;
    constexpr int NA2 = (T2 * STEPS2 + 3) & ~3;
    return EPL * 64 + 16 + NA2 * 64 + T2 * 16 + 8 * EPL;
}

// STREAMED operands: chains whose blocks do not fit the LDS together (RealNVP at D = 256: 22 KB per coupling) keep them
// in global memory; the workgroup holds TFK_STREAM_DEPTH + 1 blocks in the LDS and, while its waves compute coupling l
// out of one, the LDS-DMA (global_load_lds_dwordx4: no registers, 1 KB per wave-instruction) fills the others with the
// couplings that follow -- after the chain's last coupling, with the first ones for the next 16 rows of every wave.  One
// barrier per coupling: it publishes the block that has landed (each wave waits for its own share first, by COUNT, so
// the later blocks stay in flight) and frees the one just computed from.
// The whole chain is ONE launch: the rows cross HBM once (two half-chain launches wrote and re-read them).
// blocks in flight ahead of the one being computed from (TFK_STREAM_DEPTH + 1 buffers in the LDS)
#ifndef TFK_STREAM_DEPTH
#define TFK_STREAM_DEPTH 1
#endif
constexpr int kStreamBufs = TFK_STREAM_DEPTH + 1;

// this wave's share of one coupling's block: global -> LDS, no registers
template <int BLOCK, int LB>
__device__ __forceinline__ void stream_block(const float *__restrict__ src_f, float *dst_f, int lane, int wave)
{
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const char *src = reinterpret_cast<const char *>(src_f);
    // (measured, RealNVP-256, 2^19 rows: 621 us per launch; 574 without the DMA -- stale operands --, 543 without DMA and
    // barriers; two or three blocks in flight instead of one: 632 / 634; every workgroup starting at a different 1 KB
    // piece so the chip's requests spread over the L2 channels: 636; 256-thread workgroups, two per CU: 835)
    for (int c = wave; c * 1024 < LB * 4; c += BLOCK / 64) {
        const int off = c * 1024 + lane * 16;
        if (off < LB * 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lptr_t)(reinterpret_cast<char *>(dst_f) + c * 1024), 16, 0, 0);
    }
}

template <int EPL, int BLOCK, int STEPS2, int KIND, bool FAST>
__device__ __forceinline__ void chain_layers_stream(float *buf0, const float *__restrict__ params, const ChainProg &prog,
                                                    int &bufi, int lane, int q, float (&a)[EPL], float (&b)[EPL],
                                                    float &ld2, float &umin)
{
    constexpr int LB = chain_block_floats<EPL, STEPS2, KIND>();
    // every wave issues at least KMIN LDS-DMA instructions per block, in order: "at most KMIN (DEPTH - 1) outstanding"
    // means the block for THIS step has landed while the later ones may still be in flight
    constexpr int KMIN = ((LB * 4) / 1024) / (BLOCK / 64);
    constexpr int WAIT = KMIN * (TFK_STREAM_DEPTH - 1);
    const int wave = threadIdx.x >> 6;
    int ahead = TFK_STREAM_DEPTH % (prog.n_c > 0 ? prog.n_c : 1);     // layer of the block to fetch next
#pragma unroll 1
    for (int l = 0; l < prog.n_c; ++l) {
#ifndef TFK_STREAM_NOBARRIER
        // this wave's share of the block for this step has landed; the barrier publishes everyone's and frees the buffer
        // the previous step computed from (no memory access of the compiler's crosses the statement)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WAIT) : "memory");
#endif
        float *cur = buf0 + bufi * LB;
        int nb = bufi + TFK_STREAM_DEPTH;
        nb = nb >= kStreamBufs ? nb - kStreamBufs : nb;
#ifndef TFK_STREAM_NODMA                                              // (measurement only: compute on stale operands)
        stream_block<BLOCK, LB>(params + prog.offset[ahead], buf0 + nb * LB, lane, wave);
#endif
        ahead = ahead + 1 < prog.n_c ? ahead + 1 : 0;
        if (((prog.first_src + l) & 1) == 0) couple_fmt<EPL, STEPS2, KIND, FAST>(cur, lane, q, a, b, ld2, umin);
        else couple_fmt<EPL, STEPS2, KIND, FAST>(cur, lane, q, b, a, ld2, umin);
        bufi = bufi + 1 < kStreamBufs ? bufi + 1 : 0;
    }
}

// (tuning hook: -DTFK_CHAIN_ATTR='__attribute__((amdgpu_waves_per_eu(5, 5)))' for occupancy experiments, tools/variants.sh)
#ifndef TFK_CHAIN_ATTR
#define TFK_CHAIN_ATTR
#endif

template <int EPL, int BLOCK, int STEPS2, int KIND, bool STREAM = false, bool CTX = false, bool ODD = false>
__global__ __launch_bounds__(BLOCK) TFK_CHAIN_ATTR
__attribute__((amdgpu_waves_per_eu((EPL == 16 && BLOCK == 768) ? 3 : (((EPL == 16 && BLOCK == 1024) || (EPL == 8 && CTX)) ? 4 : 1)))) void k_flow_chain(
    const float *__restrict__ x, float *z, float *logdet, const float *__restrict__ gauss_loc,
    const float *__restrict__ gauss_log_scale, float *logprob, long long N,
    const float *__restrict__ params, int n_params, ChainProg prog, int flags, int xw)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = 8 * EPL, HALF = 4 * EPL;
    const int accumulate = flags & 1;
    const bool reverse_out = (flags & 2) != 0;
    const bool base_of_input = (flags & 4) != 0;
    // resident: the whole parameter block; streamed: two coupling blocks | the closing TFK_OP_EW_FMA block
    constexpr int LB = chain_block_floats<EPL, STEPS2, KIND>();
    static_assert(!STREAM || KIND < 4, "streamed operands: affine / shift couplings");
    static_assert(!CTX || (KIND < 4 && STEPS2 != 0 && !STREAM), "context: affine / shift couplings, fp32 format, resident");
    float *ew_s = STREAM ? lds + kStreamBufs * LB : lds + (prog.ew_offset >= 0 ? prog.ew_offset : 0);
    if constexpr (STREAM) {
        if (prog.ew_offset >= 0)
            for (int i = threadIdx.x; i < 2 * D + 4; i += BLOCK) ew_s[i] = params[prog.ew_offset + i];
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(params);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < (n_params >> 2); i += BLOCK) dst[i] = src[i];
    }
    // base density as  -0.5 sum ((z - loc) / scale)^2 - sum (log scale + 0.5 log 2 pi):  loc[D] | 1/scale[D] | const
    float *base_s = STREAM ? lds + kStreamBufs * LB + 2 * D + 4 : lds + n_params;
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

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    constexpr int rows_per_block = (BLOCK / 64) * 16;
    const long long stride = (long long)gridDim.x * rows_per_block;
    const float base_const = logprob ? base_s[2 * D] : 0.0f;
    // (tfk_flow_run_mfma_sum) fp64 sum of the log-probabilities this thread wrote: in the LDS, not in two registers that
    // would be live across the whole chain (130 VGPRs = 3 waves per SIMD instead of 124 = 4: 254 -> 293 us, measured)
    double *lp_slot = reinterpret_cast<double *>(base_s + 2 * D + 4) + threadIdx.x;
    if (prog.sum_ws) *lp_slot = 0.0;
    int bufi = 0;                                                     // (streamed) buffer of the next coupling
    if constexpr (STREAM) {                                           // the first TFK_STREAM_DEPTH blocks
#pragma unroll
        for (int k = 0; k < TFK_STREAM_DEPTH; ++k)
            stream_block<BLOCK, LB>(params + prog.offset[k % (prog.n_c > 0 ? prog.n_c : 1)], lds + k * LB, lane, wave);
    }
    // (streamed: the barriers inside the chain need every wave of the workgroup in every iteration -- the loop runs on
    // the workgroup's first row, and waves past the end compute row N - 1 again and store nothing)
    for (long long blk0 = (long long)blockIdx.x * rows_per_block; blk0 < (STREAM ? N : N - wave * 16); blk0 += stride) {
        const long long row0 = blk0 + wave * 16;
        const long long row = row0 + j;
        const long long rr = row < N ? row : N - 1;    // tail: compute a valid row, store nothing
        float a[EPL], b[EPL];
        auto load_rows = [&]() {
        if (xw == D) {
            if constexpr (EPL >= 4) {
            const float4 *pa = reinterpret_cast<const float4 *>(x + rr * D + EPL * q);
            const float4 *pb = reinterpret_cast<const float4 *>(x + rr * D + HALF + EPL * q);
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                const float4 va = pa[i], vb = pb[i];
                a[4 * i] = va.x; a[4 * i + 1] = va.y; a[4 * i + 2] = va.z; a[4 * i + 3] = va.w;
                b[4 * i] = vb.x; b[4 * i + 1] = vb.y; b[4 * i + 2] = vb.z; b[4 * i + 3] = vb.w;
            }
            } else {
                load2(x + rr * D + EPL * q, a[0], a[1]);
                load2(x + rr * D + HALF + EPL * q, b[0], b[1]);
            }
        } else {
            // rows narrower than the kernel's planes (event sizes that are not 32 / 64 / 128 / 256): the caller's rows
            // are read as they are -- first half into the head of plane A, second half into the head of plane B, zeros
            // behind them (the padding is an exact identity by construction of the weights, fused.py)
            // (an ODD width: hl sources, then the middle element -- which goes to plane B's last column --, then hl targets)
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
        float ld = (q == 0 && logdet && accumulate) ? logdet[rr] : 0.0f;
        float sq = 0.0f;                                              // sum of squared standardised elements
        auto base_terms = [&]() {                                     // gaussian.py:46-54
            if constexpr (EPL < 4) {
                float la[2], lb[2], ia[2], ib[2];
                load2(base_s + EPL * q, la[0], la[1]);
                load2(base_s + HALF + EPL * q, lb[0], lb[1]);
                load2(base_s + D + EPL * q, ia[0], ia[1]);
                load2(base_s + D + HALF + EPL * q, ib[0], ib[1]);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float ta = (a[k] - la[k]) * ia[k];
                    const float tb = (b[k] - lb[k]) * ib[k];
                    sq = fmaf(ta, ta, sq);
                    sq = fmaf(tb, tb, sq);
                }
            }
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
        if (logprob && base_of_input) base_terms();                   // Flow.sample (flows.py:699-707)

        float ld2 = 0.0f;                                             // this lane's share, in base 2
        float umin = 0.0f;                                            // smallest scale logit seen (log2_scales)
        constexpr bool kShortcut = TFK_LOG_SHORTCUT && (KIND < 2 || KIND >= 4);
        if constexpr (STREAM) {
            chain_layers_stream<EPL, BLOCK, STEPS2, KIND, kShortcut>(lds, params, prog, bufi, lane, q, a, b, ld2, umin);
            if constexpr (kShortcut) {
                if (__syncthreads_or(umin < kLogShortcutMin)) {       // (workgroup-wide: the re-run streams the blocks again)
                    load_rows();
                    ld2 = 0.0f;
                    chain_layers_stream<EPL, BLOCK, STEPS2, KIND, false>(lds, params, prog, bufi, lane, q, a, b, ld2, umin);
                }
            }
        } else if constexpr (CTX) {
            // conditional flows: this lane's context elements 4 s + q of row j; the elementwise ops in front of the couplings
            float cx[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                cx[s4] = (4 * s4 + q < prog.ctx_n) ? prog.context[rr * prog.ctx_n + 4 * s4 + q] : 0.0f;
            float ld_pre = 0.0f;
            auto pre_ops = [&]() {
                ld_pre = 0.0f;
#pragma unroll 1
                for (int i = 0; i < kChainSideOps; ++i)
                    if (prog.pre_kind[i]) side_op<EPL>(prog.pre_kind[i], lds + prog.pre_off[i], prog.ctx_steps, lane, q, a, b, ld_pre, cx);
            };
            pre_ops();
            chain_layers<EPL, STEPS2, KIND, kShortcut, true>(lds, prog, lane, q, a, b, ld2, umin, cx);
            if constexpr (kShortcut) {
                if (__builtin_amdgcn_ballot_w64(umin < kLogShortcutMin) != 0) {
                    load_rows();
                    ld2 = 0.0f;
                    pre_ops();
                    chain_layers<EPL, STEPS2, KIND, false, true>(lds, prog, lane, q, a, b, ld2, umin, cx);
                }
            }
            if constexpr (KIND == 0) ld = fmaf(ld2, __int_as_float(0x3f317218), ld);          // ln 2
            else if constexpr (KIND == 1) ld = fmaf(ld2, -__int_as_float(0x3f317218), ld);
            ld2 = 0.0f;
            if (prog.ew_offset >= 0) ew_fma_apply<EPL>(ew_s, q, a, b, ld);
            ld += ld_pre;
#pragma unroll 1
            for (int i = 0; i < kChainSideOps; ++i)                   // the elementwise layers behind the couplings
                if (prog.post_kind[i]) side_op<EPL>(prog.post_kind[i], lds + prog.post_off[i], prog.ctx_steps, lane, q, a, b, ld, cx);
        } else {
        chain_layers<EPL, STEPS2, KIND, kShortcut, false, ODD>(lds, prog, lane, q, a, b, ld2, umin);
        if constexpr (kShortcut) {
            if (__builtin_amdgcn_ballot_w64(umin < kLogShortcutMin) != 0) {   // scales near the 1e-10 floor: with logarithms
                load_rows();                                          // (x is still intact: z is stored below)
                ld2 = 0.0f;
                chain_layers<EPL, STEPS2, KIND, false, false, ODD>(lds, prog, lane, q, a, b, ld2, umin);
            }
        }
        }
        if constexpr (KIND == 0 || KIND == 4) ld = fmaf(ld2, __int_as_float(0x3f317218), ld);          // ln 2
        else if constexpr (KIND == 1 || KIND == 5) ld = fmaf(ld2, -__int_as_float(0x3f317218), ld);

        if (!CTX && prog.ew_offset >= 0) {                            // what is still pending, one fma per element
            const float *ew = ew_s;
            if constexpr (EPL < 4) {
                float ld_const = 0.0f;                                // (the constant log-det is added below)
                ew_fma_apply<EPL>(ew, q, a, b, ld_const);
            }
#pragma unroll
            for (int i = 0; i < EPL / 4; ++i) {
                const cf32x4 sa = *reinterpret_cast<const cf32x4 *>(ew + EPL * q + 4 * i);
                const cf32x4 sb = *reinterpret_cast<const cf32x4 *>(ew + HALF + EPL * q + 4 * i);
                const cf32x4 ta = *reinterpret_cast<const cf32x4 *>(ew + D + EPL * q + 4 * i);
                const cf32x4 tb = *reinterpret_cast<const cf32x4 *>(ew + D + HALF + EPL * q + 4 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[4 * i + k] = fmaf(sa[k], a[4 * i + k], ta[k]);
                    b[4 * i + k] = fmaf(sb[k], b[4 * i + k], tb[k]);
                }
            }
            if (q == 0) ld = ld + ew[2 * D];
        }
        if (logprob && !base_of_input) base_terms();

        ld += __shfl_xor(ld, 16, kWave);
        ld += __shfl_xor(ld, 32, kWave);
        if (logprob) {
            sq += __shfl_xor(sq, 16, kWave);
            sq += __shfl_xor(sq, 32, kWave);
        }
        if (row < N) {
            if constexpr (EPL < 4) {
                if (z) {                                              // (reversed: element e of plane A lands in column D - 1 - e)
                    float *pa = reverse_out ? z + row * D + D - EPL * (q + 1) : z + row * D + EPL * q;
                    float *pb = reverse_out ? z + row * D + HALF - EPL * (q + 1) : z + row * D + HALF + EPL * q;
                    cf32x2 va, vb;
                    va[0] = reverse_out ? a[1] : a[0]; va[1] = reverse_out ? a[0] : a[1];
                    vb[0] = reverse_out ? b[1] : b[0]; vb[1] = reverse_out ? b[0] : b[1];
                    *reinterpret_cast<cf32x2 *>(pa) = va;
                    *reinterpret_cast<cf32x2 *>(pb) = vb;
                }
            } else
            if (z && !reverse_out) {
                float4 *qa = reinterpret_cast<float4 *>(z + row * D + EPL * q);
                float4 *qb = reinterpret_cast<float4 *>(z + row * D + HALF + EPL * q);
#pragma unroll
                for (int i = 0; i < EPL / 4; ++i) {
                    qa[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
                    qb[i] = make_float4(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]);
                }
            } else if (z) {                                           // a reversal after the program, folded into the store
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
                    const float lp = (fmaf(-0.5f, sq, -base_const)) + ld;                 // flows.py:648
                    logprob[row] = lp;
                    if (prog.sum_ws) *lp_slot += (double)lp;
                }
            }
        }
    }
    if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the block prefetched for a step that never came)
    if (prog.sum_ws) {
        const double mine = (q == 0) ? *lp_slot : 0.0;
        finish_sum_f64<BLOCK>(mine, reinterpret_cast<double *>(lds), prog.sum_ws, prog.sum_out);
    }
}

template <int EPL, int BLOCK, int STEPS2, int KIND, bool STREAM = false, bool CTX = false, bool ODD = false>
static int launch_chain_b(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                          float *logprob, int64_t N, const float *params, int n_params, const ChainProg &prog,
                          int flags, int xw, hipStream_t s, const char *fn)
{
    constexpr int D = 8 * EPL;
    const size_t lds = (STREAM ? ((size_t)kStreamBufs * chain_block_floats<EPL, STEPS2, KIND>() + 2 * (2 * D + 4)) * sizeof(float)
                               : ((size_t)n_params + 2 * D + 4) * sizeof(float))
                       + (prog.sum_ws ? (size_t)BLOCK * sizeof(double) : 0);       // one fp64 slot per thread
    if (lds > 160 * 1024)
        return fail(TFK_EINVAL, "%s: %zu bytes of parameters do not fit the 160 KiB LDS; split the program", fn, lds);
    auto kern = &k_flow_chain<EPL, BLOCK, STEPS2, KIND, STREAM, CTX, ODD>;
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
                       params, n_params, prog, flags, xw);
    return check_launch(fn);
}

template <int EPL, int KIND>
static int launch_chain_k(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                          float *logprob, int64_t N, const float *params, int n_params, const ChainProg &prog,
                          int steps2, int flags, int xw, hipStream_t s, const char *fn)
{
    // operands that do not fit the LDS beside each other are streamed (chain_layers_stream): D >= 128, fp32 format
    if ((size_t)n_params + 16 * EPL + 4 > 160 * 1024 / sizeof(float) || ((flags & 8) && EPL >= 16 && KIND < 4 && (steps2 != 0 || EPL == 32))) {
        if (prog.move_mask)
            return fail(TFK_EINVAL, "%s: programs of odd event sizes keep their operands resident (no streaming)", fn);
        if constexpr (EPL >= 16 && KIND < 4) {
#ifndef TFK_STREAM_BLOCK
#define TFK_STREAM_BLOCK 512
#endif
#define TFK_CS(ST_) \
    launch_chain_b<EPL, TFK_STREAM_BLOCK, ST_, KIND, true>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, flags, xw, s, fn)
            switch (steps2) {
            case 0:                 // bf16 x 3 operands (round 4): 42 KB per coupling at D = 256, two blocks resident
                if constexpr (EPL == 32) return TFK_CS(0);
                break;
            case 1: return TFK_CS(1);
            case 2: return TFK_CS(2);
            case 3: return TFK_CS(3);
            case 4: return TFK_CS(4);
            default: break;
            }
#undef TFK_CS
        }
        return fail(TFK_EINVAL, "%s: %d floats of parameters do not fit the 160 KiB LDS, and streamed operands exist for "
                    "affine / shift couplings at D >= 128 with fp32 operands; split the program", fn, n_params);
    }
    if (prog.context) {                                      // conditional flows: one workgroup size, fp32 operands
        // (D = 64 / 128: at D = 256 the context variant spills -- 660 B of scratch at the 256-VGPR cap, half the
        // interpreter's rate -- and is not built; the packer leaves conditional affine chains of that size to the interpreter)
        if constexpr (EPL >= 8 && EPL <= 16 && KIND < 4) {
            constexpr int BC = (EPL == 16) ? 768 : 512;
#define TFK_CC(ST_) \
    launch_chain_b<EPL, BC, ST_, KIND, false, true>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, flags, xw, s, fn)
            switch (steps2) {
            case 1: return TFK_CC(1);
            case 2: return TFK_CC(2);
            case 3: return TFK_CC(3);
            case 4: return TFK_CC(4);
            default: break;
            }
#undef TFK_CC
        }
        return fail(TFK_EINVAL, "%s: context-conditioned lean chains: affine / shift couplings, fp32 operands, D = 64 or 128", fn);
    }
    const bool big = N >= (int64_t)cu_count() * 3 * 128;
    // (D = 256: 768-thread workgroups capped at 168 VGPRs -- 3 waves per SIMD -- spill inside the coupling loop here:
    // 736 us per launch against 389 with 512 threads, measured; the interpreter's trick does not carry over)
    // (D = 128: the chain's operands, ~90 KB, allow one workgroup per CU: 768 threads at <= 168 VGPRs put 3 waves on every
    // SIMD -- RealNVP(128) 1.51e9 -> 1.54e9, RealNVP(100) 1.43e9 -> 1.52e9 evals/s; 1024 threads at 128 VGPRs: 1.38e9)
#ifndef TFK_CHAIN_BIG16
#define TFK_CHAIN_BIG16 768
#endif
    constexpr int BIG = (EPL == 16) ? TFK_CHAIN_BIG16 : 512;
    if (prog.move_mask) {                                    // odd event sizes: the instantiation with the middle-element moves
        if constexpr (KIND < 4) {
#define TFK_CBO(BLOCK_, ST_) \
    launch_chain_b<EPL, BLOCK_, ST_, KIND, false, false, true>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, flags, xw, s, fn)
            switch (steps2) {
            case 1: return big ? TFK_CBO(BIG, 1) : TFK_CBO(kBlock, 1);
            case 2: return big ? TFK_CBO(BIG, 2) : TFK_CBO(kBlock, 2);
            case 3: return big ? TFK_CBO(BIG, 3) : TFK_CBO(kBlock, 3);
            case 4: return big ? TFK_CBO(BIG, 4) : TFK_CBO(kBlock, 4);
            default: break;
            }
#undef TFK_CBO
        }
        return fail(TFK_EINVAL, "%s: programs of odd event sizes: affine / shift couplings with fp32 operands, 1..4 GEMM-2 steps", fn);
    }
#define TFK_CB(BLOCK_, ST_) \
    launch_chain_b<EPL, BLOCK_, ST_, KIND>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, flags, xw, s, fn)
    switch (steps2) {
    case 0:                         // bf16 x 3 operands: ~85 KB for RealNVP-64, one 1024-thread workgroup per CU
        if constexpr (EPL == 8 && KIND < 4) return big ? TFK_CB(1024, 0) : TFK_CB(kBlock, 0);
        else return fail(TFK_EINVAL, "%s: the bf16 x 3 operand format of lean couplings is built for D = 64", fn);
    case 1: return big ? TFK_CB(BIG, 1) : TFK_CB(kBlock, 1);
    case 2: return big ? TFK_CB(BIG, 2) : TFK_CB(kBlock, 2);
    case 3: return big ? TFK_CB(BIG, 3) : TFK_CB(kBlock, 3);
    case 4: return big ? TFK_CB(BIG, 4) : TFK_CB(kBlock, 4);
    default: return fail(TFK_EINVAL, "%s: lean couplings need 1..4 GEMM-2 steps, got %d", fn, steps2);
    }
#undef TFK_CB
}

template <int EPL>
static int launch_chain(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                        float *logprob, int64_t N, const float *params, int n_params, const ChainProg &prog,
                        int kind, int steps2, int flags, int xw, hipStream_t s, const char *fn)
{
    if constexpr (EPL < 4) {            // 16-wide rows: affine / shift couplings only (fp32 operands, resident, no context)
        if (kind >= 4 || steps2 == 0 || prog.context)
            return fail(TFK_EINVAL, "%s: 16-wide lean programs hold affine / shift couplings with fp32 operands and no context", fn);
        switch (kind) {
        case 0: return launch_chain_k<EPL, 0>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
        case 1: return launch_chain_k<EPL, 1>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
        case 2: return launch_chain_k<EPL, 2>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
        default: return launch_chain_k<EPL, 3>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
        }
    } else
    switch (kind) {
    case 0: return launch_chain_k<EPL, 0>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
    case 1: return launch_chain_k<EPL, 1>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
    case 2: return launch_chain_k<EPL, 2>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
    case 3: return launch_chain_k<EPL, 3>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
    case 4: return launch_chain_k<EPL, 4>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
    default: return launch_chain_k<EPL, 5>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, steps2, flags, xw, s, fn);
    }
}

}  // namespace tfk
