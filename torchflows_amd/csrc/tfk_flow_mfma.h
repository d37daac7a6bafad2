// tfk_flow_mfma.h -- fused flow programs with the conditioner GEMMs on the matrix cores
// (templates; instantiated per row width in tfk_flow_mfma_{8,16,32}.hip, entry point in tfk_flow_mfma.hip).
//
// Same contract as tfk_flow.hip (rows in registers, a list of ops applied to them, one launch
// for a whole chain of ElementwiseAffine / ActNorm / folded reversals / affine or shift
// couplings with their FeedForward(tanh) conditioner), but the two conditioner GEMMs run as
// v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: bit-for-bit an fmaf chain, so the
// numerics are those of the VALU version).
//
// The register layout is chosen so that NOTHING has to move between the steps of a layer
// (D = 8*EPL, EPL = elements per lane per plane = 8, 16 or 32):
//   * a wavefront owns 16 rows; lane l = (q = l >> 4, j = l & 15) holds, of row j, the
//     elements [EPL*q, EPL*(q+1)) of plane A (first half of the row) and of plane B
//     -- 32..64 contiguous bytes per plane: coalesced 16-byte loads;
//   * that is exactly the B-operand layout of the MFMA (lane supplies B[k = l>>4][j = l&15]):
//     step s of GEMM 1 feeds element EPL*q + s of the source plane as k = 4 s + q, with the
//     columns of W1 permuted to match when the weights are packed;
//   * the D-output layout (lane (q, j), register r <-> row 4q + r, column j) leaves lane
//     (q, j) with 4 hidden pre-activations of ITS row; packing W1's rows as "D-row 4q+r <->
//     hidden unit 4r+q" makes register r the B-operand of GEMM 2's step r without a move;
//   * GEMM 2 is tiled so that D-row 4q+r of tile t is parameter (r & 1) of target element
//     EPL*q + 2t + (r >> 1) (affine) / EPL*q + 4t + r (shift): every lane receives the
//     parameters of exactly the target elements it holds.
// Each hidden activation is computed once (4 tanh per lane instead of one per lane per unit),
// there is no cross-lane reduction inside a layer (per-lane log-det partial sums are combined
// once at the end with two ds_bpermute steps), and the matrix pipe runs beside the vector
// pipe that evaluates exp / log of the transform.
#pragma once
#include <cstring>

#include "tfk_common.h"
#include "tfk_spline.h"

namespace tfk {

constexpr int kMaxOpsM = 96;
constexpr int kMaxEplRqs = 16;   // RQS couplings on this kernel: D <= 128
constexpr int kCtxSteps = 4;     // context-conditioned programs: context size <= 16 (4 k-steps of 4)

struct MOp {
    int kind;        // TFK_OP_*
    int src_plane;   // coupling: which plane feeds the conditioner (bit 0); bits 4..7: k-steps of CONTEXT in GEMM 1
    int steps2;      // coupling: k-steps of GEMM 2 = ceil(H / 4)
    int offset;      // first float of the op's parameters in the staged block
    int K;           // RQS: number of bins (8)
    float boundary;  // RQS: spline box half-width
    float scale;     // RQS: 1 - min_bin_size * K
    float c;         // RQS: boundary_u_delta
};

struct MProgram {
    int n_ops;
    int pad[3];
    MOp op[kMaxOpsM];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tanh_act_m(float x) {       // see tanh_act in tfk_flow.hip
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// One coupling op with the kind and the roles of the planes fixed at compile time (src feeds
// the conditioner, tgt is transformed): straight-line code, the scheduler is free to slide the
// vector work of one tile under the MFMAs of the next.  apply_op_m dispatches once per op.
// HT = 16-unit tiles of the hidden layer (hidden width <= 16 HT): GEMM 1 keeps HT accumulators,
// GEMM 2 runs up to 4 HT k-steps (the unused ones are skipped uniformly).
template <int EPL, int KIND, int HT, bool CTX = false>
__device__ __forceinline__ void couple_m(const MOp op, const float *prm, int lane, int q,
                                         const float (&src)[EPL], float (&tgt)[EPL], float &ld,
                                         const float (&cx)[kCtxSteps] = {0.0f, 0.0f, 0.0f, 0.0f})
{
    constexpr bool affine = (KIND == TFK_OP_AFFINE_FWD || KIND == TFK_OP_AFFINE_INV);
    constexpr int T2 = affine ? EPL / 2 : EPL / 4;
    const float *A1 = prm;
    const float *b1 = prm + EPL * HT * 64;
    const float *A2 = b1 + HT * 16;
    const float *b2 = A2 + T2 * op.steps2 * 64;

    // GEMM 1: hidden pre-activations of the 16 rows of this wave
    f32x4 acc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) acc[t] = *reinterpret_cast<const f32x4 *>(b1 + t * 16 + 4 * q);   // units 16t + 4r + q
#pragma unroll
    for (int s = 0; s < EPL; ++s)
#pragma unroll
        for (int t = 0; t < HT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[(s * HT + t) * 64 + lane], src[s], acc[t], 0, 0, 0);
    if constexpr (CTX) {
        // the context enters the conditioner by concatenation behind x_A (conditioning/context.py:46-60): its columns
        // are further k-steps of GEMM 1, lane (q, j) supplying context element 4 s + q of row j
        const int cs = op.src_plane >> 4;
        const float *A1c = b2 + T2 * 16;
#pragma unroll
        for (int s = 0; s < kCtxSteps; ++s)
            if (s < cs) {
#pragma unroll
                for (int t = 0; t < HT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1c[(s * HT + t) * 64 + lane], cx[s], acc[t], 0, 0, 0);
            }
    }
    float hid[4 * HT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[4 * t + r] = tanh_act_m(acc[t][r]);    // transforms.py:293-304

    // GEMM 2 + transform, two (affine) or four (shift) target elements per tile
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < T2; ++t) {
        f32x4 o = *reinterpret_cast<const f32x4 *>(b2 + (t * 4 + q) * 4);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2) * 64 + lane], hid[0], o, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 4 * HT; ++k)
            if (op.steps2 > k)
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + k) * 64 + lane], hid[k], o, 0, 0, 0);
        if constexpr (affine) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 2 * t + i;
                const float al = aff_alpha_lean(o[2 * i]);          // affine.py:33-34
                const float be = o[2 * i + 1];
                part += log_lean(al);                               // affine.py:42
                if constexpr (KIND == TFK_OP_AFFINE_FWD) tgt[e] = al * tgt[e] + be;   // affine.py:48
                else tgt[e] = (tgt[e] - be) * __builtin_amdgcn_rcpf(al);              // affine.py:59 (v_rcp: 1 ulp)
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * t + i;
                if constexpr (KIND == TFK_OP_SHIFT_FWD) tgt[e] = tgt[e] + o[i];       // affine.py:150
                else tgt[e] = tgt[e] - o[i];                                          // affine.py:158
            }
        }
    }
    if constexpr (KIND == TFK_OP_AFFINE_FWD) ld = ld + part;
    else if constexpr (KIND == TFK_OP_AFFINE_INV) ld = ld + (-part);
}

// MADE-based affine layer, parallel map (MaskedAutoregressiveBijection.forward, layers_base.py:201-206;
// MADE = two masked linear layers, transforms.py:184-267, masks folded into the packed weights): the
// conditioner reads BOTH planes, every element of both planes is transformed with parameters that
// depend on the preceding elements only -- all of them computed from the untouched row.
template <int EPL, bool DIVIDE, int HT>
__device__ __forceinline__ void made_m(const MOp op, const float *prm, int lane, int q,
                                       float (&a)[EPL], float (&b)[EPL], float &ld)
{
    constexpr int T2 = EPL / 2;
    const float *A1 = prm;
    const float *b1 = prm + 2 * EPL * HT * 64;
    const float *A2 = b1 + HT * 16;
    const float *b2 = A2 + 2 * T2 * op.steps2 * 64;
    f32x4 acc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) acc[t] = *reinterpret_cast<const f32x4 *>(b1 + t * 16 + 4 * q);
#pragma unroll
    for (int s = 0; s < EPL; ++s)
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[(s * HT + t) * 64 + lane], a[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[((EPL + s) * HT + t) * 64 + lane], b[s], acc[t], 0, 0, 0);
        }
    float hid[4 * HT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[4 * t + r] = tanh_act_m(acc[t][r]);
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < 2 * T2; ++t) {
        f32x4 o = *reinterpret_cast<const f32x4 *>(b2 + (t * 4 + q) * 4);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2) * 64 + lane], hid[0], o, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 4 * HT; ++k)
            if (op.steps2 > k)
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + k) * 64 + lane], hid[k], o, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float al = aff_alpha_lean(o[2 * i]);
            const float be = o[2 * i + 1];
            part += log_lean(al);
            float &v = (t < T2) ? a[2 * t + i] : b[2 * (t - T2) + i];
            if (!DIVIDE) v = al * v + be;
            else v = (v - be) * __builtin_amdgcn_rcpf(al);
        }
    }
    ld = ld + (DIVIDE ? -part : part);
}

// MADE-based RQ-spline layer, parallel map (MaskedAutoregressiveRQNSF.forward / InverseAutoregressiveRQNSF
// .inverse): GEMM 1 over BOTH planes as in made_m, then for each of the lane's 2 EPL elements its 23 spline
// parameters as 6 tiles (made of the untouched row's hidden activations) and the spline's forward map.
// Parameter block: A1[2 EPL][64] | b1[4][4] | A2[2 EPL * 6][steps2][64] | b2[2 EPL * 6][4][4]; tiles
// [0, 6 EPL) belong to plane A, the rest to plane B.
template <int EPL>
__device__ __forceinline__ void made_rqs_m(const MOp op, const float *prm, int lane, int q,
                                           float (&a)[EPL], float (&b)[EPL], float &ld)
{
    constexpr int T2 = EPL * 6;
    const float *A1 = prm;
    const float *b1 = prm + 2 * EPL * 64;
    const float *A2 = b1 + 16;
    const float *b2 = A2 + 2 * T2 * op.steps2 * 64;
    RqsConst C;
    C.minimum = -op.boundary;
    C.maximum = op.boundary;
    C.span = op.boundary + op.boundary;
    C.scale = op.scale;
    C.c = op.c;
    f32x4 acc = *reinterpret_cast<const f32x4 *>(b1 + 4 * q);
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[s * 64 + lane], a[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[(EPL + s) * 64 + lane], b[s], acc, 0, 0, 0);
    }
    float hid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[r] = tanh_act_m(acc[r]);
    float part = 0.0f;
    for (int e = 0; e < 2 * EPL; ++e) {
        float p[24];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int t = e * 6 + c;
            f32x4 o = *reinterpret_cast<const f32x4 *>(b2 + (t * 4 + q) * 4);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2) * 64 + lane], hid[0], o, 0, 0, 0);
            if (op.steps2 > 1) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + 1) * 64 + lane], hid[1], o, 0, 0, 0);
            if (op.steps2 > 2) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + 2) * 64 + lane], hid[2], o, 0, 0, 0);
            if (op.steps2 > 3) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + 3) * 64 + lane], hid[3], o, 0, 0, 0);
            p[4 * c] = o[0]; p[4 * c + 1] = o[1]; p[4 * c + 2] = o[2]; p[4 * c + 3] = o[3];
        }
        float v = a[0];
#pragma unroll
        for (int i = 1; i < EPL; ++i) v = (e == i) ? a[i] : v;
#pragma unroll
        for (int i = 0; i < EPL; ++i) v = (e == EPL + i) ? b[i] : v;
        float out = v, l = 0.0f;
        if (v > C.minimum && v < C.maximum) rqs_eval<8, false, true, float[24]>(p, 8, v, C, out, l);
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            a[i] = (e == i) ? out : a[i];
            b[i] = (e == EPL + i) ? out : b[i];
        }
        part += l;
    }
    ld = ld + part;
}

// RQ-spline coupling on the matrix cores (layers.py:154-163): GEMM 1 as above; GEMM 2 produces,
// for each of this lane's EPL target elements, its 23 (+1 pad) spline parameters as 6 tiles of
// 4 (D-row 4q+r of tile 6e+c <-> parameter 4c+r of target element EPL*q+e), i.e. the record
// lands in 24 registers of the lane that owns the element; rqs_eval then runs out of registers.
// The element loop is a run-time loop (one copy of the ~500-op spline per variant).
template <int EPL, bool INVERSE, bool CTX = false>
__device__ __forceinline__ void couple_rqs_m(const MOp op, const float *prm, int lane, int q,
                                             const float (&src)[EPL], float (&tgt)[EPL], float &ld,
                                             const float (&cx)[kCtxSteps] = {0.0f, 0.0f, 0.0f, 0.0f})
{
    constexpr int T2 = EPL * 6;
    const float *A1 = prm;
    const float *b1 = prm + EPL * 64;
    const float *A2 = b1 + 16;
    const float *b2 = A2 + T2 * op.steps2 * 64;
    RqsConst C;
    C.minimum = -op.boundary;
    C.maximum = op.boundary;
    C.span = op.boundary + op.boundary;
    C.scale = op.scale;
    C.c = op.c;

    f32x4 acc = *reinterpret_cast<const f32x4 *>(b1 + 4 * q);
#pragma unroll
    for (int s = 0; s < EPL; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[s * 64 + lane], src[s], acc, 0, 0, 0);
    if constexpr (CTX) {
        const int cs = op.src_plane >> 4;
        const float *A1c = b2 + T2 * 16;
#pragma unroll
        for (int s = 0; s < kCtxSteps; ++s)
            if (s < cs) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A1c[s * 64 + lane], cx[s], acc, 0, 0, 0);
    }
    float hid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[r] = tanh_act_m(acc[r]);

    float part = 0.0f;
    for (int e = 0; e < EPL; ++e) {
        float p[24];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int t = e * 6 + c;
            f32x4 o = *reinterpret_cast<const f32x4 *>(b2 + (t * 4 + q) * 4);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2) * 64 + lane], hid[0], o, 0, 0, 0);
            if (op.steps2 > 1) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + 1) * 64 + lane], hid[1], o, 0, 0, 0);
            if (op.steps2 > 2) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + 2) * 64 + lane], hid[2], o, 0, 0, 0);
            if (op.steps2 > 3) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[(t * op.steps2 + 3) * 64 + lane], hid[3], o, 0, 0, 0);
            p[4 * c] = o[0]; p[4 * c + 1] = o[1]; p[4 * c + 2] = o[2]; p[4 * c + 3] = o[3];
        }
        float v = tgt[0];
#pragma unroll
        for (int i = 1; i < EPL; ++i) v = (e == i) ? tgt[i] : v;
        float out = v, l = 0.0f;                                    // spline/base.py:54-55
        if (v > C.minimum && v < C.maximum)                         // strict box, base.py:29-33
            rqs_eval<8, INVERSE, true, float[24]>(p, 8, v, C, out, l);
#pragma unroll
        for (int i = 0; i < EPL; ++i) tgt[i] = (e == i) ? out : tgt[i];
        part += l;
    }
    ld = ld + part;                                                 // base.py:59 + :222
}

// Elementwise affine layer whose parameters are predicted from the CONTEXT by a Linear conditioner
// (ElementwiseBijection with a context_shape, layers_base.py:300-318): one GEMM context -> (D, 2), then the affine
// transform of every element of both planes.  Block: Ac[EPL][cs][64] | bc[EPL][4][4]; tile t < EPL / 2 holds the
// parameters of this lane's elements 2 t, 2 t + 1 of plane A, the other tiles those of plane B.
template <int EPL, bool DIVIDE>
__device__ __forceinline__ void ewc_m(const MOp op, const float *prm, int lane, int q, float (&a)[EPL], float (&b)[EPL],
                                      float &ld, const float (&cx)[kCtxSteps])
{
    const int cs = op.src_plane >> 4;
    const float *Ac = prm;
    const float *bc = prm + EPL * cs * 64;
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        f32x4 o = *reinterpret_cast<const f32x4 *>(bc + (t * 4 + q) * 4);
#pragma unroll
        for (int s = 0; s < kCtxSteps; ++s)
            if (s < cs) o = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[(t * cs + s) * 64 + lane], cx[s], o, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float al = aff_alpha_lean(o[2 * i]);
            const float be = o[2 * i + 1];
            part += log_lean(al);
            float &v = (t < EPL / 2) ? a[2 * t + i] : b[2 * (t - EPL / 2) + i];
            if (!DIVIDE) v = al * v + be;                                   // affine.py:48
            else v = (v - be) * __builtin_amdgcn_rcpf(al);                  // affine.py:59
        }
    }
    ld = ld + (DIVIDE ? -part : part);
}

// Parameter block of a coupling op (floats), EPL source steps, T2 tiles of GEMM 2:
//   A1[EPL][HT][64] | b1[HT][4][4] | A2[T2][steps2][64] | b2[T2][4][4],  HT = ceil(steps2 / 4) rounded
//   up to 1, 2, 4 or 8 (hidden width <= 128)
// Elementwise ops use the layout of tfk_flow.hip: alpha[D] | beta[D] | ldc, pad[3] | 1/alpha[D].
template <int EPL, int KIND, int HTMAX, bool CTX = false>
__device__ __forceinline__ void couple_any(const MOp op, const float *prm, int lane, int q,
                                           const float (&src)[EPL], float (&tgt)[EPL], float &ld,
                                           const float (&cx)[kCtxSteps] = {0.0f, 0.0f, 0.0f, 0.0f})
{
    if constexpr (HTMAX == 1) {
        couple_m<EPL, KIND, 1, CTX>(op, prm, lane, q, src, tgt, ld, cx);
    } else {
        if (op.steps2 <= 4) couple_m<EPL, KIND, 1>(op, prm, lane, q, src, tgt, ld);
        else if (op.steps2 <= 8) couple_m<EPL, KIND, 2>(op, prm, lane, q, src, tgt, ld);
        else if (HTMAX == 4 || op.steps2 <= 16) couple_m<EPL, KIND, 4>(op, prm, lane, q, src, tgt, ld);
        else if constexpr (HTMAX == 8) couple_m<EPL, KIND, 8>(op, prm, lane, q, src, tgt, ld);
    }
}

template <int EPL, int HTMAX, bool MADE, bool CTX = false>
__device__ __forceinline__ void apply_op_m(const MOp op, const float *prm, int lane, int q,
                                           float (&a)[EPL], float (&b)[EPL], float &ld,
                                           const float (&cx)[kCtxSteps] = {0.0f, 0.0f, 0.0f, 0.0f})
{
    constexpr int D = 8 * EPL, HALF = 4 * EPL;
    if constexpr (EPL <= 16) if (op.kind == TFK_OP_PLANE_SWAP) {      // (not at D = 256: 64 more live registers there)
        // odd event sizes (one element changes halves at every reversal): mask[HALF] != 0 exchanges the
        // elements at that index of the two planes -- both live in this lane's registers
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const float4 m = *reinterpret_cast<const float4 *>(prm + EPL * q + 4 * i);
            const float mk[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 4 * i + k;
                const float ta = a[e], tb = b[e];
                a[e] = mk[k] != 0.0f ? tb : ta;
                b[e] = mk[k] != 0.0f ? ta : tb;
            }
        }
        return;
    }
    if (op.kind == TFK_OP_EW_MULADD || op.kind == TFK_OP_EW_SUBDIV) {
        // this lane's EPL columns of each plane, four at a time: float4 reads (the address depends
        // on q only); chunked so that D = 256 (64 row registers per lane) does not spill
        const bool sub = (op.kind == TFK_OP_EW_SUBDIV);
        const float *ra = prm + 2 * D + 4;                          // 1/alpha (SUBDIV only)
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const float4 v0 = *reinterpret_cast<const float4 *>(prm + EPL * q + 4 * i);
            const float4 v1 = *reinterpret_cast<const float4 *>(prm + HALF + EPL * q + 4 * i);
            const float4 v2 = *reinterpret_cast<const float4 *>(prm + D + EPL * q + 4 * i);
            const float4 v3 = *reinterpret_cast<const float4 *>(prm + D + HALF + EPL * q + 4 * i);
            const float al_a[4] = {v0.x, v0.y, v0.z, v0.w}, al_b[4] = {v1.x, v1.y, v1.z, v1.w};
            const float be_a[4] = {v2.x, v2.y, v2.z, v2.w}, be_b[4] = {v3.x, v3.y, v3.z, v3.w};
            if (!sub) {                                             // affine.py:48
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * i + k;
                    a[e] = al_a[k] * a[e] + be_a[k];
                    b[e] = al_b[k] * b[e] + be_b[k];
                }
            } else {                                                // affine.py:59
                const float4 r0 = *reinterpret_cast<const float4 *>(ra + EPL * q + 4 * i);
                const float4 r1 = *reinterpret_cast<const float4 *>(ra + HALF + EPL * q + 4 * i);
                const float ra_a[4] = {r0.x, r0.y, r0.z, r0.w}, ra_b[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * i + k;
                    // (x - beta) * (1 / alpha), 1 / alpha rounded once on the host: within 1 ulp of the
                    // reference's division; the residual correction that made it the correctly rounded quotient
                    // cost 2 of the 4 operations per element on a datapath-bound kernel (section 3.4 of DESIGN.md)
                    a[e] = (a[e] - be_a[k]) * ra_a[k];
                    b[e] = (b[e] - be_b[k]) * ra_b[k];
                }
            }
        }
        if (q == 0) ld = ld + prm[2 * D];                           // base.py:222 (once per row)
        return;
    }
    if constexpr (CTX) {
        if (op.kind == TFK_OP_EWC_MULADD) { ewc_m<EPL, false>(op, prm, lane, q, a, b, ld, cx); return; }
        if (op.kind == TFK_OP_EWC_SUBDIV) { ewc_m<EPL, true>(op, prm, lane, q, a, b, ld, cx); return; }
    }
    // (src_plane bit 0 = 1: plane B conditions plane A)
    switch (op.kind * 2 + (op.src_plane & 1)) {
    case TFK_OP_AFFINE_FWD * 2: couple_any<EPL, TFK_OP_AFFINE_FWD, HTMAX, CTX>(op, prm, lane, q, a, b, ld, cx); return;
    case TFK_OP_AFFINE_FWD * 2 + 1: couple_any<EPL, TFK_OP_AFFINE_FWD, HTMAX, CTX>(op, prm, lane, q, b, a, ld, cx); return;
    case TFK_OP_AFFINE_INV * 2: couple_any<EPL, TFK_OP_AFFINE_INV, HTMAX, CTX>(op, prm, lane, q, a, b, ld, cx); return;
    case TFK_OP_AFFINE_INV * 2 + 1: couple_any<EPL, TFK_OP_AFFINE_INV, HTMAX, CTX>(op, prm, lane, q, b, a, ld, cx); return;
    case TFK_OP_SHIFT_FWD * 2: couple_any<EPL, TFK_OP_SHIFT_FWD, HTMAX, CTX>(op, prm, lane, q, a, b, ld, cx); return;
    case TFK_OP_SHIFT_FWD * 2 + 1: couple_any<EPL, TFK_OP_SHIFT_FWD, HTMAX, CTX>(op, prm, lane, q, b, a, ld, cx); return;
    case TFK_OP_SHIFT_INV * 2: couple_any<EPL, TFK_OP_SHIFT_INV, HTMAX, CTX>(op, prm, lane, q, a, b, ld, cx); return;
    case TFK_OP_SHIFT_INV * 2 + 1: couple_any<EPL, TFK_OP_SHIFT_INV, HTMAX, CTX>(op, prm, lane, q, b, a, ld, cx); return;
    default: break;
    }
    if constexpr (MADE) if (op.kind == TFK_OP_MADE_FWD || op.kind == TFK_OP_MADE_INV) {
        const bool div = (op.kind == TFK_OP_MADE_INV);
        if constexpr (HTMAX == 1) {
            if (div) made_m<EPL, true, 1>(op, prm, lane, q, a, b, ld);
            else made_m<EPL, false, 1>(op, prm, lane, q, a, b, ld);
        } else {
            if (op.steps2 <= 4) { if (div) made_m<EPL, true, 1>(op, prm, lane, q, a, b, ld); else made_m<EPL, false, 1>(op, prm, lane, q, a, b, ld); }
            else if (op.steps2 <= 8) { if (div) made_m<EPL, true, 2>(op, prm, lane, q, a, b, ld); else made_m<EPL, false, 2>(op, prm, lane, q, a, b, ld); }
            else if (HTMAX == 4 || op.steps2 <= 16) { if (div) made_m<EPL, true, 4>(op, prm, lane, q, a, b, ld); else made_m<EPL, false, 4>(op, prm, lane, q, a, b, ld); }
            else if constexpr (HTMAX == 8) { if (div) made_m<EPL, true, 8>(op, prm, lane, q, a, b, ld); else made_m<EPL, false, 8>(op, prm, lane, q, a, b, ld); }
        }
        return;
    }
    if constexpr (MADE && EPL <= kMaxEplRqs) if (op.kind == TFK_OP_MADE_RQS) {
        made_rqs_m<EPL>(op, prm, lane, q, a, b, ld);
        return;
    }
    // the spline op exists for D <= 128 only: at D = 256 one coupling's parameters (209 KB) exceed
    // the LDS anyway, and its 24-register record beside 64 row registers would live in scratch
    if constexpr (EPL <= kMaxEplRqs) {
        switch (op.kind * 2 + (op.src_plane & 1)) {
        case TFK_OP_RQS_FWD * 2: couple_rqs_m<EPL, false, CTX>(op, prm, lane, q, a, b, ld, cx); return;
        case TFK_OP_RQS_FWD * 2 + 1: couple_rqs_m<EPL, false, CTX>(op, prm, lane, q, b, a, ld, cx); return;
        case TFK_OP_RQS_INV * 2: couple_rqs_m<EPL, true, CTX>(op, prm, lane, q, a, b, ld, cx); return;
        case TFK_OP_RQS_INV * 2 + 1: couple_rqs_m<EPL, true, CTX>(op, prm, lane, q, b, a, ld, cx); return;
        default: break;
        }
    }
}

// Dynamic LDS: the parameter block [+ 3*D floats of base density].
// HTMAX = 1: every coupling of the program has hidden width <= 16 (the presets: 96 VGPRs at
// D = 64); HTMAX = 4: up to 64 (more registers: its own instantiation so that the narrow
// programs keep their occupancy).
// MADE: the program holds MADE ops (their own instantiation: they need more registers).
template <int EPL, int BLOCK, int HTMAX, bool MADE, bool CTX = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((EPL == 32 && BLOCK == 768) ? 3 : 1)))
void k_flow_run_mfma(
    const float *__restrict__ x, float *z, float *logdet, const float *__restrict__ gauss_loc,
    const float *__restrict__ gauss_log_scale, float *logprob, long long N,
    const float *__restrict__ params, int n_params, MProgram prog, int flags,
    const float *__restrict__ context = nullptr, int C = 0)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = 8 * EPL, HALF = 4 * EPL;
    const int accumulate = flags & 1;
    const bool reverse_out = (flags & 2) != 0;          // store z[row, D-1-c] = value of column c
    const bool base_of_input = (flags & 4) != 0;        // logprob = base density of the INPUT rows + log-det
    {
        const float4 *src = reinterpret_cast<const float4 *>(params);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < (n_params >> 2); i += BLOCK) dst[i] = src[i];
    }
    float *base_s = lds + n_params;                   // loc[D] | scale[D] | log_scale[D]
    if (logprob) {
        for (int e = threadIdx.x; e < D; e += BLOCK) {
            base_s[e] = gauss_loc[e];
            base_s[D + e] = expf(gauss_log_scale[e]);
            base_s[2 * D + e] = gauss_log_scale[e];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    constexpr int rows_per_block = (BLOCK / 64) * 16;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row0 = (long long)blockIdx.x * rows_per_block + wave * 16; row0 < N; row0 += stride) {
        const long long row = row0 + j;
        const long long rr = row < N ? row : N - 1;    // tail: compute a valid row, store nothing
        float a[EPL], b[EPL];
        const float4 *pa = reinterpret_cast<const float4 *>(x + rr * D + EPL * q);
        const float4 *pb = reinterpret_cast<const float4 *>(x + rr * D + HALF + EPL * q);
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const float4 va = pa[i], vb = pb[i];
            a[4 * i] = va.x; a[4 * i + 1] = va.y; a[4 * i + 2] = va.z; a[4 * i + 3] = va.w;
            b[4 * i] = vb.x; b[4 * i + 1] = vb.y; b[4 * i + 2] = vb.z; b[4 * i + 3] = vb.w;
        }
        // per-lane share of the row's log-det; lane q == 0 carries the running value
        float ld = (q == 0 && logdet && accumulate) ? logdet[rr] : 0.0f;
        float lp = 0.0f;
        if (logprob && base_of_input) {                             // Flow.sample: base_log_prob(z) of the rows
#pragma unroll                                                      // as they come in (flows.py:699-707)
            for (int e = 0; e < EPL; ++e) {
                const int ia = EPL * q + e, ib = HALF + EPL * q + e;
                const float ta = div_fast(a[e] - base_s[ia], base_s[D + ia]);
                const float tb = div_fast(b[e] - base_s[ib], base_s[D + ib]);
                float ua = 0.5f * (ta * ta), ub = 0.5f * (tb * tb);
                ua = ua + kHalfLog2Pi; ub = ub + kHalfLog2Pi;
                ua = ua + base_s[2 * D + ia]; ub = ub + base_s[2 * D + ib];
                lp += -ua;
                lp += -ub;
            }
        }
        float cx[kCtxSteps] = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (CTX) {                                          // this lane's share of the row's context
#pragma unroll
            for (int s = 0; s < kCtxSteps; ++s)
                cx[s] = (4 * s + q < C) ? context[rr * C + 4 * s + q] : 0.0f;
        }
        for (int o = 0; o < prog.n_ops; ++o)
            apply_op_m<EPL, HTMAX, MADE, CTX>(prog.op[o], lds + prog.op[o].offset, lane, q, a, b, ld, cx);
        if (logprob && !base_of_input) {                            // gaussian.py:46-54
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int ia = EPL * q + e, ib = HALF + EPL * q + e;
                const float ta = div_fast(a[e] - base_s[ia], base_s[D + ia]);
                const float tb = div_fast(b[e] - base_s[ib], base_s[D + ib]);
                float ua = 0.5f * (ta * ta), ub = 0.5f * (tb * tb);
                ua = ua + kHalfLog2Pi; ub = ub + kHalfLog2Pi;
                ua = ua + base_s[2 * D + ia]; ub = ub + base_s[2 * D + ib];
                lp += -ua;
                lp += -ub;
            }
        }
        // combine the four lanes of a row (l, l^16, l^32, l^48)
        ld += __shfl_xor(ld, 16, kWave);
        ld += __shfl_xor(ld, 32, kWave);
        if (logprob) {
            lp += __shfl_xor(lp, 16, kWave);
            lp += __shfl_xor(lp, 32, kWave);
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
                // a ReversePermutationMatrix after the program (matrix/permutation.py:34-37) folded
                // into the store: plane A lands mirrored in the second half, plane B in the first
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
                if (logprob) logprob[row] = lp + ld;                // flows.py:648
            }
        }
    }
}

template <int EPL, int BLOCK, int HTMAX, bool MADE, bool CTX = false>
static int launch_mb(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                     float *logprob, int64_t N, const float *params, int n_params,
                     const MProgram &prog, int accumulate, hipStream_t s, const char *fn,
                     const float *context = nullptr, int C = 0)
{
    constexpr int D = 8 * EPL;
    const size_t lds = ((size_t)n_params + 3 * D) * sizeof(float);
    if (lds > 160 * 1024)
        return fail(TFK_EINVAL, "%s: %zu bytes of parameters do not fit the 160 KiB LDS; split the program", fn, lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flow_run_mfma<EPL, BLOCK, HTMAX, MADE, CTX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", fn, lds, hipGetErrorString(e));
        }
    }
    // resident workgroups per CU as the runtime computes them (registers, LDS, wave slots); the
    // grid is a few resident sets, grid-strided over the rows (kGridOversubscribe, tfk_common.h)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_flow_run_mfma<EPL, BLOCK, HTMAX, MADE, CTX>, BLOCK, lds) != hipSuccess
        || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    constexpr int rows_per_block = (BLOCK / 64) * 16;
    const int64_t want = (N + rows_per_block - 1) / rows_per_block;
    const int64_t cap = (int64_t)cu_count() * per_cu * kGridOversubscribe;
    const int grid = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL((k_flow_run_mfma<EPL, BLOCK, HTMAX, MADE, CTX>), dim3(grid), dim3(BLOCK), lds, s, x, z, logdet, loc,
                       log_scale, logprob, (long long)N, params, n_params, prog, accumulate, context, C);
    return check_launch(fn);
}

// the parameter block is per workgroup: with a whole program resident (~50 KB for RealNVP
// D = 64, 8 layers; 96 VGPRs) two 512-thread workgroups = 16 waves fit a CU, 256-thread ones would
// stop at 3 x 4 waves on LDS; small batches take the 256-thread variant (more workgroups)
template <int EPL>
static int launch_m(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                    float *logprob, int64_t N, const float *params, int n_params,
                    const MProgram &prog, int accumulate, hipStream_t s, const char *fn,
                    const float *context = nullptr, int C = 0)
{
    if (context) {
        // context-conditioned programs: their own instantiation (hidden width <= 16, no MADE ops) so that the
        // others keep their registers
        const bool big_c = N >= (int64_t)cu_count() * 3 * 128;
        return big_c ? launch_mb<EPL, 512, 1, false, true>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog,
                                                          accumulate, s, fn, context, C)
                     : launch_mb<EPL, kBlock, 1, false, true>(x, z, logdet, loc, log_scale, logprob, N, params, n_params,
                                                             prog, accumulate, s, fn, context, C);
    }
    bool wide = false, wider = false;      // a coupling with hidden width > 16 (> 64) in the program?
    for (int i = 0; i < prog.n_ops; ++i)
        if ((prog.op[i].kind >= TFK_OP_AFFINE_FWD && prog.op[i].kind <= TFK_OP_SHIFT_INV) ||
            (prog.op[i].kind == TFK_OP_MADE_FWD || prog.op[i].kind == TFK_OP_MADE_INV)) {
            wide = wide || prog.op[i].steps2 > 4;
            wider = wider || prog.op[i].steps2 > 16;
        }
    const bool big = N >= (int64_t)cu_count() * 3 * 128;
    bool made = false;
    for (int i = 0; i < prog.n_ops; ++i)      // (TFK_OP_PLANE_SWAP = 11 lies above the MADE kinds: not a MADE op)
        made = made || (prog.op[i].kind >= TFK_OP_MADE_FWD && prog.op[i].kind <= TFK_OP_MADE_RQS);
#define TFK_MB(BLOCK_, HT_, MADE_) \
    launch_mb<EPL, BLOCK_, HT_, MADE_>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, accumulate, s, fn)
    // hidden width 65..128: 8 hidden tiles = 32 accumulators + 32 activations beside the row; one instantiation
    // (256 threads: such a coupling's operands are 48 KB at D = 64 and the LDS holds one per launch anyway)
    if (wider) return made ? TFK_MB(kBlock, 8, true) : TFK_MB(kBlock, 8, false);
    if (made) {
        if (wide) return big ? TFK_MB(512, 4, true) : TFK_MB(kBlock, 4, true);
        return big ? TFK_MB(512, 1, true) : TFK_MB(kBlock, 1, true);
    }
    if (wide) return big ? TFK_MB(512, 4, false) : TFK_MB(kBlock, 4, false);
    if constexpr (EPL == 32) {
        // D = 256: 64 row elements per lane = 181 VGPRs = 2 waves / SIMD, and the parameter block (~85 KB for
        // 4 layers) allows one workgroup per CU: 768 threads at 168 VGPRs put 3 waves on every SIMD
        if (big) return TFK_MB(768, 1, false);
    }
    return big ? TFK_MB(512, 1, false) : TFK_MB(kBlock, 1, false);
#undef TFK_MB
}

}  // namespace tfk
