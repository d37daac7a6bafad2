// tfk_flow.hip -- fused "flow program" interpreter: a whole chain of layers in ONE launch.
//
// conditioner-in-kernel fusion + layer folding.  The host (torchflows_amd/
// fused.py) compiles a BijectiveComposition made of
//     ElementwiseAffine / ActNorm        (layers_base.py:237-318, layers.py:19-69)
//     ReversePermutationMatrix           (matrix/permutation.py:8-37)
//     Affine / Shift coupling with a FeedForward(tanh) conditioner on the HalfSplit mask
//                                        (layers_base.py:51-163, conditioning/transforms.py:274-307)
// into a short list of ops over PHYSICAL element positions: the reversals are folded into
// the order in which weights are packed (they only decide which half is the conditioner's
// input and in which order the weights are stored), so no data ever moves for them.
//
// Execution model (D = 8*G, G in {2..64} a power of two):
//   * a row lives in the registers of G consecutive lanes of one wavefront: lane j holds
//     a = elements [4j, 4j+4) of plane A (first half) and b = the same of plane B;
//   * every op of the program is applied to the registers; rows are read once (16-byte
//     coalesced loads) and written at most once -- h, x_A copies, per-layer log-dets never
//     exist in HBM: 4*D + 4 bytes per log_prob evaluation instead of ~15 KB;
//   * the program's parameters (a few KB to tens of KB) are staged in LDS once per
//     workgroup and read as broadcast ds_read_b128 (all row-groups of a wave read the same
//     weights; the G lanes of a group read G consecutive float4 = conflict-free);
//   * conditioner: hidden_k = tanh(b1_k + sum_s W1[k,s] x_s) -- each lane does its 4-element
//     partial dot product, the G partials are summed with DPP (quad_perm / row mirrors) or
//     ds_swizzle butterflies; hidden_k is immediately folded into the lane's 8 (affine) or 4
//     (shift) output accumulators, so no hidden vector is ever stored;
//   * per-row log-det stays in a register and is reduced with the same butterflies.
// With the conditioner in the kernel the work is fp32-VALU-bound, not HBM-bound.
//
// Arithmetic: the transform itself keeps the reference's op order (affine.py:33-59,
// -ffp-contract=off); the dot products use fmaf and a different (tree) summation order than
// ATen's GEMM -- both are within a few ulp of the exact sum.
#include <cmath>
#include <cstring>

#include "tfk_common.h"
#include "tfk_spline.h"

namespace tfk {

constexpr int kMaxOps = 96;
constexpr int kMaxHidden = 32;     // conditioner width an RQS op may have (activations parked in LDS)
constexpr int kRqsPad = 24;        // 3*8-1 = 23 spline parameters per element, padded to 6 float4

struct FlowOp {
    int kind;        // TFK_OP_*
    int src_plane;   // coupling: 0 = plane A conditions plane B, 1 = the opposite
    int H;           // coupling: hidden width of the conditioner
    int offset;      // first float of this op's parameters in the staged block (multiple of 4)
    int K;           // RQS: number of bins (8)
    float boundary;  // RQS: spline box half-width
    float scale;     // RQS: 1 - min_bin_size * K  (python double, cast once)
    float c;         // RQS: boundary_u_delta = log(expm1(1 - min_delta))
};

struct FlowProgram {
    int n_ops;
    int pad[3];
    FlowOp op[kMaxOps];
};

// ---- cross-lane sums inside a G-lane group (result in every lane of the group) ----------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(moved);
}

template <int G>
__device__ __forceinline__ float group_allsum(float v) {
    if (G >= 2) v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]: lane ^ 1
    if (G >= 4) v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]: lane ^ 2
    if (G >= 8) v = dpp_add<0x141>(v);    // row_half_mirror: i <-> 7 - i   (quads are uniform)
    if (G >= 16) v = dpp_add<0x140>(v);   // row_mirror: i <-> 15 - i       (halves are uniform)
    if (G >= 32) v += __shfl_xor(v, 16, kWave);
    if (G >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// tanh of a hidden pre-activation, branch-free: 1 - 2 / (1 + exp(2x)) with the hardware
// exp2 and reciprocal (v_exp_f32 / v_rcp_f32, ~1 ulp each).  Absolute error <= ~2e-7, which
// is what matters for a conditioner activation (it is multiplied by O(1) weights and added);
// ocml's tanhf is a two-branch routine that diverges across the rows of a wave and costs
// ~4x as many instructions.  Saturates correctly: exp2(+big) = inf -> 1, exp2(-big) = 0 -> -1.
__device__ __forceinline__ float tanh_act(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // exp(2x)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// 4-element dot product, scalar FMAs on purpose: on gfx950 packed fp32 ops (v_pk_mul_f32 /
// v_pk_fma_f32) issue slower than the two scalar ops they replace -- measured here: the
// packed form of this function made the whole RealNVP program 29 % slower, and building with
// -fno-slp-vectorize (no compiler-made v_pk_*) gained another 7-10 % on every kernel.
__device__ __forceinline__ float dot4(const float4 w, const float4 s) {
    return fmaf(w.w, s.w, fmaf(w.z, s.z, fmaf(w.y, s.y, w.x * s.x)));
}

// One op on the registers of R rows (the weights are read from LDS once for all of them).
// prm = this op's parameters in LDS.
template <int G, int R>
__device__ __forceinline__ void apply_op(const FlowOp op, const float *prm, int j, float4 (&a)[R],
                                         float4 (&b)[R], float (&ld)[R], float *hk_s)
{
    constexpr int D = 8 * G, HALF = 4 * G;
    if (op.kind == TFK_OP_EW_MULADD || op.kind == TFK_OP_EW_SUBDIV) {
        // alpha[D] | beta[D] | log-det constant (already signed)
        const float4 al_a = *reinterpret_cast<const float4 *>(prm + 4 * j);
        const float4 al_b = *reinterpret_cast<const float4 *>(prm + HALF + 4 * j);
        const float4 be_a = *reinterpret_cast<const float4 *>(prm + D + 4 * j);
        const float4 be_b = *reinterpret_cast<const float4 *>(prm + D + HALF + 4 * j);
        const float ldc = prm[2 * D];
        // (x - beta) / alpha with the correctly rounded reciprocal r = 1/alpha packed by the host:
        // q = n*r; q += r*(n - alpha*q) -- the IEEE quotient except on near-ties (see div_fast)
        float4 ra_a = al_a, ra_b = al_b;
        if (op.kind == TFK_OP_EW_SUBDIV) {
            ra_a = *reinterpret_cast<const float4 *>(prm + 2 * D + 4 + 4 * j);
            ra_b = *reinterpret_cast<const float4 *>(prm + 2 * D + 4 + HALF + 4 * j);
        }
        auto quot = [](float n, float d, float r) {
            const float q = n * r;
            return fmaf(fmaf(-d, q, n), r, q);
        };
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (op.kind == TFK_OP_EW_MULADD) {                  // affine.py:48
                a[r].x = al_a.x * a[r].x + be_a.x; a[r].y = al_a.y * a[r].y + be_a.y;
                a[r].z = al_a.z * a[r].z + be_a.z; a[r].w = al_a.w * a[r].w + be_a.w;
                b[r].x = al_b.x * b[r].x + be_b.x; b[r].y = al_b.y * b[r].y + be_b.y;
                b[r].z = al_b.z * b[r].z + be_b.z; b[r].w = al_b.w * b[r].w + be_b.w;
            } else {                                            // affine.py:59
                a[r].x = quot(a[r].x - be_a.x, al_a.x, ra_a.x); a[r].y = quot(a[r].y - be_a.y, al_a.y, ra_a.y);
                a[r].z = quot(a[r].z - be_a.z, al_a.z, ra_a.z); a[r].w = quot(a[r].w - be_a.w, al_a.w, ra_a.w);
                b[r].x = quot(b[r].x - be_b.x, al_b.x, ra_b.x); b[r].y = quot(b[r].y - be_b.y, al_b.y, ra_b.y);
                b[r].z = quot(b[r].z - be_b.z, al_b.z, ra_b.z); b[r].w = quot(b[r].w - be_b.w, al_b.w, ra_b.w);
            }
            ld[r] = ld[r] + ldc;                                // base.py:222
        }
        return;
    }
    const int H = op.H, H4 = (H + 3) & ~3;
    if (op.kind == TFK_OP_RQS_FWD || op.kind == TFK_OP_RQS_INV) {
        // RQ-spline coupling (layers.py:154-163): W1t[H][HALF] | b1[H4] | W2t[H][4][G][24] | b2[4][G][24]
        // -- the 4 target elements of lane j are (e, j), e = 0..3, physical position 4j + e;
        // 23 spline parameters each, padded to 24 so a lane's slices are 6 aligned float4 and
        // the G lanes of a group read 96-byte-strided slices (bank-conflict-free).
        const float *W1t = prm;
        const float *b1 = prm + H * HALF;
        const float *W2t = b1 + H4;
        const float *b2 = W2t + H * HALF * kRqsPad;
        RqsConst C;
        C.minimum = -op.boundary;
        C.maximum = op.boundary;
        C.span = op.boundary + op.boundary;
        C.scale = op.scale;
        C.c = op.c;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 src = op.src_plane ? b[r] : a[r];
            float4 tgt = op.src_plane ? a[r] : b[r];
            // conditioner layer 1: the hidden activations go to this row-group's LDS scratch
            // (same wave writes and reads them: program order, no barrier)
            for (int k = 0; k < H; ++k) {
                const float4 w1 = *reinterpret_cast<const float4 *>(W1t + k * HALF + 4 * j);
                const float hk = tanh_act(group_allsum<G>(dot4(w1, src)) + b1[k]);
                if (j == 0) hk_s[k] = hk;
            }
            float part = 0.0f;
            for (int e = 0; e < 4; ++e) {
                // conditioner layer 2 for this element: its 23 (+1 pad) spline parameters
                float p[kRqsPad];
                const float4 *bb = reinterpret_cast<const float4 *>(b2 + (e * G + j) * kRqsPad);
#pragma unroll
                for (int i = 0; i < kRqsPad / 4; ++i) {
                    const float4 q = bb[i];
                    p[4 * i] = q.x; p[4 * i + 1] = q.y; p[4 * i + 2] = q.z; p[4 * i + 3] = q.w;
                }
                for (int k = 0; k < H; ++k) {
                    const float hk = hk_s[k];
                    const float4 *ww = reinterpret_cast<const float4 *>(W2t + ((k * 4 + e) * G + j) * kRqsPad);
#pragma unroll
                    for (int i = 0; i < kRqsPad / 4; ++i) {
                        const float4 q = ww[i];
                        p[4 * i] = fmaf(q.x, hk, p[4 * i]);
                        p[4 * i + 1] = fmaf(q.y, hk, p[4 * i + 1]);
                        p[4 * i + 2] = fmaf(q.z, hk, p[4 * i + 2]);
                        p[4 * i + 3] = fmaf(q.w, hk, p[4 * i + 3]);
                    }
                }
                const float v = e == 0 ? tgt.x : (e == 1 ? tgt.y : (e == 2 ? tgt.z : tgt.w));
                float o = v, l = 0.0f;                          // spline/base.py:54-55
                if (v > C.minimum && v < C.maximum) {           // strict box, base.py:29-33
                    if (op.kind == TFK_OP_RQS_FWD)
                        rqs_eval<8, false, true, float[kRqsPad]>(p, 8, v, C, o, l);
                    else
                        rqs_eval<8, true, true, float[kRqsPad]>(p, 8, v, C, o, l);
                }
                if (e == 0) tgt.x = o; else if (e == 1) tgt.y = o; else if (e == 2) tgt.z = o; else tgt.w = o;
                part += l;
            }
            ld[r] = ld[r] + group_allsum<G>(part);              // base.py:59 + :222
            if (op.src_plane) a[r] = tgt; else b[r] = tgt;
        }
        return;
    }
    // affine / shift coupling: W1t[H][HALF] | b1[H4] | W2t[H][HALF*P] | b2[HALF*P]   (physical order)
    const bool affine = (op.kind == TFK_OP_AFFINE_FWD || op.kind == TFK_OP_AFFINE_INV);
    float4 src[R], tgt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        src[r] = op.src_plane ? b[r] : a[r];
        tgt[r] = op.src_plane ? a[r] : b[r];
    }
    const float *W1t = prm;
    const float *b1 = prm + H * HALF;
    if (affine) {
        const float *W2t = b1 + H4;
        const float *b2 = W2t + H * (2 * HALF);
        float4 acc0[R], acc1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc0[r] = *reinterpret_cast<const float4 *>(b2 + 8 * j);      // (u0, be0, u1, be1)
            acc1[r] = *reinterpret_cast<const float4 *>(b2 + 8 * j + 4);  // (u2, be2, u3, be3)
        }
        for (int k = 0; k < H; ++k) {
            const float4 w1 = *reinterpret_cast<const float4 *>(W1t + k * HALF + 4 * j);
            const float b1k = b1[k];
            const float4 w2a = *reinterpret_cast<const float4 *>(W2t + k * (2 * HALF) + 8 * j);
            const float4 w2b = *reinterpret_cast<const float4 *>(W2t + k * (2 * HALF) + 8 * j + 4);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float hk = tanh_act(group_allsum<G>(dot4(w1, src[r])) + b1k);   // transforms.py:293-304
                acc0[r].x = fmaf(w2a.x, hk, acc0[r].x); acc0[r].y = fmaf(w2a.y, hk, acc0[r].y);
                acc0[r].z = fmaf(w2a.z, hk, acc0[r].z); acc0[r].w = fmaf(w2a.w, hk, acc0[r].w);
                acc1[r].x = fmaf(w2b.x, hk, acc1[r].x); acc1[r].y = fmaf(w2b.y, hk, acc1[r].y);
                acc1[r].z = fmaf(w2b.z, hk, acc1[r].z); acc1[r].w = fmaf(w2b.w, hk, acc1[r].w);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float a0 = aff_alpha(acc0[r].x), a1 = aff_alpha(acc0[r].z);
            const float a2 = aff_alpha(acc1[r].x), a3 = aff_alpha(acc1[r].z);
            float part = log_normal(a0);                               // affine.py:42
            part += log_normal(a1);
            part += log_normal(a2);
            part += log_normal(a3);
            part = group_allsum<G>(part);
            if (op.kind == TFK_OP_AFFINE_FWD) {
                tgt[r].x = a0 * tgt[r].x + acc0[r].y; tgt[r].y = a1 * tgt[r].y + acc0[r].w;
                tgt[r].z = a2 * tgt[r].z + acc1[r].y; tgt[r].w = a3 * tgt[r].w + acc1[r].w;
                ld[r] = ld[r] + part;
            } else {
                tgt[r].x = div_fast(tgt[r].x - acc0[r].y, a0); tgt[r].y = div_fast(tgt[r].y - acc0[r].w, a1);
                tgt[r].z = div_fast(tgt[r].z - acc1[r].y, a2); tgt[r].w = div_fast(tgt[r].w - acc1[r].w, a3);
                ld[r] = ld[r] + (-part);
            }
            if (op.src_plane) a[r] = tgt[r]; else b[r] = tgt[r];
        }
    } else {                                                 // shift (NICE), affine.py:137-159
        const float *W2t = b1 + H4;
        const float *b2 = W2t + H * HALF;
        float4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = *reinterpret_cast<const float4 *>(b2 + 4 * j);
        for (int k = 0; k < H; ++k) {
            const float4 w1 = *reinterpret_cast<const float4 *>(W1t + k * HALF + 4 * j);
            const float b1k = b1[k];
            const float4 w2 = *reinterpret_cast<const float4 *>(W2t + k * HALF + 4 * j);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float hk = tanh_act(group_allsum<G>(dot4(w1, src[r])) + b1k);
                acc[r].x = fmaf(w2.x, hk, acc[r].x); acc[r].y = fmaf(w2.y, hk, acc[r].y);
                acc[r].z = fmaf(w2.z, hk, acc[r].z); acc[r].w = fmaf(w2.w, hk, acc[r].w);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (op.kind == TFK_OP_SHIFT_FWD) {
                tgt[r].x += acc[r].x; tgt[r].y += acc[r].y; tgt[r].z += acc[r].z; tgt[r].w += acc[r].w;
            } else {
                tgt[r].x -= acc[r].x; tgt[r].y -= acc[r].y; tgt[r].z -= acc[r].z; tgt[r].w -= acc[r].w;
            }
            if (op.src_plane) a[r] = tgt[r]; else b[r] = tgt[r];
        }
    }
}

// Dynamic LDS: the launch's parameter block (n_params floats) [+ 3*D floats for the base].
template <int G, int BLOCK, int R>
__global__ __launch_bounds__(BLOCK) void k_flow_run(
    const float4 *__restrict__ x, float4 *z, float *logdet, const float *__restrict__ gauss_loc,
    const float *__restrict__ gauss_log_scale, float *logprob, long long N,
    const float *__restrict__ params, int n_params, FlowProgram prog, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = 8 * G, HALF = 4 * G;
    {   // stage the parameter block (16-byte coalesced loads)
        const float4 *src = reinterpret_cast<const float4 *>(params);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < (n_params >> 2); i += BLOCK) dst[i] = src[i];
    }
    float *base_s = lds + n_params;                   // loc[D] | scale[D] | log_scale[D]
    // per row-group scratch for RQS ops (hidden activations), after the base block
    float *hk_s = lds + n_params + 3 * (8 * G) + (threadIdx.x / G) * kMaxHidden;
    if (logprob) {
        for (int e = threadIdx.x; e < D; e += BLOCK) {
            base_s[e] = gauss_loc[e];
            base_s[D + e] = expf(gauss_log_scale[e]);
            base_s[2 * D + e] = gauss_log_scale[e];
        }
    }
    __syncthreads();

    const int j = threadIdx.x & (G - 1);
    constexpr int rows_per_pass = BLOCK / G;           // rows one workgroup covers per r
    constexpr int rows_per_block = R * rows_per_pass;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long row0 = (long long)blockIdx.x * rows_per_block + threadIdx.x / G; row0 < N;
         row0 += stride) {
        float4 a[R], b[R];
        float ld[R];
        long long row[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            row[r] = row0 + (long long)r * rows_per_pass;
            const long long rr = row[r] < N ? row[r] : N - 1;     // tail: compute a valid row, store nothing
            a[r] = x[rr * (2 * G) + j];
            b[r] = x[rr * (2 * G) + G + j];
            // running log-det: continues the previous launch's sum so that the layer-order fp32
            // accumulation of base.py:210-222 is the same however the program is segmented
            ld[r] = (logdet && accumulate) ? logdet[rr] : 0.0f;
        }
        for (int o = 0; o < prog.n_ops; ++o)
            apply_op<G, R>(prog.op[o], lds + prog.op[o].offset, j, a, b, ld, hk_s);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = 0.0f;
            if (logprob) {                                          // gaussian.py:46-54
                auto term = [&](float v, int e) {
                    const float t = div_fast(v - base_s[e], base_s[D + e]);
                    float q = 0.5f * (t * t);
                    q = q + kHalfLog2Pi;
                    q = q + base_s[2 * D + e];
                    acc += -q;
                };
                term(a[r].x, 4 * j); term(a[r].y, 4 * j + 1); term(a[r].z, 4 * j + 2); term(a[r].w, 4 * j + 3);
                term(b[r].x, HALF + 4 * j); term(b[r].y, HALF + 4 * j + 1);
                term(b[r].z, HALF + 4 * j + 2); term(b[r].w, HALF + 4 * j + 3);
                acc = group_allsum<G>(acc);
            }
            if (row[r] < N) {
                if (z) {
                    z[row[r] * (2 * G) + j] = a[r];
                    z[row[r] * (2 * G) + G + j] = b[r];
                }
                if (j == 0) {
                    if (logdet) logdet[row[r]] = ld[r];
                    if (logprob) logprob[row[r]] = acc + ld[r];     // flows.py:648
                }
            }
        }
    }
}

template <int G, int BLOCK, int R>
static int launch_flow_b(const float *x, float *z, float *logdet, const float *loc,
                         const float *log_scale, float *logprob, int64_t N, const float *params,
                         int n_params, const FlowProgram &prog, int accumulate, hipStream_t s,
                         const char *fn)
{
    constexpr int D = 8 * G;
    bool has_rqs = false;
    for (int i = 0; i < prog.n_ops; ++i)
        has_rqs = has_rqs || prog.op[i].kind == TFK_OP_RQS_FWD || prog.op[i].kind == TFK_OP_RQS_INV;
    const size_t lds = ((size_t)n_params + 3 * D + (has_rqs ? (BLOCK / G) * kMaxHidden : 0)) * sizeof(float);
    if (lds > 160 * 1024)
        return fail(TFK_EINVAL, "%s: %zu bytes of parameters do not fit the 160 KiB LDS; split the program", fn, lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flow_run<G, BLOCK, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", fn, lds, hipGetErrorString(e));
        }
    }
    // residency: 32 waves per CU and 160 KiB of LDS; the parameter block is per workgroup,
    // so big workgroups share it between more waves
    int per_cu = 2048 / BLOCK;
    if (lds && (int)((160 * 1024) / lds) < per_cu) per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    constexpr int rows_per_block = R * (BLOCK / G);
    const int64_t want = (N + rows_per_block - 1) / rows_per_block;
    const int64_t cap = (int64_t)cu_count() * per_cu * kGridOversubscribe;
    const int grid = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL((k_flow_run<G, BLOCK, R>), dim3(grid), dim3(BLOCK), lds, s,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(z), logdet,
                       loc, log_scale, logprob, (long long)N, params, n_params, prog, accumulate);
    return check_launch(fn);
}

template <int G>
static int launch_flow(const float *x, float *z, float *logdet, const float *loc,
                       const float *log_scale, float *logprob, int64_t N, const float *params,
                       int n_params, const FlowProgram &prog, int accumulate, hipStream_t s,
                       const char *fn)
{
    // enough rows to fill the chip with 1024-thread workgroups (2 per CU)? else stay small
    // programs with RQ-spline ops are register-heavy (23 parameters + the spline state per
    // element): 512-thread workgroups, one row per thread
    for (int i = 0; i < prog.n_ops; ++i)
        if (prog.op[i].kind == TFK_OP_RQS_FWD || prog.op[i].kind == TFK_OP_RQS_INV) {
            if (N * G >= (int64_t)cu_count() * 512)
                return launch_flow_b<G, 512, 1>(x, z, logdet, loc, log_scale, logprob, N, params,
                                                n_params, prog, accumulate, s, fn);
            return launch_flow_b<G, kBlock, 1>(x, z, logdet, loc, log_scale, logprob, N, params,
                                               n_params, prog, accumulate, s, fn);
        }
    // (each thread then carries two rows: the weights come out of LDS once for both)
    if (N * G >= (int64_t)cu_count() * 2 * 1024 * 2)
        return launch_flow_b<G, 1024, 2>(x, z, logdet, loc, log_scale, logprob, N, params, n_params,
                                         prog, accumulate, s, fn);
    return launch_flow_b<G, kBlock, 1>(x, z, logdet, loc, log_scale, logprob, N, params, n_params,
                                       prog, accumulate, s, fn);
}

}  // namespace tfk

using namespace tfk;

extern "C" {

int tfk_flow_supported(int32_t D)
{
    return (D >= 16 && D <= 512 && (D & (D - 1)) == 0) ? 1 : 0;
}

int tfk_flow_run(const float *x, float *z, float *logdet, const float *gauss_loc,
                 const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                 const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                 int32_t accumulate, void *stream)
{
    const char *fn = "tfk_flow_run";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (!tfk_flow_supported(D))
        return fail(TFK_EINVAL, "%s: D = %d is not a power of two in [16, 512]", fn, D);
    if (n_ops < 0 || n_ops > kMaxOps)
        return fail(TFK_EINVAL, "%s: n_ops = %d must be in [0, %d]", fn, n_ops, kMaxOps);
    if (n_params < 0 || (n_params & 3))
        return fail(TFK_EINVAL, "%s: n_params = %lld must be a non-negative multiple of 4", fn, (long long)n_params);
    if (N == 0) return TFK_OK;
    if (!x || (n_ops > 0 && (!ops || !params)))
        return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!z && !logdet && !logprob) return fail(TFK_EINVAL, "%s: no output requested", fn);
    if (logprob && (!gauss_loc || !gauss_log_scale))
        return fail(TFK_EINVAL, "%s: logprob needs gauss_loc and gauss_log_scale", fn);
    if (!aligned16(x) || (z && !aligned16(z)) || !aligned16(params))
        return fail(TFK_EINVAL, "%s: x, z and params must be 16-byte aligned", fn);

    FlowProgram prog;
    prog.n_ops = n_ops;
    const int HALF = D / 2;
    for (int i = 0; i < n_ops; ++i) {
        FlowOp &o = prog.op[i];
        const int32_t *rec = ops + 8 * i;
        o.kind = rec[0];
        o.src_plane = rec[1];
        o.H = rec[2];
        o.offset = rec[3];
        o.K = rec[4];
        memcpy(&o.boundary, rec + 5, 4);
        memcpy(&o.scale, rec + 6, 4);
        memcpy(&o.c, rec + 7, 4);
        int64_t need;
        if (o.kind == TFK_OP_EW_MULADD) {
            need = 2 * (int64_t)D + 4;
        } else if (o.kind == TFK_OP_EW_SUBDIV) {
            need = 3 * (int64_t)D + 4;
        } else if (o.kind >= TFK_OP_AFFINE_FWD && o.kind <= TFK_OP_SHIFT_INV) {
            const int P = (o.kind <= TFK_OP_AFFINE_INV) ? 2 : 1;
            if (o.H < 1 || o.H > 4096) return fail(TFK_EINVAL, "%s: op %d: hidden width %d", fn, i, o.H);
            if (o.src_plane != 0 && o.src_plane != 1) return fail(TFK_EINVAL, "%s: op %d: src_plane %d", fn, i, o.src_plane);
            need = (int64_t)o.H * HALF + ((o.H + 3) & ~3) + (int64_t)o.H * HALF * P + (int64_t)HALF * P;
        } else if (o.kind == TFK_OP_RQS_FWD || o.kind == TFK_OP_RQS_INV) {
            if (o.H < 1 || o.H > kMaxHidden) return fail(TFK_EINVAL, "%s: op %d: hidden width %d not in [1, %d]", fn, i, o.H, kMaxHidden);
            if (o.src_plane != 0 && o.src_plane != 1) return fail(TFK_EINVAL, "%s: op %d: src_plane %d", fn, i, o.src_plane);
            if (o.K != 8) return fail(TFK_EINVAL, "%s: op %d: fused RQS supports n_bins = 8, got %d", fn, i, o.K);
            if (!(o.boundary > 0.0f)) return fail(TFK_EINVAL, "%s: op %d: boundary must be positive", fn, i);
            need = (int64_t)o.H * HALF + ((o.H + 3) & ~3) + (int64_t)o.H * HALF * kRqsPad + (int64_t)HALF * kRqsPad;
        } else {
            return fail(TFK_EINVAL, "%s: op %d: unknown kind %d", fn, i, o.kind);
        }
        if (o.offset < 0 || (o.offset & 3) || o.offset + need > n_params)
            return fail(TFK_EINVAL, "%s: op %d: parameters [%d, %lld) outside the block of %lld floats",
                        fn, i, o.offset, (long long)(o.offset + need), (long long)n_params);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int np = (int)n_params;
    switch (D / 8) {
    case 2: return launch_flow<2>(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, np, prog, accumulate, s, fn);
    case 4: return launch_flow<4>(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, np, prog, accumulate, s, fn);
    case 8: return launch_flow<8>(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, np, prog, accumulate, s, fn);
    case 16: return launch_flow<16>(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, np, prog, accumulate, s, fn);
    case 32: return launch_flow<32>(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, np, prog, accumulate, s, fn);
    default: return launch_flow<64>(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, np, prog, accumulate, s, fn);
    }
}

}  // extern "C"
