// tfk_flow_mfma.hip -- C-ABI entry point of the matrix-core flow programs (kernels: tfk_flow_mfma.h,
// instantiated per row width in tfk_flow_mfma_{8,16,32}.hip).
#include "tfk_flow_mfma.h"
#include "tfk_flow_chain.h"
#include "tfk_flow_rqs_chain.h"

namespace tfk {
int flow_chain_launch_2(const float *, float *, float *, const float *, const float *, float *, int64_t,
                        const float *, int, const ChainProg &, int, int, int, int, hipStream_t, const char *);
int flow_chain_launch_4(const float *, float *, float *, const float *, const float *, float *, int64_t,
                        const float *, int, const ChainProg &, int, int, int, int, hipStream_t, const char *);
int flow_chain_launch_8(const float *, float *, float *, const float *, const float *, float *, int64_t,
                        const float *, int, const ChainProg &, int, int, int, int, hipStream_t, const char *);
int flow_chain_launch_16(const float *, float *, float *, const float *, const float *, float *, int64_t,
                         const float *, int, const ChainProg &, int, int, int, int, hipStream_t, const char *);
int flow_chain_launch_32(const float *, float *, float *, const float *, const float *, float *, int64_t,
                         const float *, int, const ChainProg &, int, int, int, int, hipStream_t, const char *);

int flow_rqs_chain_launch_4(const float *, float *, float *, const float *, const float *, float *, int64_t,
                            const float *, const RqsChainProg &, int, int, int, int, hipStream_t, const char *, const float *, int);
int flow_rqs_chain_launch_8(const float *, float *, float *, const float *, const float *, float *, int64_t,
                            const float *, const RqsChainProg &, int, int, int, int, hipStream_t, const char *, const float *, int);
int flow_rqs_chain_launch_16(const float *, float *, float *, const float *, const float *, float *, int64_t,
                             const float *, const RqsChainProg &, int, int, int, int, hipStream_t, const char *, const float *, int);
int flow_rqs_chain_launch_32(const float *, float *, float *, const float *, const float *, float *, int64_t,
                             const float *, const RqsChainProg &, int, int, int, int, hipStream_t, const char *, const float *, int);

// A program of lean spline couplings (TFK_OP_RQS_*_LEAN of one direction, one hidden width, one spline box, blocks
// at a constant stride, source plane alternating; optionally ended by one TFK_OP_EW_FMA): ONE launch of
// tfk_flow_rqs_chain.h, the operands streamed layer by layer from `params` (global memory).
static int run_rqs_chain(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                         float *logprob, int64_t N, int32_t D, const int32_t *ops, int32_t n_ops, const float *params,
                         int64_t n_params, int32_t flags, int32_t xw, hipStream_t s, const char *fn,
                         double *sum_ws, double *sum_out, const float *context = nullptr, int32_t Cn = 0)
{
    const int EPL = D / 8, HALF = D / 2;
    const int cs_want = context ? (Cn + 3) / 4 : 0;
    // operand format: K = 8 -> fp32 A-operands (chunks of 8 elements); K = 8 + 256 -> bf16 x 3 (chunks of 4 elements)
    int first = 0;                                           // the first spline op (elementwise ops may precede it)
    while (first < n_ops - 1 && (ops[8 * first] == TFK_OP_EW_FMA || ops[8 * first] == TFK_OP_EWC_MULADD ||
                                 ops[8 * first] == TFK_OP_EWC_SUBDIV))
        ++first;
    const bool fmt3 = n_ops > 0 && (ops[8 * first + 4] >> 8) == 1;
    // bf16 x 3: gemm2_steps = ceil((H + 1) / 4) counts the bias unit; more than 4 steps = two hidden tiles
    const int ht3 = (fmt3 && n_ops > 0 && ops[8 * first + 2] > 4) ? 2 : 1;
    int k0 = -1;                                             // the first spline op (context programs: never the first op? it is)
    for (int i = 0; i < n_ops && k0 < 0; ++i)
        if (ops[8 * i] != TFK_OP_EW_FMA && ops[8 * i] != TFK_OP_EWC_MULADD && ops[8 * i] != TFK_OP_EWC_SUBDIV) k0 = ops[8 * i];
    const bool made = k0 >= TFK_OP_MADE_RQS_FWD_LEAN && k0 <= TFK_OP_MADE_LRS_INV_LEAN;
    const bool lrs = k0 == TFK_OP_LRS_FWD_LEAN || k0 == TFK_OP_LRS_INV_LEAN || k0 == TFK_OP_MADE_LRS_FWD_LEAN ||
                     k0 == TFK_OP_MADE_LRS_INV_LEAN;
    if (made && (context || !fmt3 || ht3 != 1 || (EPL != 8 && EPL != 16)))
        return fail(TFK_EINVAL, "%s: lean MADE spline layers: bf16 x 3 operands, hidden width <= 15, D = 64 or 128, no context", fn);
    if (lrs && !fmt3) return fail(TFK_EINVAL, "%s: lean linear-rational-spline ops use the bf16 x 3 operand format (K = 8 + 256)", fn);
    if (context && !fmt3) return fail(TFK_EINVAL, "%s: context-conditioned lean spline chains use the bf16 x 3 operand format", fn);
    const int64_t block = made ? (int64_t)2 * EPL * 64 + 16 + 2 * D + (int64_t)(2 * EPL / 4) * (lrs ? 16384 : kRqsChunk3Dwords)
                        : fmt3 ? (int64_t)EPL * ht3 * 64 + ht3 * 16 + 2 * HALF + (cs_want ? ht3 * 256 : 0)
                                     + (int64_t)(EPL * ht3 / 4) * (lrs ? 16384 : kRqsChunk3Dwords)
                               : (int64_t)EPL * 64 + 16 + 2 * HALF + (int64_t)(EPL / 8) * kRqsChunkFloats;
    RqsChainProg prog;
    memset(&prog, 0, sizeof(prog));
    prog.ew_offset = -1;
    prog.sum_ws = sum_ws;
    prog.sum_out = sum_out;
    int kind = -1, steps2 = 0;
    float boundary = 0.0f, scale = 0.0f, c = 0.0f;
    for (int i = 0; i < n_ops; ++i) {
        const int32_t *rec = ops + 8 * i;
        const int k = rec[0], st = rec[2], off = rec[3];
        int src = rec[1];
        const bool ewc = (k == TFK_OP_EWC_MULADD || k == TFK_OP_EWC_SUBDIV);
        if (k == TFK_OP_EW_FMA || ewc) {
            // plain programs: one TFK_OP_EW_FMA, last.  Context programs: elementwise ops (constant or context-conditioned)
            // may also stand in front of the couplings and behind the closing TFK_OP_EW_FMA (staged in the LDS)
            if (ewc && (!context || (src >> 4) != cs_want))
                return fail(TFK_EINVAL, "%s: op %d: a context-conditioned elementwise op needs the call's context", fn, i);
            const int64_t need = ewc ? (int64_t)EPL * cs_want * 64 + (int64_t)EPL * 16 : 2 * (int64_t)D + 4;
            if (off < 0 || (off & 3) || off + need > n_params)
                return fail(TFK_EINVAL, "%s: op %d: parameters outside the block", fn, i);
            const int phase = prog.n_layers == 0 ? 0 : (prog.ew_offset < 0 ? 1 : 2);
            if (phase == 1 && !ewc) { prog.ew_offset = off; continue; }
            if (phase == 1) return fail(TFK_EINVAL, "%s: op %d: a TFK_OP_EW_FMA closes the couplings before this op", fn, i);
            if (!context) {
                if (i != n_ops - 1) return fail(TFK_EINVAL, "%s: op %d: TFK_OP_EW_FMA must end a lean program", fn, i);
                prog.ew_offset = off;
                continue;
            }
            int *kinds = phase == 0 ? prog.pre_kind : prog.post_kind, *offs = phase == 0 ? prog.pre_off : prog.post_off;
            int n = 0;
            while (n < kChainSideOps && kinds[n]) ++n;
            if (n == kChainSideOps) return fail(TFK_EINVAL, "%s: op %d: more than %d elementwise ops on one side of the couplings", fn, i, kChainSideOps);
            if (EPL > 16) return fail(TFK_EINVAL, "%s: op %d: elementwise ops inside a lean spline program need D <= 128", fn, i);
            kinds[n] = ewc ? (k == TFK_OP_EWC_MULADD ? 2 : 3) : 1;
            offs[n] = off;
            (phase == 0 ? prog.pre_lds : prog.post_lds)[n] = prog.side_floats;
            prog.side_floats += (int)need;
            continue;
        }
        if (prog.ew_offset >= 0) return fail(TFK_EINVAL, "%s: op %d: a coupling behind the closing TFK_OP_EW_FMA", fn, i);
        if ((src >> 4) != cs_want)                           // src_plane bits 4..7: k-steps of context in GEMM 1
            return fail(TFK_EINVAL, "%s: op %d: %d context k-steps but the call carries a context of %d elements", fn, i, src >> 4, Cn);
        src &= 15;
        if (src & 4) {                                       // odd event sizes: this layer first takes the middle element over
            if (made || context || prog.n_layers >= 64)
                return fail(TFK_EINVAL, "%s: op %d: only plain spline couplings (no MADE layers, no context) move a middle element", fn, i);
            prog.move_mask |= 1ull << prog.n_layers;
            src &= 3;
        }
        if (k != TFK_OP_RQS_FWD_LEAN && k != TFK_OP_RQS_INV_LEAN && k != TFK_OP_LRS_FWD_LEAN && k != TFK_OP_LRS_INV_LEAN &&
            !(k >= TFK_OP_MADE_RQS_FWD_LEAN && k <= TFK_OP_MADE_LRS_INV_LEAN))
            return fail(TFK_EINVAL, "%s: op %d: kind %d cannot be mixed with lean spline ops", fn, i, k);
        float bnd, sc, cc;
        memcpy(&bnd, rec + 5, 4);
        memcpy(&sc, rec + 6, 4);
        memcpy(&cc, rec + 7, 4);
        if ((rec[4] & 255) != 8) return fail(TFK_EINVAL, "%s: op %d: lean spline ops support n_bins = 8, got %d", fn, i, rec[4] & 255);
        if (((rec[4] >> 8) == 1) != fmt3 || (rec[4] >> 8) > 1)
            return fail(TFK_EINVAL, "%s: op %d: the ops of a lean spline program share one operand format", fn, i);
        if (st < 1 || st > (fmt3 ? 8 : 4))
            return fail(TFK_EINVAL, "%s: op %d: GEMM-2 steps %d not in [1, %d]", fn, i, st, fmt3 ? 8 : 4);
        if (src != 0 && src != 1) return fail(TFK_EINVAL, "%s: op %d: src_plane %d", fn, i, src);
        if (!(bnd > 0.0f)) return fail(TFK_EINVAL, "%s: op %d: boundary must be positive", fn, i);
        if (off < 0 || (off & 3) || off + block > n_params)
            return fail(TFK_EINVAL, "%s: op %d: parameters [%d, %lld) outside the block of %lld floats", fn, i, off,
                        (long long)(off + block), (long long)n_params);
        if (prog.n_layers == 0) {
            kind = k; steps2 = st; boundary = bnd; scale = sc; c = cc;
            prog.first_src = src;
            prog.offset0 = off;
        } else {
            if (k != kind || st != steps2 || bnd != boundary || sc != scale || cc != c)
                return fail(TFK_EINVAL, "%s: op %d: a lean spline program holds couplings of one direction, hidden width and box", fn, i);
            if (!made && src != ((prog.first_src + prog.n_layers) & 1))
                return fail(TFK_EINVAL, "%s: op %d: the source plane of lean couplings must alternate", fn, i);
            const int64_t stride = (int64_t)off - (prog.offset0 + (int64_t)(prog.n_layers - 1) * prog.layer_stride);
            if (prog.n_layers == 1) prog.layer_stride = (int)stride;
            else if (stride != prog.layer_stride)
                return fail(TFK_EINVAL, "%s: op %d: lean spline blocks must lie at a constant stride", fn, i);
        }
        ++prog.n_layers;
    }
    if (prog.n_layers == 0) return fail(TFK_EINVAL, "%s: a lean spline program needs at least one coupling", fn);
    const double span = 2.0 * (double)boundary;
    prog.C.minimum = -boundary;
    prog.C.maximum = boundary;
    prog.C.g = (float)(span * (double)scale);
    prog.C.cmin = (float)(span * (lrs ? 1e-2 : 1e-3));
    for (int j = 1; j < 8; ++j) prog.C.knot_c[j - 1] = (float)(-(double)boundary + j * span * (lrs ? 1e-2 : 1e-3));
    prog.C.d_edge = lrs ? (float)((double)c * 1.4426950408889634)
                        : (float)(((double)c + (double)c / 1000.0) * 1.4426950408889634);
    const int inverse = kind == TFK_OP_RQS_INV_LEAN || kind == TFK_OP_LRS_INV_LEAN || kind == TFK_OP_MADE_RQS_INV_LEAN ||
                        kind == TFK_OP_MADE_LRS_INV_LEAN;
    prog.made = made ? 1 : 0;
    if (fmt3) steps2 = (ht3 == 2 ? 8 : 0) + (lrs ? 16 : 0);
    prog.ctx_steps = cs_want;
    if (EPL == 4) return flow_rqs_chain_launch_4(x, z, logdet, loc, log_scale, logprob, N, params, prog, inverse, steps2, flags, xw, s, fn, context, Cn);
    if (EPL == 8) return flow_rqs_chain_launch_8(x, z, logdet, loc, log_scale, logprob, N, params, prog, inverse, steps2, flags, xw, s, fn, context, Cn);
    if (EPL == 32) return flow_rqs_chain_launch_32(x, z, logdet, loc, log_scale, logprob, N, params, prog, inverse, steps2, flags, xw, s, fn, context, Cn);
    return flow_rqs_chain_launch_16(x, z, logdet, loc, log_scale, logprob, N, params, prog, inverse, steps2, flags, xw, s, fn, context, Cn);
}

// A program made of lean ops only (TFK_OP_*_LEAN couplings of one kind and one GEMM-2 step count whose source
// plane alternates, optionally ended by one TFK_OP_EW_FMA) runs on the straight-line kernel of tfk_flow_chain.h.
static int run_chain(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                     float *logprob, int64_t N, int32_t D, const int32_t *ops, int32_t n_ops, const float *params,
                     int64_t n_params, int32_t flags, int32_t xw, hipStream_t s, const char *fn,
                     double *sum_ws, double *sum_out, const float *context = nullptr, int32_t Cn = 0)
{
    const int EPL = D / 8, HALF = D / 2;
    const int cs_want = context ? (Cn + 3) / 4 : 0;
    ChainProg prog;
    prog.n_c = 0;
    prog.first_src = 0;
    prog.ew_offset = -1;
    prog.pad = 0;
    prog.sum_ws = sum_ws;
    prog.sum_out = sum_out;
    prog.context = context;
    prog.ctx_n = Cn;
    prog.ctx_steps = cs_want;
    prog.move_mask = 0;
    for (int i = 0; i < kChainSideOps; ++i) prog.pre_kind[i] = prog.post_kind[i] = prog.pre_off[i] = prog.post_off[i] = 0;
    int phase = 0;                                           // 0 before the couplings, 1 in them, 2 behind the closing EW_FMA
    int kind = -1, steps2 = 1;
    bool fmt3 = false;
    for (int i = 0; i < n_ops; ++i) {
        const int32_t *rec = ops + 8 * i;
        const int k = rec[0], st = rec[2], off = rec[3];
        int src = rec[1];
        int64_t need;
        const bool ewc = (k == TFK_OP_EWC_MULADD || k == TFK_OP_EWC_SUBDIV);
        if (k != TFK_OP_EW_FMA && (src >> 4) != (((k >= TFK_OP_AFFINE_FWD_LEAN && k <= TFK_OP_SHIFT_INV_LEAN) || ewc) ? cs_want : 0))
            return fail(TFK_EINVAL, "%s: op %d: %d context k-steps but the call carries a context of %d elements", fn, i, src >> 4, Cn);
        src &= 15;
        const bool move_mid = (src & 4) != 0;                // odd event sizes: take the middle element over first
        src &= 3;
        if (move_mid && !(k >= TFK_OP_AFFINE_FWD_LEAN && k <= TFK_OP_SHIFT_INV_LEAN))
            return fail(TFK_EINVAL, "%s: op %d: only affine / shift couplings move a middle element", fn, i);
        if (k == TFK_OP_EW_FMA || ewc) {
            // plain programs: one TFK_OP_EW_FMA, last.  Context programs: elementwise ops (constant or context-conditioned)
            // may also stand in front of the couplings and behind the closing TFK_OP_EW_FMA
            if (ewc && !context) return fail(TFK_EINVAL, "%s: op %d: a context-conditioned elementwise op needs a context", fn, i);
            need = ewc ? (int64_t)EPL * cs_want * 64 + (int64_t)EPL * 16 : 2 * (int64_t)D + 4;
            const int sk = ewc ? (k == TFK_OP_EWC_MULADD ? 2 : 3) : 1;
            if (phase == 1 && !ewc) { prog.ew_offset = off; phase = 2; }
            else if (phase == 1) return fail(TFK_EINVAL, "%s: op %d: a TFK_OP_EW_FMA closes the couplings before this op", fn, i);
            else if (!context) {
                if (i != n_ops - 1) return fail(TFK_EINVAL, "%s: op %d: TFK_OP_EW_FMA must end a lean program", fn, i);
                prog.ew_offset = off;
            } else {
                int *kinds = phase == 0 ? prog.pre_kind : prog.post_kind, *offs = phase == 0 ? prog.pre_off : prog.post_off;
                int n = 0;
                while (n < kChainSideOps && kinds[n]) ++n;
                if (n == kChainSideOps) return fail(TFK_EINVAL, "%s: op %d: more than %d elementwise ops on one side of the couplings", fn, i, kChainSideOps);
                kinds[n] = sk;
                offs[n] = off;
            }
        } else if (k == TFK_OP_MADE_FWD_LEAN || k == TFK_OP_MADE_INV_LEAN) {
            if (context) return fail(TFK_EINVAL, "%s: op %d: lean MADE layers take no context", fn, i);
            phase = 1;
            if (prog.n_c == kMaxChainOps) return fail(TFK_EINVAL, "%s: more than %d lean ops", fn, kMaxChainOps);
            if (st < 1 || st > 4) return fail(TFK_EINVAL, "%s: op %d: GEMM-2 steps %d not in [1, 4] (hidden width <= 16)", fn, i, st);
            const int mk = 4 + (k - TFK_OP_MADE_FWD_LEAN);
            if (prog.n_c == 0) { kind = mk; steps2 = st; }
            else if (mk != kind || st != steps2)
                return fail(TFK_EINVAL, "%s: op %d: a lean program holds layers of one kind and one hidden width", fn, i);
            const int nA2 = (EPL * st + 3) & ~3;
            need = (int64_t)2 * EPL * 64 + 16 + (int64_t)nA2 * 64 + (int64_t)EPL * 16 + 2 * (int64_t)D;
            prog.offset[prog.n_c++] = off;
        } else if (k >= TFK_OP_AFFINE_FWD_LEAN && k <= TFK_OP_SHIFT_INV_LEAN) {
            if (phase == 2) return fail(TFK_EINVAL, "%s: op %d: a coupling behind the closing TFK_OP_EW_FMA", fn, i);
            phase = 1;
            if (kind >= 4) return fail(TFK_EINVAL, "%s: op %d: couplings and MADE layers cannot share a lean program", fn, i);
            if (prog.n_c == kMaxChainOps) return fail(TFK_EINVAL, "%s: more than %d lean couplings", fn, kMaxChainOps);
            if (src != 0 && src != 1) return fail(TFK_EINVAL, "%s: op %d: src_plane %d", fn, i, src);
            if (st < 1 || st > 4) return fail(TFK_EINVAL, "%s: op %d: GEMM-2 steps %d not in [1, 4] (hidden width <= 16)", fn, i, st);
            if (prog.n_c == 0) {
                kind = k - TFK_OP_AFFINE_FWD_LEAN;
                steps2 = st;
                prog.first_src = src;
            } else {
                if (k - TFK_OP_AFFINE_FWD_LEAN != kind || st != steps2)
                    return fail(TFK_EINVAL, "%s: op %d: a lean program holds couplings of one kind and one hidden width", fn, i);
                if (src != ((prog.first_src + prog.n_c) & 1))
                    return fail(TFK_EINVAL, "%s: op %d: the source plane of lean couplings must alternate", fn, i);
            }
            const int T2 = (kind < 2) ? EPL / 2 : (EPL + 3) / 4;
            const int nA2 = (T2 * st + 3) & ~3;
            const bool f3 = rec[4] == 256;               // K field: 256 = bf16 x 3 operands (no b2, A23[T2][2][64][4])
            if (prog.n_c == 0) fmt3 = f3;
            else if (f3 != fmt3) return fail(TFK_EINVAL, "%s: op %d: the ops of a lean program share one operand format", fn, i);
            if (f3 && context) return fail(TFK_EINVAL, "%s: op %d: context-conditioned lean couplings use fp32 operands", fn, i);
            need = f3 ? (int64_t)EPL * 64 + 16 + (int64_t)T2 * 2 * 64 * 4 + 2 * HALF
                      : (int64_t)EPL * 64 + 16 + (int64_t)nA2 * 64 + (int64_t)T2 * 16 + 2 * HALF + (cs_want ? 256 : 0);
            if (move_mid) {
                if (context) return fail(TFK_EINVAL, "%s: op %d: context programs do not move a middle element", fn, i);
                prog.move_mask |= 1ull << prog.n_c;
            }
            prog.offset[prog.n_c++] = off;
        } else {
            return fail(TFK_EINVAL, "%s: op %d: kind %d cannot be mixed with lean ops", fn, i, k);
        }
        if (off < 0 || (off & 3) || off + need > n_params)
            return fail(TFK_EINVAL, "%s: op %d: parameters [%d, %lld) outside the block of %lld floats", fn, i,
                        off, (long long)(off + need), (long long)n_params);
    }
    if (kind < 0) kind = 2;
    if (fmt3) steps2 = 0;
    if (EPL == 2) return flow_chain_launch_2(x, z, logdet, loc, log_scale, logprob, N, params, (int)n_params, prog, kind, steps2, flags, xw, s, fn);
    if (EPL == 4) return flow_chain_launch_4(x, z, logdet, loc, log_scale, logprob, N, params, (int)n_params, prog, kind, steps2, flags, xw, s, fn);
    if (EPL == 8) return flow_chain_launch_8(x, z, logdet, loc, log_scale, logprob, N, params, (int)n_params, prog, kind, steps2, flags, xw, s, fn);
    if (EPL == 32) return flow_chain_launch_32(x, z, logdet, loc, log_scale, logprob, N, params, (int)n_params, prog, kind, steps2, flags, xw, s, fn);
    return flow_chain_launch_16(x, z, logdet, loc, log_scale, logprob, N, params, (int)n_params, prog, kind, steps2, flags, xw, s, fn);
}

int flow_mfma_launch_8(const float *, float *, float *, const float *, const float *, float *, int64_t,
                       const float *, int, const MProgram &, int, hipStream_t, const char *, const float *, int);
int flow_mfma_launch_16(const float *, float *, float *, const float *, const float *, float *, int64_t,
                        const float *, int, const MProgram &, int, hipStream_t, const char *, const float *, int);
int flow_mfma_launch_32(const float *, float *, float *, const float *, const float *, float *, int64_t,
                        const float *, int, const MProgram &, int, hipStream_t, const char *, const float *, int);
}  // namespace tfk

using namespace tfk;

extern "C" {

int tfk_flow_mfma_supported(int32_t D) { return (D == 64 || D == 128 || D == 256) ? 1 : 0; }

// (D = 16: chains of affine / shift couplings only -- fp32 operands, no context, no MADE / spline ops)
int tfk_flow_lean_supported(int32_t D) { return (D == 16 || D == 32 || tfk_flow_mfma_supported(D)) ? 1 : 0; }

static int flow_run_mfma_impl(const float *x, int32_t x_width, float *z, float *logdet, const float *gauss_loc,
                              const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                              const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                              int32_t accumulate, void *stream, const char *fn,
                              const float *context = nullptr, int32_t C = 0,
                              double *sum_ws = nullptr, double *sum_out = nullptr)
{
    if (context && (C < 1 || C > 4 * kCtxSteps))
        return fail(TFK_EINVAL, "%s: context size %d must be in [1, %d]", fn, C, 4 * kCtxSteps);
    const int cs_want = context ? (C + 3) / 4 : 0;
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    // a LEAN program: its first coupling / MADE op is of a lean kind (elementwise ops may stand in front of it in a context
    // program), or it consists of TFK_OP_EW_FMA alone
    int first_kind = -1;
    for (int i = 0; ops && i < n_ops && first_kind < 0; ++i) {
        const int k = ops[8 * i];
        if (k != TFK_OP_EW_FMA && k != TFK_OP_EWC_MULADD && k != TFK_OP_EWC_SUBDIV && k != TFK_OP_EW_MULADD &&
            k != TFK_OP_EW_SUBDIV && k != TFK_OP_PLANE_SWAP)
            first_kind = k;
    }
    const bool lean = n_ops > 0 && ops &&
                      (first_kind < 0 ? ops[0] == TFK_OP_EW_FMA
                                      : ((first_kind >= TFK_OP_AFFINE_FWD_LEAN && first_kind <= TFK_OP_RQS_INV_LEAN &&
                                          first_kind != TFK_OP_EW_FMA) ||
                                         (first_kind >= TFK_OP_MADE_FWD_LEAN && first_kind <= TFK_OP_MADE_LRS_INV_LEAN)));
    if (!(lean ? tfk_flow_lean_supported(D) : tfk_flow_mfma_supported(D)))
        return fail(TFK_EINVAL, "%s: D = %d must be 64, 128 or 256 (lean programs: 32 as well, affine / shift chains: 16)", fn, D);
    if (D == 16 && (context || (first_kind >= 0 && !(first_kind >= TFK_OP_AFFINE_FWD_LEAN && first_kind <= TFK_OP_SHIFT_INV_LEAN))))
        return fail(TFK_EINVAL, "%s: D = 16: chains of affine / shift couplings without a context only", fn);
    if (n_ops < 0 || n_ops > kMaxOpsM) return fail(TFK_EINVAL, "%s: n_ops = %d must be in [0, %d]", fn, n_ops, kMaxOpsM);
    if (n_params < 0 || (n_params & 3)) return fail(TFK_EINVAL, "%s: n_params must be a non-negative multiple of 4", fn);
    if (N == 0) return TFK_OK;
    if (!x || (n_ops > 0 && (!ops || !params))) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!z && !logdet && !logprob) return fail(TFK_EINVAL, "%s: no output requested", fn);
    if (logprob && (!gauss_loc || !gauss_log_scale)) return fail(TFK_EINVAL, "%s: logprob needs the base parameters", fn);
    if ((x_width == D && !aligned16(x)) || (z && !aligned16(z)) || !aligned16(params))
        return fail(TFK_EINVAL, "%s: x, z and params must be 16-byte aligned", fn);
    const bool lean_spline = lean && (first_kind == TFK_OP_RQS_FWD_LEAN || first_kind == TFK_OP_RQS_INV_LEAN ||
                                      first_kind == TFK_OP_LRS_FWD_LEAN || first_kind == TFK_OP_LRS_INV_LEAN ||
                                      (first_kind >= TFK_OP_MADE_RQS_FWD_LEAN && first_kind <= TFK_OP_MADE_LRS_INV_LEAN));
    if (lean && context && !lean_spline && !(first_kind >= TFK_OP_AFFINE_FWD_LEAN && first_kind <= TFK_OP_SHIFT_INV_LEAN))
        return fail(TFK_EINVAL, "%s: of the lean programs only coupling chains take a context", fn);
    if (sum_ws && (!lean || !logprob || !sum_out))
        return fail(TFK_EINVAL, "%s: the in-kernel sum needs a lean program, logprob and sum_out", fn);
    // (an odd width: affine / shift chains whose couplings move the middle element -- it is read into plane B's last column)
    const bool odd_ok = (x_width & 1) && x_width >= 3 && (x_width + 1) / 2 <= D / 2 &&
                        ((first_kind >= TFK_OP_AFFINE_FWD_LEAN && first_kind <= TFK_OP_SHIFT_INV_LEAN) ||
                         first_kind == TFK_OP_RQS_FWD_LEAN || first_kind == TFK_OP_RQS_INV_LEAN ||
                         first_kind == TFK_OP_LRS_FWD_LEAN || first_kind == TFK_OP_LRS_INV_LEAN ||
                         first_kind == TFK_OP_MADE_FWD_LEAN || first_kind == TFK_OP_MADE_INV_LEAN);
    if (x_width != D && (!lean || x_width < 2 || x_width > D || ((x_width & 1) && !odd_ok)))
        return fail(TFK_EINVAL, "%s: x_width = %d: narrower input rows need a lean program and an even width <= D = %d "
                    "(odd: a chain of couplings, (x_width + 1) / 2 <= D / 2)", fn, x_width, D);
    if (lean_spline)
        return run_rqs_chain(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, D, ops, n_ops, params, n_params,
                             accumulate, x_width, static_cast<hipStream_t>(stream), fn, sum_ws, sum_out, context, C);
    if (lean)
        return run_chain(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, D, ops, n_ops, params, n_params,
                         accumulate, x_width, static_cast<hipStream_t>(stream), fn, sum_ws, sum_out, context, C);
    const int EPL = D / 8;
    MProgram prog;
    prog.n_ops = n_ops;
    for (int i = 0; i < n_ops; ++i) {
        MOp &o = prog.op[i];
        const int32_t *rec = ops + 8 * i;
        o.kind = rec[0];
        o.src_plane = rec[1];
        o.steps2 = rec[2];
        o.offset = rec[3];
        o.K = rec[4];
        memcpy(&o.boundary, rec + 5, 4);
        memcpy(&o.scale, rec + 6, 4);
        memcpy(&o.c, rec + 7, 4);
        int64_t need;
        const int cs = o.src_plane >> 4;                 // k-steps of context in GEMM 1 (couplings) / in the one GEMM (EWC)
        if (cs != 0 && (cs != cs_want || !context))
            return fail(TFK_EINVAL, "%s: op %d: %d context k-steps but the call carries a context of %d elements", fn, i, cs, C);
        if (o.kind == TFK_OP_EWC_MULADD || o.kind == TFK_OP_EWC_SUBDIV) {
            if (!context || cs < 1) return fail(TFK_EINVAL, "%s: op %d: a context-conditioned elementwise op needs a context", fn, i);
            need = (int64_t)EPL * cs * 64 + (int64_t)EPL * 16;
        } else if (o.kind == TFK_OP_PLANE_SWAP) {
            if (D > 128) return fail(TFK_EINVAL, "%s: op %d: plane swaps exist for D <= 128", fn, i);
            need = D / 2;
        }
        else if (o.kind == TFK_OP_EW_MULADD) need = 2 * (int64_t)D + 4;
        else if (o.kind == TFK_OP_EW_SUBDIV) need = 3 * (int64_t)D + 4;
        else if (o.kind >= TFK_OP_AFFINE_FWD && o.kind <= TFK_OP_SHIFT_INV) {
            if (o.steps2 < 1 || o.steps2 > 32) return fail(TFK_EINVAL, "%s: op %d: GEMM-2 steps %d not in [1, 32] (hidden width <= 128)", fn, i, o.steps2);
            if ((o.src_plane & ~0xf1) != 0) return fail(TFK_EINVAL, "%s: op %d: src_plane %d", fn, i, o.src_plane);
            const int T2 = (o.kind <= TFK_OP_AFFINE_INV) ? EPL / 2 : EPL / 4;
            const int HT = o.steps2 <= 4 ? 1 : (o.steps2 <= 8 ? 2 : (o.steps2 <= 16 ? 4 : 8));
            if (cs && HT != 1) return fail(TFK_EINVAL, "%s: op %d: context-conditioned couplings need hidden width <= 16", fn, i);
            need = (int64_t)EPL * HT * 64 + HT * 16 + (int64_t)T2 * o.steps2 * 64 + (int64_t)T2 * 16 + (int64_t)cs * HT * 64;
        } else if (o.kind == TFK_OP_RQS_FWD || o.kind == TFK_OP_RQS_INV) {
            if (o.steps2 < 1 || o.steps2 > 4) return fail(TFK_EINVAL, "%s: op %d: GEMM-2 steps %d not in [1, 4] (hidden width <= 16)", fn, i, o.steps2);
            if ((o.src_plane & ~0xf1) != 0) return fail(TFK_EINVAL, "%s: op %d: src_plane %d", fn, i, o.src_plane);
            if (EPL > kMaxEplRqs) return fail(TFK_EINVAL, "%s: op %d: RQS couplings need D <= %d on the MFMA path", fn, i, 8 * kMaxEplRqs);
            if (o.K != 8) return fail(TFK_EINVAL, "%s: op %d: fused RQS supports n_bins = 8, got %d", fn, i, o.K);
            if (!(o.boundary > 0.0f)) return fail(TFK_EINVAL, "%s: op %d: boundary must be positive", fn, i);
            const int T2 = EPL * 6;
            need = (int64_t)EPL * 64 + 16 + (int64_t)T2 * o.steps2 * 64 + (int64_t)T2 * 16 + (int64_t)cs * 64;
        } else if (o.kind == TFK_OP_MADE_FWD || o.kind == TFK_OP_MADE_INV) {
            if (o.steps2 < 1 || o.steps2 > 32) return fail(TFK_EINVAL, "%s: op %d: GEMM-2 steps %d not in [1, 32] (hidden width <= 128)", fn, i, o.steps2);
            const int HT = o.steps2 <= 4 ? 1 : (o.steps2 <= 8 ? 2 : (o.steps2 <= 16 ? 4 : 8));
            need = (int64_t)2 * EPL * HT * 64 + HT * 16 + (int64_t)EPL * o.steps2 * 64 + (int64_t)EPL * 16;
        } else if (o.kind == TFK_OP_MADE_RQS) {
            if (o.steps2 < 1 || o.steps2 > 4) return fail(TFK_EINVAL, "%s: op %d: GEMM-2 steps %d not in [1, 4] (hidden width <= 16)", fn, i, o.steps2);
            if (EPL > kMaxEplRqs) return fail(TFK_EINVAL, "%s: op %d: spline ops need D <= %d on the MFMA path", fn, i, 8 * kMaxEplRqs);
            if (o.K != 8) return fail(TFK_EINVAL, "%s: op %d: fused RQS supports n_bins = 8, got %d", fn, i, o.K);
            if (!(o.boundary > 0.0f)) return fail(TFK_EINVAL, "%s: op %d: boundary must be positive", fn, i);
            const int T2 = EPL * 6;
            need = (int64_t)2 * EPL * 64 + 16 + (int64_t)2 * T2 * o.steps2 * 64 + (int64_t)2 * T2 * 16;
        } else return fail(TFK_EINVAL, "%s: op %d: kind %d is not supported on the MFMA path", fn, i, o.kind);
        if (o.offset < 0 || (o.offset & 3) || o.offset + need > n_params)
            return fail(TFK_EINVAL, "%s: op %d: parameters [%d, %lld) outside the block of %lld floats", fn, i,
                        o.offset, (long long)(o.offset + need), (long long)n_params);
    }
    if (context)
        for (int i = 0; i < n_ops; ++i) {
            const int k = prog.op[i].kind;
            if (k >= TFK_OP_MADE_FWD && k <= TFK_OP_MADE_RQS)
                return fail(TFK_EINVAL, "%s: op %d: MADE ops take no context", fn, i);
            if (k >= TFK_OP_AFFINE_FWD && k <= TFK_OP_SHIFT_INV && prog.op[i].steps2 > 4)
                return fail(TFK_EINVAL, "%s: op %d: context-conditioned programs need hidden width <= 16", fn, i);
        }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (EPL == 8)
        return flow_mfma_launch_8(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, (int)n_params, prog, accumulate, s, fn, context, C);
    if (EPL == 32)
        return flow_mfma_launch_32(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, (int)n_params, prog, accumulate, s, fn, context, C);
    return flow_mfma_launch_16(x, z, logdet, gauss_loc, gauss_log_scale, logprob, N, params, (int)n_params, prog, accumulate, s, fn, context, C);
}

int tfk_flow_run_mfma(const float *x, float *z, float *logdet, const float *gauss_loc,
                      const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                      const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                      int32_t accumulate, void *stream)
{
    return flow_run_mfma_impl(x, D, z, logdet, gauss_loc, gauss_log_scale, logprob, N, D, ops, n_ops, params, n_params,
                              accumulate, stream, "tfk_flow_run_mfma");
}

int tfk_flow_run_mfma_in(const float *x, int32_t x_width, float *z, float *logdet, const float *gauss_loc,
                         const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                         const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                         int32_t accumulate, void *stream)
{
    return flow_run_mfma_impl(x, x_width, z, logdet, gauss_loc, gauss_log_scale, logprob, N, D, ops, n_ops, params,
                              n_params, accumulate, stream, "tfk_flow_run_mfma_in");
}

int64_t tfk_flow_sum_workspace_bytes(void) { return (int64_t)(1 + cu_count() * 8 * kGridOversubscribe) * (int64_t)sizeof(double); }

int tfk_flow_run_mfma_sum(const float *x, int32_t x_width, float *z, float *logdet, const float *gauss_loc,
                          const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                          const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                          int32_t accumulate, void *sum_workspace, double *sum_out, void *stream)
{
    const char *fn = "tfk_flow_run_mfma_sum";
    if (!sum_workspace || !sum_out) return fail(TFK_EINVAL, "%s: null sum_workspace / sum_out", fn);
    if (N <= 0) return fail(TFK_EINVAL, "%s: N = %lld: the in-kernel sum needs at least one row", fn, (long long)N);
    return flow_run_mfma_impl(x, x_width, z, logdet, gauss_loc, gauss_log_scale, logprob, N, D, ops, n_ops, params,
                              n_params, accumulate, stream, fn, nullptr, 0, static_cast<double *>(sum_workspace), sum_out);
}

int tfk_flow_run_mfma_ctx(const float *x, const float *context, int32_t C, float *z, float *logdet,
                          const float *gauss_loc, const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                          const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                          int32_t accumulate, void *stream)
{
    const char *fn = "tfk_flow_run_mfma_ctx";
    if (!context) return fail(TFK_EINVAL, "%s: null context", fn);
    return flow_run_mfma_impl(x, D, z, logdet, gauss_loc, gauss_log_scale, logprob, N, D, ops, n_ops, params, n_params,
                              accumulate, stream, fn, context, C);
}

}  // extern "C"
