"""fp64 emulator of the LEAN flow programs (csrc/tfk_flow_chain.h) for host-side tests: decodes the packed
parameter blocks exactly as the kernel reads them (lane-major MFMA A-operands, accumulator layouts, pre-affine,
base-2 log-det) so that the packer in torchflows_amd/fused.py can be checked without a GPU."""
import math

import torch

LN2 = math.log(2.0)


def _mfma(A_lane, B_lane, acc):
    """v_mfma_f32_16x16x4_f32 on 'lane' tensors: A_lane (64,) = A[i = l & 15][k = l >> 4], B_lane (64, W) batched over
    W waves = B[k = l >> 4][j = l & 15], acc (64, 4, W): reg r of lane (q, j) = D[4 q + r][j]."""
    lane = torch.arange(64)
    A = torch.zeros(16, 4, dtype=torch.float64)
    A[lane & 15, lane >> 4] = A_lane
    Bm = torch.zeros(4, 16, B_lane.shape[1], dtype=torch.float64)
    Bm[lane >> 4, lane & 15] = B_lane
    Dm = torch.einsum("ik,kjw->ijw", A, Bm)                   # (16, 16, W)
    q, j = lane >> 4, lane & 15
    out = acc.clone()
    for r in range(4):
        out[:, r] = acc[:, r] + Dm[4 * q + r, j]
    return out


def rqs_lean_eval(p, v, C, inverse):
    """The spline element of csrc/tfk_flow_rqs_chain.h in fp64: p (..., 24) pre-scaled parameters, v (...) inputs.
    Returns (out, log2-det)."""
    minimum, maximum, g, cmin, d_edge = C
    ux, uy, ud = p[..., 0:8], p[..., 8:16], p[..., 16:23]
    ex = torch.exp2(ux - ux.max(-1, keepdim=True).values)
    ey = torch.exp2(uy - uy.max(-1, keepdim=True).values)
    gx, gy = g / ex.sum(-1, keepdim=True), g / ey.sum(-1, keepdim=True)
    jj = torch.arange(1, 8, dtype=torch.float64)
    X = torch.cat([torch.full_like(v[..., None], minimum), torch.cumsum(ex * gx, -1)[..., :7] + minimum + jj * cmin,
                   torch.full_like(v[..., None], maximum)], -1)
    Y = torch.cat([torch.full_like(v[..., None], minimum), torch.cumsum(ey * gy, -1)[..., :7] + minimum + jj * cmin,
                   torch.full_like(v[..., None], maximum)], -1)
    Dl = torch.cat([torch.full_like(v[..., None], d_edge), ud, torch.full_like(v[..., None], d_edge)], -1)
    S = Y if inverse else X
    k = ((S[..., 1:8] < v[..., None]).sum(-1, keepdim=True)).clamp(0, 7)
    pick = lambda t, i: torch.gather(t, -1, i).squeeze(-1)
    bxk, bxk1, byk, byk1 = pick(X, k), pick(X, k + 1), pick(Y, k), pick(Y, k + 1)
    dk = 1e-5 + LN2 * torch.log2(1 + torch.exp2(pick(Dl, k)))
    dk1 = 1e-5 + LN2 * torch.log2(1 + torch.exp2(pick(Dl, k + 1)))
    wk, hk = bxk1 - bxk, byk1 - byk
    s = hk / wk
    term1 = dk1 + dk - 2 * s
    if not inverse:
        xi = ((v - bxk) / wk).clamp(0, 1)
    else:
        term0 = v - byk
        term2 = hk * dk
        a = (hk * s - term2) + term0 * term1
        b = term2 - term0 * term1
        c = -s * term0
        r = torch.sqrt((b * b - 4 * a * c).clamp(min=0))
        xi = (2 * c / (-b - r)).clamp(0, 1)
    q = xi * (1 - xi)
    den = s + term1 * q
    out = byk + hk * (s * xi * xi + dk * q) / den if not inverse else xi * wk + bxk
    inner = dk1 * xi * xi + 2 * s * q + dk * (1 - xi) ** 2
    l2 = torch.log2((s / den) ** 2 * inner)
    inb = (v > minimum) & (v < maximum)
    return torch.where(inb, out, v), torch.where(inb, -l2 if inverse else l2, torch.zeros_like(l2))


def lrs_lean_eval(p, v, C, inverse):
    """The linear-rational-spline element of csrc/tfk_flow_rqs_chain.h (lrs_eval_lean) in fp64: p (..., 32)."""
    minimum, maximum, g, cmin, d_edge = C
    ux, uy, nl, ud, uw = p[..., 0:8], p[..., 8:16], p[..., 16:24], p[..., 24:31], p[..., 31]
    ex = torch.exp2(ux - ux.max(-1, keepdim=True).values)
    ey = torch.exp2(uy - uy.max(-1, keepdim=True).values)
    gx, gy = g / ex.sum(-1, keepdim=True), g / ey.sum(-1, keepdim=True)
    jj = torch.arange(1, 8, dtype=torch.float64)
    edge = lambda val: torch.full_like(v[..., None], val)
    X = torch.cat([edge(minimum), torch.cumsum(ex * gx, -1)[..., :7] + minimum + jj * cmin, edge(maximum)], -1)
    Y = torch.cat([edge(minimum), torch.cumsum(ey * gy, -1)[..., :7] + minimum + jj * cmin, edge(maximum)], -1)
    Dl = torch.cat([edge(d_edge), ud, edge(d_edge)], -1)
    S = Y if inverse else X
    k = ((S[..., 1:8] < v[..., None]).sum(-1, keepdim=True)).clamp(0, 7)
    pick = lambda t, i: torch.gather(t, -1, i).squeeze(-1)
    xk, xk1, yk, yk1 = pick(X, k), pick(X, k + 1), pick(Y, k), pick(Y, k + 1)
    dk = 1e-5 + LN2 * torch.log2(1 + torch.exp2(pick(Dl, k)))
    dk1 = 1e-5 + LN2 * torch.log2(1 + torch.exp2(pick(Dl, k + 1)))
    lam = 1.0 / (1.0 + torch.exp2(pick(nl, k)))
    w0 = LN2 * torch.log2(1 + torch.exp2(uw))
    wk, wk1 = w0 / torch.sqrt(dk), w0 / torch.sqrt(dk1)
    ym = ((1 - lam) * wk * yk + lam * wk1 * yk1) / ((1 - lam) * wk + lam * wk1)
    dx = xk1 - xk
    wm = (lam * wk * dk + (1 - lam) * wk1 * dk1) * (dx / (yk1 - yk))
    n_lo, n_hi = lam * wk * wm * (ym - yk), (1 - lam) * wm * wk1 * (yk1 - ym)
    if not inverse:
        phi = (v - xk) / dx
        up = phi > lam
        den = torch.where(up, wm * (1 - phi) + wk1 * (phi - lam), wk * (lam - phi) + wm * phi)
        num = torch.where(up, wm * ym * (1 - phi) + wk1 * yk1 * (phi - lam), wk * yk * (lam - phi) + wm * ym * phi)
        out = num / den
        l2 = torch.log2(torch.where(up, n_hi, n_lo) / ((den * den + 5e-10) * dx))
    else:
        up = v > ym
        den = torch.where(up, wk1 * (yk1 - v), wk * (yk - v)) + wm * (v - ym)
        num = torch.where(up, lam * wk1 * (yk1 - v) + wm * (v - ym), lam * wk * (yk - v))
        out = num / den * dx + xk
        l2 = torch.log2(torch.where(up, n_hi, n_lo) * dx / (den * den + 5e-10))
    inb = (v > minimum) & (v < maximum)
    return torch.where(inb, out, v), torch.where(inb, l2, torch.zeros_like(l2))


def _pieces(v32):
    """fp32 tensor -> (hi, mid, lo) bf16 pieces by truncation, as fp64."""
    bits = v32.contiguous().view(torch.int32)
    hi = (bits & -65536).view(torch.float32)
    r1 = v32 - hi
    mid = (r1.contiguous().view(torch.int32) & -65536).view(torch.float32)
    r2 = r1 - mid
    lo = (r2.contiguous().view(torch.int32) & -65536).view(torch.float32)
    return hi.double(), mid.double(), lo.double()


def _unpack_bf16(d):
    """(64, 2 dwords) int32 -> (64, 4) fp64 values of the 4 packed bf16."""
    lo16 = ((d & 0xFFFF) << 16).view(torch.float32).double()
    hi16 = (d & -65536).view(torch.float32).double()
    return torch.stack([lo16[:, 0], hi16[:, 0], lo16[:, 1], hi16[:, 1]], dim=1)


def run_lean(ops, params, rows, D, context=None):
    """rows (N, D) fp64, N a multiple of 16 -> (rows out, logdet) of one lean segment.  ``context`` (N, C) fp64 for
    context programs: src_plane bits 4..7 of the couplings = k-steps of context in GEMM 1 (A1c behind the block's pre_t),
    TFK_OP_EWC_* ops (lean format: scale logits pre-scaled for exp2) in front of / behind the couplings."""
    EPL, HALF = D // 8, D // 2
    N = rows.shape[0]
    W = N // 16
    lane = torch.arange(64)
    q, j = lane >> 4, lane & 15
    x = rows.reshape(W, 16, D)
    cx = None
    if context is not None:                                   # cx[s][lane, w] = context element 4 s + q of row j of wave w
        C_ = context.shape[1]
        cpad = torch.zeros(N, 16, dtype=torch.float64)
        cpad[:, :C_] = context
        cw = cpad.reshape(W, 16, 16)
        cx = [cw[:, j, 4 * s_ + q].t().contiguous() for s_ in range(4)]         # each (64, W)
    # lane registers: a[l, e, w] = row j of wave w, element EPL q + e of plane A
    idx = (EPL * q)[:, None] + torch.arange(EPL)[None, :]                      # (64, EPL)
    a = x[:, j][:, torch.arange(64)[:, None], idx].permute(1, 2, 0).clone()   # (64, EPL, W)
    b = x[:, j][:, torch.arange(64)[:, None], HALF + idx].permute(1, 2, 0).clone()
    ld2 = torch.zeros(64, W, dtype=torch.float64)
    ld = torch.zeros(64, W, dtype=torch.float64)
    prm = params.double()
    sign = 0.0
    op_extra = {(op[0], op[3]): op[4:8] for op in ops if len(op) > 4}
    for kind, plane, steps2, off in [op[:4] for op in ops]:
        cs = plane >> 4                                       # k-steps of context (context programs)
        plane = plane & 15
        if kind in (19, 20):                                  # TFK_OP_EWC_* inside a lean program (ewc_lean)
            Ac = prm[off:off + EPL * cs * 64].reshape(EPL, cs, 64)
            bc = prm[off + EPL * cs * 64:off + EPL * cs * 64 + EPL * 16].reshape(EPL, 4, 4)
            part = torch.zeros(64, W, dtype=torch.float64)
            a, b = a.clone(), b.clone()
            for t_ in range(EPL):
                o = bc[t_][q][:, :, None].expand(64, 4, W).clone()
                for s_ in range(cs):
                    o = _mfma(Ac[t_, s_], cx[s_], o)
                pl, tt = (a, t_) if t_ < EPL // 2 else (b, t_ - EPL // 2)
                for i in range(2):
                    al = torch.exp2(o[:, 2 * i]) + 1e-10
                    part = part + torch.log2(al)
                    e = 2 * tt + i
                    pl[:, e] = al * pl[:, e] + o[:, 2 * i + 1] if kind == 19 else (pl[:, e] - o[:, 2 * i + 1]) / al
            ld = ld + (LN2 if kind == 19 else -LN2) * part
            continue
        if kind == 16:                                        # TFK_OP_EW_FMA
            s, t = prm[off:off + D], prm[off + D:off + 2 * D]
            a = s[idx][:, :, None] * a + t[idx][:, :, None]
            b = s[HALF + idx][:, :, None] * b + t[HALF + idx][:, :, None]
            ld[q == 0] += prm[off + 2 * D]
            continue
        if kind in (17, 18, 23, 24):                          # lean RQ-spline / linear-rational-spline coupling
            lrs = kind >= 23
            TPE = 8 if lrs else 6
            K, boundary, scale, cdelta = op_extra[(kind, off)]
            fmt3 = (int(K) >> 8) == 1
            HT = 2 if (fmt3 and steps2 > 4) else 1
            HEAD = EPL * HT * 64 + HT * 16 + 2 * HALF + (HT * 256 if cs else 0)
            A1 = prm[off:off + EPL * HT * 64].reshape(EPL // 4, HT, 64, 4)
            b1 = prm[off + EPL * HT * 64:off + EPL * HT * 64 + HT * 16].reshape(HT, 16)
            pre = prm[off + EPL * HT * 64 + HT * 16:off + EPL * HT * 64 + HT * 16 + 2 * HALF]
            A1c = prm[off + EPL * HT * 64 + HT * 16 + 2 * HALF:off + HEAD].reshape(HT, 64, 4) if cs else None
            src, tgt = (a, b) if plane == 0 else (b, a)
            hids = []
            for th in range(HT):
                acc = b1[th][(4 * q)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
                for s_ in range(EPL):
                    acc = _mfma(A1[s_ // 4, th, :, s_ % 4], src[:, s_], acc)
                for s_ in range(cs):
                    acc = _mfma(A1c[th, :, s_], cx[s_], acc)
                hids.append(1.0 - 2.0 / (torch.exp2(acc) + 1.0))
            tgt = pre[idx][:, :, None] * tgt + pre[HALF + idx][:, :, None]
            hid = hids[0]
            tgt = tgt.clone()
            span = 2.0 * boundary
            C = (-boundary, boundary, span * scale, span * (1e-2 if lrs else 1e-3),
                 (cdelta if lrs else cdelta + cdelta / 1000.0) * 1.4426950408889634)
            if fmt3:                                          # bf16 x 3 operands: chunks of 4 / HT elements
                hids[HT - 1] = hids[HT - 1].clone()
                hids[HT - 1][q == 3, 3] = 1.0                 # the last hidden unit carries the bias
                hp = [_pieces(h.float()) for h in hids]       # per hidden tile: (hi, mid, lo), each (64, 4, W)
                raw = params.contiguous().view(torch.int32)
                A = raw[off + HEAD:off + HEAD + EPL * TPE * HT * 2 * 64 * 4].reshape(EPL, TPE, HT, 2, 64, 4)
                for e in range(EPL):
                    pp = []
                    for c in range(TPE):
                        o = torch.zeros(64, 4, W, dtype=torch.float64)
                        for th in range(HT):
                            w_hi, w_mid = _unpack_bf16(A[e, c, th, 0][:, 0:2]), _unpack_bf16(A[e, c, th, 0][:, 2:4])
                            w_lo = _unpack_bf16(A[e, c, th, 1][:, 0:2])
                            assert torch.equal(w_hi, _unpack_bf16(A[e, c, th, 1][:, 2:4]))
                            h_hi, h_mid, h_lo = hp[th]
                            for Wl, Bl in ((w_hi, h_hi), (w_mid, h_hi), (w_hi, h_mid), (w_mid, h_mid), (w_lo, h_hi), (w_hi, h_lo)):
                                for i in range(4):
                                    o = _mfma(Wl[:, i], Bl[:, i], o)
                        pp.append(o)
                    pvec = torch.cat(pp, 1).permute(0, 2, 1)
                    out, l2 = (lrs_lean_eval(pvec, tgt[:, e], C, kind == 24) if lrs
                               else rqs_lean_eval(pvec, tgt[:, e], C, kind == 18))
                    tgt[:, e] = out
                    ld2 = ld2 + l2
                sign = 1.0
                if plane == 0:
                    b = tgt
                else:
                    a = tgt
                continue
            CH = 48 * 256 + 48 * 16
            for ch in range(EPL // 8):
                base = off + HEAD + ch * CH
                A2 = prm[base:base + 48 * 256].reshape(48, 64, 4)
                b2 = prm[base + 48 * 256:base + CH]
                for e in range(8):
                    pp = []
                    for c in range(6):
                        t_ = e * 6 + c
                        o = b2[((t_ * 4 + q) * 4)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
                        for k in range(steps2):
                            o = _mfma(A2[t_, :, k], hid[:, k], o)
                        pp.append(o)
                    pvec = torch.cat(pp, 1).permute(0, 2, 1)                       # (64, W, 24)
                    out, l2 = rqs_lean_eval(pvec, tgt[:, 8 * ch + e], C, kind == 18)
                    tgt[:, 8 * ch + e] = out
                    ld2 = ld2 + l2
            sign = 1.0
            if plane == 0:
                b = tgt
            else:
                a = tgt
            continue
        if kind in (25, 27):                                  # lean MADE spline layer (rqs_made_layer3), bf16 x 3, HT = 1
            lrs = kind == 27
            TPE = 8 if lrs else 6
            K, boundary, scale, cdelta = op_extra[(kind, off)]
            HEAD = 2 * EPL * 64 + 16 + 2 * D
            A1 = prm[off:off + 2 * EPL * 64].reshape(2 * EPL // 4, 64, 4)
            b1 = prm[off + 2 * EPL * 64:off + 2 * EPL * 64 + 16]
            pre = prm[off + 2 * EPL * 64 + 16:off + HEAD]
            acc = b1[(4 * q)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
            for s_ in range(EPL):
                acc = _mfma(A1[s_ // 4, :, s_ % 4], a[:, s_], acc)
                acc = _mfma(A1[EPL // 4 + s_ // 4, :, s_ % 4], b[:, s_], acc)
            a = (pre[idx][:, :, None] * a + pre[D + idx][:, :, None]).clone()
            b = (pre[HALF + idx][:, :, None] * b + pre[D + HALF + idx][:, :, None]).clone()
            hid = 1.0 - 2.0 / (torch.exp2(acc) + 1.0)
            hid[q == 3, 3] = 1.0
            h_hi, h_mid, h_lo = _pieces(hid.float())
            span = 2.0 * boundary
            C = (-boundary, boundary, span * scale, span * (1e-2 if lrs else 1e-3),
                 (cdelta if lrs else cdelta + cdelta / 1000.0) * 1.4426950408889634)
            raw = params.contiguous().view(torch.int32)
            per_plane = EPL * TPE * 2 * 64 * 4
            for plane_i, tgt in enumerate((a, b)):
                A = raw[off + HEAD + plane_i * per_plane:off + HEAD + (plane_i + 1) * per_plane].reshape(EPL, TPE, 1, 2, 64, 4)
                for e in range(EPL):
                    pp = []
                    for c in range(TPE):
                        o = torch.zeros(64, 4, W, dtype=torch.float64)
                        w_hi, w_mid = _unpack_bf16(A[e, c, 0, 0][:, 0:2]), _unpack_bf16(A[e, c, 0, 0][:, 2:4])
                        w_lo = _unpack_bf16(A[e, c, 0, 1][:, 0:2])
                        for Wl, Bl in ((w_hi, h_hi), (w_mid, h_hi), (w_hi, h_mid), (w_mid, h_mid), (w_lo, h_hi), (w_hi, h_lo)):
                            for i in range(4):
                                o = _mfma(Wl[:, i], Bl[:, i], o)
                        pp.append(o)
                    pvec = torch.cat(pp, 1).permute(0, 2, 1)
                    out, l2 = (lrs_lean_eval(pvec, tgt[:, e], C, False) if lrs else rqs_lean_eval(pvec, tgt[:, e], C, False))
                    tgt[:, e] = out
                    ld2 = ld2 + l2
            sign = 1.0
            continue
        if kind in (21, 22):                                  # lean MADE layer (made_lean)
            nA2 = (EPL * steps2 + 3) & ~3
            A1 = prm[off:off + 2 * EPL * 64].reshape(2 * EPL // 4, 64, 4)
            b1 = prm[off + 2 * EPL * 64:off + 2 * EPL * 64 + 16]
            o2 = off + 2 * EPL * 64 + 16
            A2 = prm[o2:o2 + nA2 * 64].reshape(nA2 // 4, 64, 4)
            b2 = prm[o2 + nA2 * 64:o2 + nA2 * 64 + EPL * 16]
            pre = prm[o2 + nA2 * 64 + EPL * 16:]
            acc = b1[(4 * q)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
            for s_ in range(EPL):
                acc = _mfma(A1[s_ // 4, :, s_ % 4], a[:, s_], acc)
                acc = _mfma(A1[EPL // 4 + s_ // 4, :, s_ % 4], b[:, s_], acc)
            a = (pre[idx][:, :, None] * a + pre[D + idx][:, :, None]).clone()
            b = (pre[HALF + idx][:, :, None] * b + pre[D + HALF + idx][:, :, None]).clone()
            hid = 1.0 - 2.0 / (torch.exp2(acc) + 1.0)
            for t_ in range(EPL):
                o = b2[((t_ * 4 + q) * 4)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
                for k in range(steps2):
                    e_ = t_ * steps2 + k
                    o = _mfma(A2[e_ // 4, :, e_ % 4], hid[:, k], o)
                plane_t, tt = (a, t_) if t_ < EPL // 2 else (b, t_ - EPL // 2)
                for i in range(2):
                    al = torch.exp2(o[:, 2 * i]) + 1e-10
                    ld2 = ld2 + torch.log2(al)
                    e = 2 * tt + i
                    plane_t[:, e] = al * plane_t[:, e] + o[:, 2 * i + 1] if kind == 21 else (plane_t[:, e] - o[:, 2 * i + 1]) / al
            sign = 1.0 if kind == 21 else -1.0
            continue
        lk = kind - 12
        affine = lk < 2
        T2 = EPL // 2 if affine else EPL // 4
        if (kind, off) in op_extra and int(op_extra[(kind, off)][0]) == 256:      # bf16 x 3 operands (couple_lean3)
            A1 = prm[off:off + EPL * 64].reshape(EPL // 4, 64, 4)
            b1 = prm[off + EPL * 64:off + EPL * 64 + 16]
            o2 = off + EPL * 64 + 16
            raw = params.contiguous().view(torch.int32)
            A23 = raw[o2:o2 + T2 * 2 * 64 * 4].reshape(T2, 2, 64, 4)
            pre = prm[o2 + T2 * 2 * 64 * 4:]
            src, tgt = (a, b) if plane == 0 else (b, a)
            acc = b1[(4 * q)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
            for s_ in range(EPL):
                acc = _mfma(A1[s_ // 4, :, s_ % 4], src[:, s_], acc)
            tgt = (pre[idx][:, :, None] * tgt + pre[HALF + idx][:, :, None]).clone()
            hid = 1.0 - 2.0 / (torch.exp2(acc) + 1.0)
            hid[q == 3, 3] = 1.0
            h_hi, h_mid, h_lo = _pieces(hid.float())
            for t_ in range(T2):
                w_hi, w_mid = _unpack_bf16(A23[t_, 0][:, 0:2]), _unpack_bf16(A23[t_, 0][:, 2:4])
                w_lo = _unpack_bf16(A23[t_, 1][:, 0:2])
                assert torch.equal(w_hi, _unpack_bf16(A23[t_, 1][:, 2:4]))
                o = torch.zeros(64, 4, W, dtype=torch.float64)
                for Wl, Bl in ((w_hi, h_hi), (w_mid, h_hi), (w_hi, h_mid), (w_mid, h_mid), (w_lo, h_hi), (w_hi, h_lo)):
                    for i in range(4):
                        o = _mfma(Wl[:, i], Bl[:, i], o)
                if affine:
                    for i in range(2):
                        e = 2 * t_ + i
                        al = torch.exp2(o[:, 2 * i]) + 1e-10
                        ld2 = ld2 + torch.log2(al)
                        tgt[:, e] = al * tgt[:, e] + o[:, 2 * i + 1] if lk == 0 else (tgt[:, e] - o[:, 2 * i + 1]) / al
                    sign = 1.0 if lk == 0 else -1.0
                else:
                    for i in range(4):
                        e = 4 * t_ + i
                        tgt[:, e] = tgt[:, e] + o[:, i] if lk == 2 else tgt[:, e] - o[:, i]
            if plane == 0:
                b = tgt
            else:
                a = tgt
            continue
        nA2 = (T2 * steps2 + 3) & ~3
        A1 = prm[off:off + EPL * 64].reshape(EPL // 4, 64, 4)
        b1 = prm[off + EPL * 64:off + EPL * 64 + 16]
        o2 = off + EPL * 64 + 16
        A2 = prm[o2:o2 + nA2 * 64].reshape(nA2 // 4, 64, 4)
        b2 = prm[o2 + nA2 * 64:o2 + nA2 * 64 + T2 * 16]
        pre = prm[o2 + nA2 * 64 + T2 * 16:]
        src, tgt = (a, b) if plane == 0 else (b, a)
        acc = b1[(4 * q)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
        for s_ in range(EPL):
            acc = _mfma(A1[s_ // 4, :, s_ % 4], src[:, s_], acc)
        if cs:                                                # the context's columns of W1: A1c[64][4] behind pre_t
            A1c = pre[2 * HALF:2 * HALF + 256].reshape(64, 4)
            for s_ in range(cs):
                acc = _mfma(A1c[:, s_], cx[s_], acc)
        tgt = pre[idx][:, :, None] * tgt + pre[HALF + idx][:, :, None]
        hid = 1.0 - 2.0 / (torch.exp2(acc) + 1.0)             # (64, 4, W)
        tgt = tgt.clone()
        for t_ in range(T2):
            o = b2[((t_ * 4 + q) * 4)[:, None] + torch.arange(4)[None, :]][:, :, None].expand(64, 4, W).clone()
            for k in range(steps2):
                e_ = t_ * steps2 + k
                o = _mfma(A2[e_ // 4, :, e_ % 4], hid[:, k], o)
            if affine:
                for i in range(2):
                    e = 2 * t_ + i
                    al = torch.exp2(o[:, 2 * i]) + 1e-10
                    ld2 = ld2 + torch.log2(al)
                    tgt[:, e] = al * tgt[:, e] + o[:, 2 * i + 1] if lk == 0 else (tgt[:, e] - o[:, 2 * i + 1]) / al
                sign = 1.0 if lk == 0 else -1.0
            else:
                for i in range(4):
                    e = 4 * t_ + i
                    tgt[:, e] = tgt[:, e] + o[:, i] if lk == 2 else tgt[:, e] - o[:, i]
        if plane == 0:
            b = tgt
        else:
            a = tgt
    ld = ld + sign * LN2 * ld2
    ld_row = torch.zeros(16, W, dtype=torch.float64)
    ld_row.index_add_(0, j, ld)
    out = torch.zeros(W, 16, D, dtype=torch.float64)
    for e in range(EPL):
        out[:, j, EPL * q + e] = a[:, e].t()
        out[:, j, HALF + EPL * q + e] = b[:, e].t()
    return out.reshape(N, D), ld_row.t().reshape(N)
