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


def run_lean(ops, params, rows, D):
    """rows (N, D) fp64, N a multiple of 16 -> (rows out, logdet) of one lean segment."""
    EPL, HALF = D // 8, D // 2
    N = rows.shape[0]
    W = N // 16
    lane = torch.arange(64)
    q, j = lane >> 4, lane & 15
    x = rows.reshape(W, 16, D)
    # lane registers: a[l, e, w] = row j of wave w, element EPL q + e of plane A
    idx = (EPL * q)[:, None] + torch.arange(EPL)[None, :]                      # (64, EPL)
    a = x[:, j][:, torch.arange(64)[:, None], idx].permute(1, 2, 0).clone()   # (64, EPL, W)
    b = x[:, j][:, torch.arange(64)[:, None], HALF + idx].permute(1, 2, 0).clone()
    ld2 = torch.zeros(64, W, dtype=torch.float64)
    ld = torch.zeros(64, W, dtype=torch.float64)
    prm = params.double()
    sign = 0.0
    for kind, plane, steps2, off in ops:
        if kind == 16:                                        # TFK_OP_EW_FMA
            s, t = prm[off:off + D], prm[off + D:off + 2 * D]
            a = s[idx][:, :, None] * a + t[idx][:, :, None]
            b = s[HALF + idx][:, :, None] * b + t[HALF + idx][:, :, None]
            ld[q == 0] += prm[off + 2 * D]
            continue
        lk = kind - 12
        affine = lk < 2
        T2 = EPL // 2 if affine else EPL // 4
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
