"""Float64 emulator of ``tfk_glow_coupling`` / ``tfk_rows_fma`` as include/tfk.h documents them: decodes the
packed buffers of a compiled image program (torchflows_amd/image_program.py) exactly the way the kernel reads them --
index tables, pending maps, packed conv weights, background images, MFMA tile order of the folded Linear layer -- and
checks the one structural assumption the kernel relies on: outside the windows that csrc/tfk_glow.hip computes per
sample, the conv blocks' outputs equal the background images.  Test infrastructure: lets the host-side compiler be
verified without a GPU (tests/test_image_program_cpu.py) and gives the GPU tests a second, independent expected value.

What it restates, in the reference's terms (paths relative to torchflows/): the coupling skeleton
bijections/finite/autoregressive/layers_base.py:145-163; the conditioner bijections/finite/multiscale/conditioning/classic.py
(ConvModifier :8-42 as ONE zero-padded convolution, ConvNetBlock.forward :60-61 = conv3x3 -> ReLU -> MaxPool2d(2) ->
BatchNorm2d in inference form, the second modifier + Linear :105-122) and the (-2, 2) bound of
conditioning/transforms.py:107-113; Affine.forward / inverse transformers/linear/affine.py:33-59, Shift :137-159,
LUTransformer transformers/linear/matrix.py:22-99 under Invertible1x1ConvolutionTransformer linear/convolution.py:33-64
(log-det once per sample).  Pinned to the reference through tests/golden/flow_glow_3x32x32.npz and
flow_glow_3x8x8.npz (tests/test_image_program_cpu.py)."""
import numpy as np
import torch

C0 = float(np.float32(np.log(1 - 1e-10)))
F = torch.nn.functional


def _windows(hi, wi, oy, ox):   # (hi, wi: the rectangle of the frame that is not the modifier's bias)
    """The windows of csrc/tfk_glow.hip:glow_windows (pooled-1 rows / cols, pooled-2 rows / cols), half-open."""
    def axis(o, n):
        lo, hi_ = max(o - 1, 0), min(o + n + 1, 32)
        c0, c1 = lo & ~1, (hi_ + 1) & ~1
        p0, p1 = c0 // 2, c1 // 2
        q0, q1 = max(p0 - 1, 0), min(p1 + 1, 16)
        d0, d1 = q0 & ~1, (q1 + 1) & ~1
        return (p0, p1), (d0 // 2, d1 // 2)
    (y1, y2), (x1, x2) = axis(oy, hi), axis(ox, wi)
    return (y1, x1), (y2, x2)


def _unpack_weights(w, c_in):
    o = 0
    def take(n):
        nonlocal o
        v = w[o:o + n]
        o += n
        return v
    out = dict(Wm=take(4 * c_in), bm=take(4))           # (c_in here = channels x kernel taps of the first modifier)
    out["W1"] = take(288).view(4, 3, 3, 8).permute(3, 0, 1, 2)     # packed [ci][ky][kx][co]
    out["b1"], out["sc1"], out["sh1"] = take(8), take(8), take(8)
    out["W2"] = take(576).view(8, 3, 3, 8).permute(3, 0, 1, 2)
    out["b2"], out["sc2"], out["sh2"] = take(8), take(8), take(8)
    out["W3"] = take(288).view(8, 3, 3, 4).permute(3, 0, 1, 2)
    out["b3"] = take(4)
    out["m2"], out["bm2"] = take(4), take(1)
    assert o == w.numel()
    return out


def _block(x, W, b, sc=None, sh=None):
    y = F.max_pool2d(torch.relu(F.conv2d(x, W, b, padding=1)), 2)
    if sc is not None:
        y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    return y


def run_step(rows, logdet, step, check_windows=True):
    """rows (N, D) float64, logdet (N,) float64 -- updated in place like the kernel does."""
    L = step.layer
    src_idx, src_st, tgt_idx, tgt_st, weights, bg1, bg2, w_eff, b_eff = (t.detach().cpu() for t in step.keep)
    dd = lambda t: t.double()
    N = rows.shape[0]
    c_in, hi, wi, oy, ox = L.c_in, L.hi, L.wi, L.oy, L.ox
    kh, kw = max(L.kh, 1), max(L.kw, 1)
    w = _unpack_weights(dd(weights), c_in * kh * kw)
    st = dd(src_st).view(-1, 2)
    v = rows[:, src_idx.long()] * st[:, 0] + st[:, 1]
    img = v.view(N, c_in, hi, wi)
    # the first ConvModifier as the reference runs it: ONE zero-padded convolution (classic.py:35-42)
    frame = F.conv2d(img, w["Wm"].view(4, c_in, kh, kw), w["bm"], padding=(oy + kh - 1, ox + kw - 1))
    assert frame.shape[-2:] == (32, 32)
    p1 = _block(frame, w["W1"], w["b1"], w["sc1"], w["sh1"])
    p2 = _block(p1, w["W2"], w["b2"], w["sc2"], w["sh2"])
    if check_windows:
        (y1, x1), (y2, x2) = _windows(hi + kh - 1, wi + kw - 1, oy, ox)
        m1 = torch.ones(16, 16, dtype=torch.bool)
        m1[y1[0]:y1[1], x1[0]:x1[1]] = False
        m2 = torch.ones(8, 8, dtype=torch.bool)
        m2[y2[0]:y2[1], x2[0]:x2[1]] = False
        e1 = float((p1[:, :, m1] - dd(bg1).view(8, 16, 16)[:, m1]).abs().max()) if m1.any() else 0.0
        e2 = float((p2[:, :, m2] - dd(bg2).view(8, 8, 8)[:, m2]).abs().max()) if m2.any() else 0.0
        assert e1 < 1e-6 and e2 < 1e-6, (e1, e2)
    p3 = _block(p2, w["W3"], w["b3"])                                   # (N, 4, 4, 4), BatchNorm 3 folded below
    V = torch.einsum("c,nchw->nhw", w["m2"], p3).reshape(N, 16) + w["bm2"]
    n_tiles = w_eff.numel() // 256
    A = dd(w_eff).view(n_tiles, 4, 16, 4)            # [t][q][i][ks] = W[16 t + i][4 ks + q]
    W_k = A.permute(0, 2, 3, 1).reshape(n_tiles * 16, 16)
    h = (V @ W_k.t() + dd(b_eff)) / 1.4426950408889634        # the packed rows carry a factor log2(e)
    h = 4.0 / (1.0 + torch.exp(-h)) - 2.0
    tst = dd(tgt_st).view(-1, 2)[:L.T]             # (affine tables are padded to whole groups of 16 targets)
    tgt = tgt_idx.long()[:L.T]
    xb = rows[:, tgt] * tst[:, 0] + tst[:, 1]
    if L.kind == 0:
        rank = torch.arange(L.T)                   # target of rank 16 m + j: logit in row 32 m + j, shift in 32 m + 16 + j
        row_u = 32 * (rank // 16) + rank % 16
        u, beta = h[:, row_u], h[:, row_u + 16]
        wl = u * 0.5 + C0
        alpha = torch.exp(wl) + 1e-10
        if not step.inverse:
            rows[:, tgt] = alpha * xb + beta
            logdet += torch.log(alpha).sum(1)
        else:
            rows[:, tgt] = (xb - beta) / alpha
            logdet -= torch.log(alpha).sum(1)
    elif L.kind == 2:                                  # shift: row 16 m + j of the tiles = the listed target's shift
        beta = h[:, :L.T]
        rows[:, tgt] = xb - beta if step.inverse else xb + beta
    else:
        n, HW = L.n_ch, L.hw
        h = h[:, :L.n_params]
        n_off = n * (n - 1) // 2
        ud = torch.exp(h[:, :n]) / 10 + 1
        U = torch.zeros(N, n, n, dtype=torch.float64)
        r, c = torch.triu_indices(n, n, 1)
        U[:, r, c] = h[:, n:n + n_off] / 10
        U[:, range(n), range(n)] = ud
        Lm = torch.zeros(N, n, n, dtype=torch.float64)
        r, c = torch.tril_indices(n, n, -1)
        Lm[:, r, c] = h[:, n + n_off:] / 10
        Lm[:, range(n), range(n)] = 1.0
        X = xb.view(N, n, HW)
        A_ = Lm @ U
        if not step.inverse:
            rows[:, tgt] = (A_ @ X).reshape(N, -1)
            logdet += torch.log(ud).sum(1)
        else:
            rows[:, tgt] = torch.linalg.solve(A_, X).reshape(N, -1)
            logdet -= torch.log(ud).sum(1)


def run_program(prog, x, check_windows=True):
    """(z, log_det) in float64 of a compiled ImageProgram on host rows ``x`` (any float dtype)."""
    N = x.shape[0]
    rows = x.reshape(N, prog.D).double().clone()
    logdet = torch.full((N,), float(prog.ld_const), dtype=torch.float64)
    for step in prog.steps:
        run_step(rows, logdet, step, check_windows)
    if prog.flush is not None:
        f = prog.flush.detach().cpu().double()
        rows = rows * f[:, 0] + f[:, 1]
    return rows.view(x.shape), logdet
