"""Image / multiscale flows as ONE libtfk launch per coupling (config 5: ``AffineGlow((3, 32, 32))``).

The reference runs ``MultiscaleBijection.forward`` (multiscale/base.py:249-296) as a Python recursion of layers, each
of which materialises its tensors: ActNorm (layers.py:39-69), the masked gather / scatter of the coupling
(layers_base.py:145-163), the ConvNet conditioner's eight modules (multiscale/conditioning/classic.py:45-122), four
squeeze / unsqueeze permutations and a chunk / cat per block (multiscale/base.py:117-175, 271-280).  Here the whole
recursion is *compiled* once per parameter version into a flat list of ``tfk_glow_coupling`` launches
(csrc/tfk_glow.hip) that work in place on one ``(N, D)`` row buffer which keeps its ``(c, h, w)`` layout throughout:

* squeeze / unsqueeze / chunk only change which physical position a logical element of the current block has: a
  ``logical -> physical`` index vector is carried through the recursion and the coupling's two masks become two int32
  tables of physical positions (conditioner-image order, transformer-target order);
* ActNorm layers are *deferred*: every physical element carries a pending map ``v = s * raw + t`` (composed here in
  float64); a coupling applies it where it reads the element and stores transformed targets in final form, the constant
  log-dets are summed up front and whatever is still pending after the last layer is flushed by one ``tfk_rows_fma``;
* per coupling everything that does not depend on the sample is evaluated here, on the host, in float64: BatchNorm
  as scale / shift, the conv blocks' response to the all-bias image outside the source image's receptive field
  (``bg1`` / ``bg2``), the second ConvModifier + BatchNorm 3 + the 84 constant inputs of the Linear layer folded into a
  ``(n_params, 16)`` matrix in MFMA tile order.

A model the compiler does not cover (another conditioner, a modifier that is a real convolution -- images larger than 32
pixels --, a transformer other than Affine / Shift / the 1x1 convolution, a context, ``invert()``-ed layers, an
ActNorm that still waits for its first batch) returns None and runs layer by layer as before.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn

from torchflows_amd import native
from torchflows_amd.utils import debug_switch

FORWARD, INVERSE = 0, 1


@dataclass
class Step:
    layer: "native.GlowLayer"
    inverse: bool
    keep: tuple                      # the device tensors the struct points into
    info: dict = field(default_factory=dict)


@dataclass
class Level:
    """Consecutive steps that work on the same set of row elements: ONE tfk_glow_level launch (rows held in the LDS)."""
    first: int                       # steps[first : first + count]
    count: int
    row_idx: Optional[torch.Tensor]  # device int32 (D_level,), ascending physical positions; None = the whole row
    D_level: int
    blob_host: torch.Tensor          # uint8, launch shape + geometry + background cell lists (native.glow_level_pack)
    blob_dev: torch.Tensor
    keep: tuple                      # device tensors the blob points into
    info: dict = field(default_factory=dict)


@dataclass
class ImageProgram:
    D: int
    steps: List[Step]
    ld_const: float
    flush: Optional[torch.Tensor]    # (D, 2) pending maps left behind the last layer, None if all identity
    version: int
    levels: Optional[List[Level]] = None     # the same steps grouped into level launches (None: one launch per step)


class _Decline(Exception):
    pass


def enabled() -> bool:
    return os.environ.get("TORCHFLOWS_AMD_IMAGE_PROGRAM", "1") != "0"


def levels_enabled() -> bool:
    """TORCHFLOWS_AMD_GLOW_LEVELS=1: one launch per LEVEL (tfk_glow_level: the level's row elements held in the LDS, HBM
    traffic 240 -> ~100 KB per row) instead of one per coupling.  OFF by default: measured on MI355X (round 4, 65 536 rows
    of AffineGlow((3, 32, 32)), same box) the three level launches take 20.2 ms against 14.2 ms for the 19 coupling
    launches -- four resident samples per workgroup (12 KB of row + 23.5 KB of activations each at the first level) leave
    one workgroup per CU, whose stages run back to back instead of overlapping."""
    return os.environ.get("TORCHFLOWS_AMD_GLOW_LEVELS", "0") == "1"


def _bn_affine(bn: nn.BatchNorm2d):
    if bn.training or not bn.track_running_stats or bn.running_var is None:
        raise _Decline("BatchNorm in training mode")
    var, mean = bn.running_var.detach().double().cpu(), bn.running_mean.detach().double().cpu()
    gamma = torch.ones_like(var) if bn.weight is None else bn.weight.detach().double().cpu()
    beta = torch.zeros_like(var) if bn.bias is None else bn.bias.detach().double().cpu()
    scale = gamma / torch.sqrt(var + bn.eps)
    return scale, beta - mean * scale


def _block(x, conv_w, conv_b, scale, shift):
    y = torch.nn.functional.conv2d(x[None], conv_w, conv_b, padding=1)
    y = torch.nn.functional.max_pool2d(torch.relu(y), 2)[0]
    return y * scale.view(-1, 1, 1) + shift.view(-1, 1, 1)


def _pack_conditioner(cond, c_in: int, hi: int, wi: int):
    """Host-side (float64) evaluation of everything sample-independent in a ConvNetConditioner; fp32 CPU tensors."""
    from torchflows_amd.bijections.finite.multiscale.conditioning.classic import (ConvModifier, ConvNet,
                                                                                  ConvNetConditioner)
    if not isinstance(cond, ConvNetConditioner) or cond.n_global_parameters != 0 or cond.context_shape is not None:
        raise _Decline("conditioner is not a plain ConvNetConditioner")
    if cond.output_lower_bound != -2.0 or cond.output_upper_bound != 2.0:
        raise _Decline("conditioner bounds are not (-2, 2)")
    net = cond.network
    if not isinstance(net, ConvNet) or len(net.blocks) != 5:
        raise _Decline("ConvNet is not three blocks between two modifiers")
    mod1, b1, b2, b3, mod2 = net.blocks
    if not (isinstance(mod1, ConvModifier) and isinstance(mod2, ConvModifier)):
        raise _Decline("unexpected ConvNet layout")
    c1, c2 = mod1.conv, mod2.conv
    kh, kw = (int(k) for k in c1.kernel_size)
    if kh > 2 or kw > 2 or tuple(c1.weight.shape) != (4, c_in, kh, kw) or hi > 32 or wi > 32:
        raise _Decline("first ConvModifier is not a padding convolution onto (4, 32, 32)")
    py, px = (int(p) for p in c1.padding)
    if (hi + 2 * py - kh + 1 != 32 or wi + 2 * px - kw + 1 != 32 or py < kh - 1 or px < kw - 1
            or tuple(c1.stride) != (1, 1) or tuple(c1.dilation) != (1, 1) or c1.bias is None):
        raise _Decline("first ConvModifier does not produce a 32x32 frame")
    oy, ox = py - (kh - 1), px - (kw - 1)                   # first output row / column that sees the image
    shapes = [(8, 4, 3, 3), (8, 8, 3, 3), (4, 8, 3, 3)]
    for blk, shp in zip((b1, b2, b3), shapes):
        cv = blk.conv
        if (tuple(cv.weight.shape) != shp or tuple(cv.padding) != (1, 1) or tuple(cv.stride) != (1, 1)
                or cv.bias is None or not isinstance(blk.pool, nn.MaxPool2d) or blk.pool.kernel_size != 2):
            raise _Decline("conv blocks are not the default (8, 8, 4) kernels with pooling")
    if (tuple(c2.weight.shape) != (1, 4, 1, 1) or tuple(int(p) for p in c2.padding) != (3, 3) or c2.bias is None
            or net.linear.in_features != 100):
        raise _Decline("second ConvModifier is not (4, 4, 4) -> (1, 10, 10)")
    dd = lambda t: t.detach().double().cpu()
    Wm, bm = dd(c1.weight).reshape(4, c_in * kh * kw), dd(c1.bias)
    parts = [Wm.reshape(-1), bm]
    affs = []
    for blk in (b1, b2, b3):
        sc, sh = _bn_affine(blk.bn)
        affs.append((sc, sh))
    for i, blk in enumerate((b1, b2)):
        parts += [dd(blk.conv.weight).permute(1, 2, 3, 0).reshape(-1), dd(blk.conv.bias), affs[i][0], affs[i][1]]
    parts += [dd(b3.conv.weight).permute(1, 2, 3, 0).reshape(-1), dd(b3.conv.bias)]       # [ci][ky][kx][co]
    wm2, bm2 = dd(c2.weight).reshape(4), dd(c2.bias).reshape(())
    sc3, sh3 = affs[2]
    parts += [wm2 * sc3, (bm2 + (wm2 * sh3).sum()).reshape(1)]
    weights = torch.cat(parts).float()
    assert weights.numel() == int(native.lib().tfk_glow_weight_floats(c_in * kh * kw))
    # the blocks' response to the all-bias frame
    frame = bm.view(4, 1, 1).expand(4, 32, 32).contiguous()
    bg1 = _block(frame, dd(b1.conv.weight), dd(b1.conv.bias), *affs[0])
    bg2 = _block(bg1, dd(b2.conv.weight), dd(b2.conv.bias), *affs[1])
    # Linear layer on the (1, 10, 10) image whose frame is the second modifier's bias
    W, b = dd(net.linear.weight), dd(net.linear.bias)
    n_params = W.shape[0]
    interior = torch.tensor([(3 + y) * 10 + 3 + x for y in range(4) for x in range(4)])
    frame_mask = torch.ones(100, dtype=torch.bool)
    frame_mask[interior] = False
    W_eff = W[:, interior]
    b_eff = b + W[:, frame_mask].sum(1) * bm2
    return dict(oy=oy, ox=ox, kh=kh, kw=kw, n_params=n_params, weights=weights, bg1=bg1.float().contiguous(),
                bg2=bg2.float().contiguous(), W_eff=W_eff, b_eff=b_eff)


def _tile_pack(W_rows: torch.Tensor, b_rows: torch.Tensor):
    """(rows, 16) / (rows,) in kernel row order -> MFMA operand tiles [t][64 lanes][4 k-steps] and the padded bias, both
    multiplied by log2(e):
    tile t, lane l = 16 q + i, k-step ks  <-  W[16 t + i][4 ks + q]."""
    n = W_rows.shape[0]
    n_tiles = (n + 15) // 16
    log2e = 1.4426950408889634          # the kernel's sigmoid is 1 / (1 + exp2(-h')): h' = h log2(e)
    Wp = torch.zeros(n_tiles * 16, 16, dtype=torch.float64)
    Wp[:n] = W_rows * log2e
    bp = torch.zeros(n_tiles * 16, dtype=torch.float64)
    bp[:n] = b_rows * log2e
    w_tiles = Wp.view(n_tiles, 16, 4, 4).permute(0, 3, 1, 2).reshape(-1)
    return w_tiles.float().contiguous(), bp.float().contiguous()


def layer_cost(layer) -> dict:
    """What one tfk_glow_coupling launch moves and computes PER ROW (for bench.py's roofline): algorithmic HBM bytes
    (source elements read, targets read and written, the log-det read and written) and the multiply-adds the kernel
    executes -- the conv blocks on the windows of csrc/tfk_glow.hip:glow_windows, the Linear layer at K = 16 -- next to
    the multiply-adds of the reference's formulation (full 32x32 / 16x16 / 8x8 convolutions, Linear at K = 100)."""
    def axis(o, n):
        lo, hi_ = max(o - 1, 0), min(o + n + 1, 32)
        c0, c1 = lo & ~1, (hi_ + 1) & ~1
        q0, q1 = max(c0 // 2 - 1, 0), min(c1 // 2 + 1, 16)
        return c1 - c0, ((q1 + 1) & ~1) - (q0 & ~1)
    kh, kw = max(layer.kh, 1), max(layer.kw, 1)
    (c1h, c2h), (c1w, c2w) = axis(layer.oy, layer.hi + kh - 1), axis(layer.ox, layer.wi + kw - 1)
    S = layer.c_in * layer.hi * layer.wi
    macs = 4 * S * kh * kw + c1h * c1w * 8 * 36 + c2h * c2w * 8 * 72 + 64 * 4 * 72 + 64 + 16 * layer.n_params
    ref = 4 * 1024 * layer.c_in * kh * kw + 1024 * 8 * 36 + 256 * 8 * 72 + 64 * 4 * 72 + 16 * 4 + 100 * layer.n_params
    if layer.kind == 1:
        macs += layer.hw * layer.n_ch * layer.n_ch
        ref += layer.hw * layer.n_ch * layer.n_ch
    return dict(bytes=4 * (S + 2 * layer.T) + 8, flops=2 * macs, flops_reference=2 * ref)


class _Builder:
    def __init__(self, D: int, device: torch.device):
        self.D, self.device = D, device
        self.s = torch.ones(D, dtype=torch.float64)
        self.t = torch.zeros(D, dtype=torch.float64)
        self.ld = 0.0
        self.steps: List[Step] = []

    # -- elementwise layers: pending maps ---------------------------------------------------------------------
    def elementwise(self, layer, d: int, M: torch.Tensor) -> None:
        from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm
        kind = layer.transformer.native_kind
        if kind not in ("affine", "inverse_affine") or not layer.use_global_parameters:
            raise _Decline("elementwise layer without a constant affine map")
        if isinstance(layer, ActNorm) and layer.training and layer.first_training_batch_pass:
            raise _Decline("ActNorm waits for its first batch")
        value = layer.value.detach().reshape(-1, 2)
        alpha = layer.transformer.constrain_scale(value[:, 0]).double().cpu()       # affine.py:33-34, fp32 like the reference
        beta = value[:, 1].double().cpu()
        divide = (d == INVERSE) != (kind == "inverse_affine")
        if divide:                                           # v' = (v - beta) / alpha
            self.s[M] = self.s[M] / alpha
            self.t[M] = (self.t[M] - beta) / alpha
            self.ld -= float(torch.log(alpha).sum())
        else:                                                # v' = alpha v + beta
            self.s[M] = self.s[M] * alpha
            self.t[M] = self.t[M] * alpha + beta
            self.ld += float(torch.log(alpha).sum())

    # -- couplings: one launch each --------------------------------------------------------------------------------
    def coupling(self, layer, d: int, M: torch.Tensor) -> None:
        kind = layer.transformer.native_kind
        if kind not in ("affine", "shift", "conv1x1") or layer.context_shape is not None:
            raise _Decline(f"no fused kernel for transformer kind {kind!r}")
        cs = tuple(layer.coupling.constant_shape)
        if len(cs) != 3:
            raise _Decline("conditioner input is not an image")
        c_in, hi, wi = (int(v) for v in cs)
        pk = _pack_conditioner(layer.conditioner_transform, c_in, hi, wi)
        src = M[layer._source_index.cpu()]
        tgt = M[layer._target_index.cpu()]
        T = int(tgt.numel())
        n_ch = hw = 0
        if kind == "conv1x1":
            n_ch = int(layer.transformer.n_channels)
            if n_ch > 16 or T % n_ch:
                raise _Decline("1x1 convolution over more than 16 channels")
            hw = T // n_ch
            if pk["n_params"] != n_ch + n_ch * (n_ch - 1):
                raise _Decline("unexpected LU parameter count")
        elif pk["n_params"] != (T if kind == "shift" else 2 * T):
            raise _Decline("unexpected parameter count")
        if int(src.numel()) != c_in * hi * wi:
            raise _Decline("source mask does not fill the conditioner image")
        dev = self.device
        st = lambda idx: torch.stack([self.s[idx], self.t[idx]], dim=1).float().contiguous()
        log2e = 1.4426950408889634
        aux = dict(src=src.clone(), tgt=tgt.clone(), W4=None, b4=None, weights=pk["weights"].contiguous())
        if kind == "shift":                                # one tile per 16 targets (sorted by position): their shifts
            order = torch.argsort(tgt, stable=True)
            tgt_k = tgt[order]
            n_groups = (T + 15) // 16
            w_tiles, b_tiles = _tile_pack(pk["W_eff"][order], pk["b_eff"][order])
            pad = (-T) % 64                                # (whole tiles of the level kernel: 64 parameters)
            tgt_i = torch.cat([tgt_k.to(torch.int32), torch.zeros(pad, dtype=torch.int32)])
            tgt_m = torch.cat([st(tgt_k), torch.zeros(pad, 2)])
            R = T + pad
            W4 = torch.zeros(R, 16, dtype=torch.float64)
            b4 = torch.zeros(R, dtype=torch.float64)
            W4[:T], b4[:T] = pk["W_eff"][order] * log2e, pk["b_eff"][order] * log2e
            aux.update(tgt=tgt_k.clone(), W4=W4.float().contiguous(), b4=b4.float().contiguous())
        elif kind == "affine":
            # the kernel takes the targets in any order: ascending physical position, so that the 16 targets of a tile
            # pair are neighbours in the row; tile 2 m = their scale logits (h[..., t, 0]), tile 2 m + 1 their shifts
            order = torch.argsort(tgt, stable=True)
            tgt_k = tgt[order]
            n_groups = (T + 15) // 16
            rank = torch.arange(T)
            row_u = 32 * (rank // 16) + rank % 16
            Wk = torch.zeros(32 * n_groups, 16, dtype=torch.float64)
            bk = torch.zeros(32 * n_groups, dtype=torch.float64)
            Wk[row_u], bk[row_u] = pk["W_eff"][2 * order], pk["b_eff"][2 * order]
            Wk[row_u + 16], bk[row_u + 16] = pk["W_eff"][2 * order + 1], pk["b_eff"][2 * order + 1]
            w_tiles, b_tiles = _tile_pack(Wk, bk)
            pad = (-T) % 64                                # the kernels read the tables in whole groups of 16 / 32 targets
            tgt_i = torch.cat([tgt_k.to(torch.int32), torch.zeros(pad, dtype=torch.int32)])
            tgt_m = torch.cat([st(tgt_k), torch.zeros(pad, 2)])
            # the level kernel's operands: rows [u of target 0, beta of target 0, u of target 1, ...] in sorted order
            R = 2 * (T + pad)
            W4 = torch.zeros(R, 16, dtype=torch.float64)
            b4 = torch.zeros(R, dtype=torch.float64)
            W4[:2 * T] = (pk["W_eff"].view(T, 2, 16)[order] * log2e).reshape(2 * T, 16)
            b4[:2 * T] = (pk["b_eff"].view(T, 2)[order] * log2e).reshape(2 * T)
            aux.update(tgt=tgt_k.clone(), W4=W4.float().contiguous(), b4=b4.float().contiguous())
        else:
            w_tiles, b_tiles = _tile_pack(pk["W_eff"], pk["b_eff"])
            tgt_i, tgt_m = tgt.to(torch.int32), st(tgt)
        keep = (src.to(torch.int32).to(dev), st(src).to(dev), tgt_i.to(dev), tgt_m.to(dev),
                pk["weights"].to(dev), pk["bg1"].to(dev), pk["bg2"].to(dev), w_tiles.to(dev), b_tiles.to(dev))
        env = lambda k: int(debug_switch("glow_" + k.lower(), "0") or 0)
        L = native.GlowLayer(kind={"affine": 0, "conv1x1": 1, "shift": 2}[kind], c_in=c_in, hi=hi, wi=wi, oy=pk["oy"], ox=pk["ox"],
                             kh=pk["kh"], kw=pk["kw"], T=T, n_params=pk["n_params"], n_ch=n_ch, hw=hw, slots=env("SLOTS"), block=env("BLOCK"),
                             cg1=env("CG1"), cg2=env("CG2"), grid=env("GRID"),
                             src_idx=keep[0].data_ptr(), src_st=keep[1].data_ptr(), tgt_idx=keep[2].data_ptr(),
                             tgt_st=keep[3].data_ptr(), weights=keep[4].data_ptr(), bg1=keep[5].data_ptr(),
                             bg2=keep[6].data_ptr(), w_eff=keep[7].data_ptr(), b_eff=keep[8].data_ptr())
        plan = native.glow_plan(L, self.D)                # validates the shape (raises NativeError otherwise)
        self.steps.append(Step(L, d == INVERSE, keep,
                               dict(kind=kind, image=(c_in, hi, wi), at=(pk["oy"], pk["ox"]), T=T, aux=aux, **plan)))
        self.s[tgt] = 1.0                                  # targets are stored in final form
        self.t[tgt] = 0.0


def _walk(b: _Builder, module, d: int, M: torch.Tensor) -> None:
    """Append the launches of ``module`` applied in direction ``d`` to the elements at physical positions ``M``."""
    from torchflows_amd.bijections.base import BijectiveComposition, method_direction
    from torchflows_amd.bijections.finite.autoregressive.layers_base import CouplingBijection, ElementwiseBijection
    from torchflows_amd.bijections.finite.multiscale.base import MultiscaleBijection, Squeeze
    if method_direction(module.forward) != FORWARD or method_direction(module.inverse) != INVERSE:
        raise _Decline("a layer's maps were exchanged by invert()")
    if isinstance(module, MultiscaleBijection):
        boards = list(module.checkerboard_layers)
        if module.n_blocks > 1:
            Ms = M[module.squeeze._fwd_index.cpu()]        # squeezed logical j  <-  unsqueezed logical fwd_index[j]
            Ma = M[module.alt_squeeze._fwd_index.cpu()]
            rest = Ma[Ma.numel() // 2:]                     # torch.chunk(x, 2, dim=-3)[1]: the last half of the channels
            chans = list(module.channel_wise_layers)
        if d == FORWARD:
            for layer in boards:
                _walk(b, layer, d, M)
            if module.n_blocks > 1:
                for layer in chans:
                    _walk(b, layer, d, Ms)
                _walk(b, module.small_bijection, d, rest)
        else:
            if module.n_blocks > 1:
                _walk(b, module.small_bijection, d, rest)
                for layer in chans[::-1]:
                    _walk(b, layer, d, Ms)
            for layer in boards[::-1]:
                _walk(b, layer, d, M)
    elif isinstance(module, BijectiveComposition):
        for layer in (module.layers if d == FORWARD else list(module.layers)[::-1]):
            _walk(b, layer, d, M)
    elif isinstance(module, Squeeze):
        raise _Decline("a bare Squeeze changes the event shape of the rows")
    elif isinstance(module, ElementwiseBijection):
        b.elementwise(module, d, M)
    elif isinstance(module, CouplingBijection):
        b.coupling(module, d, M)
    else:
        raise _Decline(f"no fused launch for {type(module).__name__}")


def _u16(values: torch.Tensor, device, pad_to: int = 4) -> torch.Tensor:
    """int64 values < 65536 as a device tensor of uint16 bit patterns (int16 storage), zero-padded to a multiple of
    ``pad_to`` entries (the kernel reads the target positions two or four at a time)."""
    import numpy as np
    a = values.numpy().astype(np.uint16)
    if a.size % pad_to:
        a = np.concatenate([a, np.zeros(pad_to - a.size % pad_to, np.uint16)])
    return torch.from_numpy(a.view(np.int16).copy()).to(device)


def build_levels(steps: List[Step], D: int, device: torch.device) -> Optional[List[Level]]:
    """Group consecutive steps with the same element footprint (sources + targets) into level launches; None when a
    level does not fit the level kernel (then the program keeps one launch per step)."""
    import ctypes as C
    env = lambda k: int(debug_switch("glow_level_" + k.lower(), "0") or 0)
    groups, foot = [], None
    for i, step in enumerate(steps):
        aux = step.info["aux"]
        f = torch.unique(torch.cat([aux["src"], aux["tgt"]]))            # ascending
        if foot is not None and f.numel() == foot.numel() and bool((f == foot).all()):
            groups[-1][1].append(i)
        else:
            groups.append((f, [i]))
            foot = f
    levels = []
    for foot, idxs in groups:
        Dl = int(foot.numel())
        if Dl > 65535:
            return None
        whole = Dl == D
        where = torch.full((D,), -1, dtype=torch.long)
        where[foot] = torch.arange(Dl)
        arr = (native.GlowLevelStep * len(idxs))()
        keep = []
        for k, i in enumerate(idxs):
            step, aux = steps[i], steps[i].info["aux"]
            src_loc, tgt_loc = _u16(where[aux["src"]], device), _u16(where[aux["tgt"]], device, 64)
            C.memmove(C.byref(arr[k].layer), C.byref(step.layer), C.sizeof(step.layer))
            arr[k].inverse = 1 if step.inverse else 0
            arr[k].src_loc, arr[k].tgt_loc = src_loc.data_ptr(), tgt_loc.data_ptr()
            arr[k].weights_host = aux["weights"].data_ptr()            # (CPU tensor, kept alive by the step's info)
            keep += [src_loc, tgt_loc]
            if aux["W4"] is not None:
                w4, b4 = aux["W4"].to(device), aux["b4"].to(device)
                arr[k].w4, arr[k].b4 = w4.data_ptr(), b4.data_ptr()
                keep += [w4, b4]
        try:
            blob = native.glow_level_pack(arr, D, Dl, env("SAMPLES"), env("BLOCK"))
        except native.NativeError:
            return None
        row_idx = None if whole else foot.to(torch.int32).to(device)
        levels.append(Level(idxs[0], len(idxs), row_idx, Dl, blob, blob.to(device), tuple(keep),
                            dict(native.glow_level_info(blob), steps=len(idxs), D_level=Dl)))
    return levels


def compile_program(module, d: int, device: torch.device) -> Optional[ImageProgram]:
    from torchflows_amd import fused
    if not enabled() or len(module.event_shape) != 3:
        return None
    D = int(module.n_dim)
    b = _Builder(D, device)
    try:
        with torch.no_grad():
            _walk(b, module, d, torch.arange(D))
    except (_Decline, native.NativeError):
        return None
    if not b.steps:
        return None
    pending = bool((b.s != 1.0).any() or (b.t != 0.0).any())
    flush = torch.stack([b.s, b.t], dim=1).float().contiguous().to(device) if pending else None
    levels = build_levels(b.steps, D, device) if levels_enabled() else None
    return ImageProgram(D, b.steps, b.ld, flush, fused._params_version(module), levels)


def get_program(module, d: int, device: torch.device) -> Optional[ImageProgram]:
    """The compiled program of ``module`` for direction ``d`` (cached per parameter version); None if not covered."""
    from torchflows_amd import fused
    cache = module.__dict__.setdefault("_tfk_image_programs", {})
    key = (d, device.index)
    version = fused._params_version(module)
    hit = cache.get(key)
    if hit is not None and hit[0] == version and hit[2] is not None and hit[2].stale(module):
        import warnings                                  # (the asynchronous .data-edit guard of fused.get_compiled)
        warnings.warn("torchflows_amd: parameters below this image flow changed without their version counters moving "
                      "(an in-place edit through .data?); the compiled program is rebuilt", fused.StaleProgramWarning,
                      stacklevel=3)
        fused.invalidate(module)
        cache = module.__dict__.setdefault("_tfk_image_programs", {})
        hit = None
    if hit is not None and hit[0] == version:
        return hit[1]
    prog = compile_program(module, d, device)
    guard = fused._Guard(module) if (prog is not None and device.type == "cuda" and fused.GUARD_EVERY > 0) else None
    cache[key] = (version, prog, guard)
    return prog


def log_prob(prog: ImageProgram, x: torch.Tensor, event_shape, loc: torch.Tensor, log_scale: torch.Tensor) -> torch.Tensor:
    """``base.log_prob(z) + log_det`` of the compiled chain without materialising z: the couplings, then ONE pass that
    applies the pending maps and evaluates the diagonal Gaussian (tfk_rows_fma_gauss_logprob) -- the flush's read + write and
    the density's read of every row become one read.  Rows wider than the fused kernel's LDS budget take ``run`` + the
    separate density kernel."""
    n_event = len(event_shape)
    batch = x.shape[:x.dim() - n_event]
    if prog.flush is None or prog.D > 3276:
        z, ld = run(prog, x, event_shape)
        out = torch.empty(ld.numel(), dtype=torch.float32, device=x.device)
        native.diag_gauss_logprob(z.reshape(-1, prog.D), loc, log_scale, ld.reshape(-1), out)
        return out.view(batch)
    rows, logdet = _couplings(prog, x)
    out = torch.empty(rows.shape[0], dtype=torch.float32, device=x.device)
    native.rows_fma_gauss_logprob(rows, prog.flush, loc, log_scale, logdet, out)
    return out.view(batch)


def _couplings(prog: ImageProgram, x: torch.Tensor):
    """(rows, logdet) behind the last coupling, pending maps NOT flushed."""
    if prog.levels is not None and x.device.type == "cuda":
        src = x.reshape(-1, prog.D)
        if not src.is_contiguous():
            src = src.contiguous()
        first = prog.levels[0]
        rows = torch.empty_like(src) if first.row_idx is None else src.clone()
        logdet = torch.full((rows.shape[0],), prog.ld_const, dtype=torch.float32, device=x.device)
        for k, lv in enumerate(prog.levels):
            native.glow_level(src if (k == 0 and lv.row_idx is None) else rows, rows, logdet, lv.row_idx,
                              lv.blob_host, lv.blob_dev)
        return rows, logdet
    rows = x.reshape(-1, prog.D).clone(memory_format=torch.contiguous_format)
    logdet = torch.full((rows.shape[0],), prog.ld_const, dtype=torch.float32, device=x.device)
    for step in prog.steps:
        native.glow_coupling(rows, logdet, step.layer, inverse=step.inverse)
    return rows, logdet


def run(prog: ImageProgram, x: torch.Tensor, event_shape):
    """(z, log_det) of the compiled chain on ``x`` (never modified: the launches work on a copy)."""
    n_event = len(event_shape)
    batch = x.shape[:x.dim() - n_event]
    rows, logdet = _couplings(prog, x)
    if prog.flush is not None:
        native.rows_fma(rows, prog.flush)
    return rows.view(x.shape), logdet.view(batch)
