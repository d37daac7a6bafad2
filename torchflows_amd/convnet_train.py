"""The ConvNet conditioner of the image couplings with gradients, on libtfk (csrc/tfk_convtrain.hip).

Reference ``multiscale/conditioning/classic.py:45-122``: ConvModifier -> 3 x [conv3x3 -> ReLU -> MaxPool2d(2) ->
BatchNorm2d] -> ConvModifier -> Linear.  ``Flow.fit`` runs it in training mode (flows.py:333), i.e. every BatchNorm
normalises with the statistics of the batch and updates its running ones: a block's output depends on all samples, so the
unit of fusion is the block, not the coupling (``image_program`` fuses a whole coupling only because inference folds
BatchNorm into a per-channel scale / shift).  One ``torch.autograd.Function`` covers the network:

  forward   frame (ConvModifier) | block 1 | block 2 | block 3 | frame (+ BatchNorm 3 on load) | fold | linear    7 launches
  backward  linear input gradient | linear weight gradient | frame_bwd | block_bwd x 3 | frame_bwd              7 launches
            (+ one dot product for the second modifier's bias)

against ~70 ATen / MIOpen launches, issued by ONE libtfk call per pass (``tfk_convnet_train_forward`` / ``_backward``).
The Linear layer sees the second modifier's bias on 84 of its 100 inputs: it is evaluated as a 16-term product with an
effective bias (``linear_prep``), as ``image_program`` does at inference.  Each block launch writes the pooled activation BEFORE normalisation and the arg-max
byte of every pooling window; the normalisation is applied by whoever reads it next.  The batch sums (statistics in the
forward; weight gradients and the two sums a BatchNorm backward needs in the reverse pass) are fixed-order sums of
per-workgroup partials finished inside the same launch.

Only the reference's own network shape is covered (``structure_ok``); anything else keeps the ATen composite path.
"""
from __future__ import annotations

import contextlib
import threading
from typing import List, Optional

import torch
import torch.nn as nn

from torchflows_amd import native

_state = threading.local()


@contextlib.contextmanager
def recomputing():
    """Inside: a forward pass that REPEATS one already made on the same batch (``autograd.ChainFunction.backward``
    re-evaluates the conditioner to differentiate it).  BatchNorm layers in training mode normalise with the batch
    statistics as before but leave their running statistics alone -- the first pass has counted this batch."""
    prev = getattr(_state, "recompute", False)
    _state.recompute = True
    try:
        yield
    finally:
        _state.recompute = prev


def is_recomputing() -> bool:
    return getattr(_state, "recompute", False)


def enabled() -> bool:
    from torchflows_amd.utils import debug_switch
    return debug_switch("convnet_train", "1") != "0"


def _modifier_ok(mod, c_out: int) -> bool:
    conv = mod.conv
    kh, kw = conv.weight.shape[2:]
    return (kh in (1, 2) and kw in (1, 2) and conv.weight.shape[0] == c_out and conv.bias is not None
            and tuple(conv.stride) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and conv.padding[0] >= kh - 1 and conv.padding[1] >= kw - 1 and conv.padding_mode == "zeros")


def structure_ok(net) -> bool:
    """The reference's default ConvNet: (c, h, w) -> modifier (a 1- or 2-wide kernel per axis) to (4, 32, 32) -> blocks
    4->8->8->4 with pooling -> 1x1 modifier to (1, 10, 10) -> Linear(100, n); cached on the module (the structure does
    not change)."""
    hit = net.__dict__.get("_tfk_ct_structure")
    if hit is not None:
        return hit
    ok = False
    try:
        blocks = list(net.blocks)
        if len(blocks) == 5 and isinstance(net.linear, nn.Linear) and net.linear.bias is not None:
            first, last = blocks[0], blocks[4]
            chans = [(4, 8, 32), (8, 8, 16), (8, 4, 8)]
            ok = (_modifier_ok(first, 4) and tuple(first.output_shape) == (4, 32, 32)
                  and _modifier_ok(last, 1) and tuple(last.output_shape) == (1, 10, 10)
                  and tuple(last.conv.weight.shape[1:]) == (4, 1, 1) and net.linear.in_features == 100)
            for blk, (ci, co, h) in zip(blocks[1:4], chans):
                bn = blk.bn
                ok = ok and (isinstance(blk.pool, nn.MaxPool2d) and blk.conv.in_channels == ci
                             and blk.conv.out_channels == co and tuple(blk.conv.kernel_size) == (3, 3)
                             and tuple(blk.conv.padding) == (1, 1) and blk.conv.bias is not None
                             and isinstance(bn, nn.BatchNorm2d) and bn.affine and bn.track_running_stats
                             and bn.momentum is not None
                             and bool(native.lib().tfk_convnet_train_block_supported(ci, co, h)))
    except AttributeError:          # (another module tree; a missing or mismatched libtfk raises NativeError: loud)
        ok = False
    net.__dict__["_tfk_ct_structure"] = ok
    return ok


def _params(net) -> List[torch.Tensor]:
    """The 18 parameter tensors in the order ConvNetFunction takes them.  The (``_parameters`` dict, name) slots are looked
    up once per network (23 000 ``Module.__getattr__`` calls per training step of config 5's model otherwise: 1 ms); a
    replaced Parameter is still seen, a replaced submodule is not -- ``_tfk_`` caches are dropped by fused.invalidate."""
    slots = net.__dict__.get("_tfk_ct_slots")
    if slots is None:
        b = net.blocks
        mods = [b[0].conv, b[0].conv]
        names = ["weight", "bias"]
        for blk in (b[1], b[2], b[3]):
            mods += [blk.conv, blk.conv, blk.bn, blk.bn]
            names += ["weight", "bias", "weight", "bias"]
        mods += [b[4].conv, b[4].conv, net.linear, net.linear]
        names += ["weight", "bias", "weight", "bias"]
        slots = net.__dict__["_tfk_ct_slots"] = ([(m._parameters, n) for m, n in zip(mods, names)],
                                                 (b[1].bn, b[2].bn, b[3].bn))
    return [d[n] for d, n in slots[0]]


def _bns(net):
    _params(net)
    return net.__dict__["_tfk_ct_slots"][1]


def _frame_ok(net, c: int, h: int, w: int) -> bool:
    kh, kw = net.blocks[0].conv.weight.shape[2:]
    return not (h > 32 or w > 32 or (32 - h + kh - 1) % 2 or (32 - w + kw - 1) % 2
                or net.blocks[0].conv.weight.shape[1] != c or (4 * kh * kw + 2) * c + 4 > 640
                or (4 * (h + 1) * (w + 1) + 2 * c * h * w) * 4 > 120 * 1024)


def static_usable(net, device) -> bool:
    """``usable`` for every batch the network will see on ``device``, judged from the module alone (the input frame
    follows from the first modifier's kernel and padding): what ``Flow.fit`` asks before it captures a training step."""
    if not enabled() or not structure_ok(net):
        return False
    conv = net.blocks[0].conv
    kh, kw = conv.weight.shape[2:]
    h, w = 32 - 2 * conv.padding[0] + kh - 1, 32 - 2 * conv.padding[1] + kw - 1
    if h < 1 or w < 1 or not _frame_ok(net, conv.weight.shape[1], h, w):
        return False
    return all(p.device == device and p.dtype == torch.float32 and p.is_contiguous() for p in _params(net))


def usable(net, x: torch.Tensor) -> bool:
    """fp32 on a HIP device, the default network, an input frame the 1x1 modifier covers, and either a gradient to
    compute or batch statistics to take (training mode)."""
    if not enabled() or x.dim() != 4 or x.device.type != "cuda" or x.dtype != torch.float32 or x.shape[0] == 0:
        return False
    if not structure_ok(net):
        return False
    _, c, h, w = x.shape
    if not _frame_ok(net, c, h, w):
        return False
    params = _params(net)
    if any(p.device != x.device or p.dtype != torch.float32 or not p.is_contiguous() for p in params):
        return False                    # (the one-call route hands raw pointers over)
    bns = _bns(net)
    training = bns[0].training
    if bns[1].training != training or bns[2].training != training:
        return False
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    return needs_grad or training


def apply(net, x: torch.Tensor) -> torch.Tensor:
    """theta (N, n_outputs) of ``net`` on images x (N, c, h, w)."""
    training = _bns(net)[0].training
    update = training and not is_recomputing()
    return ConvNetFunction.apply(net, training, update, x.contiguous(), *_params(net))


class ConvNetFunction(torch.autograd.Function):
    """One call into libtfk per pass (tfk_convnet_train_forward / _backward: the launches listed at the top, in sequence);
    ``TORCHFLOWS_AMD_DEBUG=convnet_calls=each`` issues them one by one from Python instead (the per-launch entry points
    the tests and tools/convtrain_bench.py also use)."""

    @staticmethod
    def forward(ctx, net, training: bool, update: bool, x: torch.Tensor, *params: torch.Tensor):
        from torchflows_amd.utils import debug_switch
        params = [p.detach() for p in params]
        ctx.training = training
        if debug_switch("convnet_calls", "one") == "each":
            return _forward_each(ctx, net, training, update, x, params)
        plan = native.ConvNetTrainPlan()
        bns = _bns(net)
        theta, acts, amax = native.convnet_train_forward(plan, params, bns, x, training, update)
        ctx.plan = plan
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(x, acts, amax, *params)          # (the parameters: their version counters guard the pointers)
        return theta

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_theta: torch.Tensor):
        need = ctx.needs_input_grad
        if getattr(ctx, "plan", None) is None:
            g_x, grads = _backward_each(ctx, g_theta)
        else:
            x = ctx.saved_tensors[0]            # (also checks that no saved tensor was modified in place)
            p, sh = ctx.plan, ctx.shapes
            g_x, bn, sums = native.convnet_train_backward(p, x, g_theta.contiguous(), ctx.training)
            M, n_m1 = p.M, 4 * p.c * p.kh * p.kw
            o1 = n_m1 + 2 * p.c + 4
            o2, o3, o4 = o1 + 296, o1 + 296 + 584, o1 + 296 + 584 + 292
            oL = o4 + 13
            grads = [sums[:n_m1].view(sh[0]), sums[o1 - 4:o1],
                     sums[o1:o1 + 288].view(sh[2]), sums[o1 + 288:o2], bn[24:32], bn[32:40],
                     sums[o2:o2 + 576].view(sh[6]), sums[o2 + 576:o3], bn[64:72], bn[72:80],
                     sums[o3:o3 + 288].view(sh[10]), sums[o3 + 288:o4], bn[92:96], bn[96:100],
                     sums[o4:o4 + 4].view(sh[14]), sums[o4 + 12:o4 + 13],
                     sums[oL:oL + 100 * M].view(sh[16]), sums[oL + 100 * M:oL + 101 * M]]
        return (None, None, None, g_x if need[3] else None,
                *[g if need[4 + k] else None for k, g in enumerate(grads)])


def _forward_each(ctx, net, training, update, x, params):
    (w_m1, b_m1, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, g3, be3, w_m2, b_m2, w_lin, b_lin) = params
    N = x.shape[0]
    blocks = net.blocks
    a0 = native.convnet_train_frame_fwd(x, None, w_m1, b_m1, 32, 32)
    y1, i1, s1 = native.convnet_train_block_fwd(a0, None, w1, b1, blocks[1].bn, training, update)
    y2, i2, s2 = native.convnet_train_block_fwd(y1, s1, w2, b2, blocks[2].bn, training, update)
    y3, i3, s3 = native.convnet_train_block_fwd(y2, s2, w3, b3, blocks[3].bn, training, update)
    a16 = native.convnet_train_frame_fwd(y3, s3, w_m2, b_m2, 4, 4).view(N, 16)     # (the frame around it == b_m2)
    W16, b_eff, w_frame = native.convnet_train_linear_prep(w_lin, b_lin, b_m2, 10, 10)
    theta = native.convnet_train_linear_fwd(a16, W16, b_eff)
    ctx.plan = None
    ctx.save_for_backward(x, a0, y1, i1, s1, y2, i2, s2, y3, i3, s3, a16, w_m1, w1, w2, w3, w_m2, b_m2, W16, w_frame)
    return theta


def _backward_each(ctx, g_theta):
    (x, a0, y1, i1, s1, y2, i2, s2, y3, i3, s3, a16, w_m1, w1, w2, w3, w_m2, b_m2, W16, w_frame) = ctx.saved_tensors
    tr = ctx.training
    N = x.shape[0]
    g_theta = g_theta.contiguous()
    g16 = native.convnet_train_linear_bwd_input(g_theta, W16)             # (N, 16): the interior of d/d(Linear input)
    dW_lin, db_lin = native.convnet_train_linear_wgrad(g_theta, a16, b_m2, 10, 10)
    gz3, dw_m2, db_m2, bn3 = native.convnet_train_frame_bwd(g16.view(N, 1, 4, 4), y3, s3, w_m2, s3, tr)
    db_m2 = db_m2 + torch.dot(db_lin, w_frame)                            # (+ the frame: 84 inputs that equal b_m2)
    gz2, dW3, db3, bn2 = native.convnet_train_block_bwd(gz3, bn3[0], y3, i3, y2, s2, w3, s2, tr)
    gz1, dW2, db2, bn1 = native.convnet_train_block_bwd(gz2, bn2[0], y2, i2, y1, s1, w2, s1, tr)
    g_a0, dW1, db1, _ = native.convnet_train_block_bwd(gz1, bn1[0], y1, i1, a0, None, w1, None, tr)
    g_x, dw_m1, db_m1, _ = native.convnet_train_frame_bwd(g_a0, x, None, w_m1, None, tr)
    return g_x, [dw_m1, db_m1,
                 dW1, db1, bn1[1], bn1[2],
                 dW2, db2, bn2[1], bn2[2],
                 dW3, db3, bn3[1], bn3[2],
                 dw_m2, db_m2, dW_lin, db_lin]
