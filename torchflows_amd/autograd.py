"""Training on the HIP path: one autograd node per composition.

The reference trains through ``torch.autograd`` over its per-layer ATen graphs
(``Flow.fit`` flows.py:226-455 -> ``log_prob`` :628-658 -> ``BijectiveComposition.forward``
bijections/base.py:203-224).  Here a whole composition is ONE ``torch.autograd.Function``:

* forward: the same libtfk layer kernels as inference, out of place, keeping only the
  (N, D) rows that entered each coupling / trainable elementwise layer.  The conditioner
  output ``h`` (2.9 KB per row per RQ-spline layer) is NOT kept.
* backward: walk the layers in reverse over one gradient row buffer ``g`` (N, D) and the
  per-row log-det gradient; per layer one reverse-mode kernel of csrc/tfk_bwd.hip, which
  recomputes alpha / knots / bins from the saved input rows and a re-evaluated ``h``
  ("recompute, not store").  The conditioner MLP's own backward (two skinny GEMMs) is
  ``torch.autograd.grad`` on PyTorch-ROCm, fed with the kernel's ``dL/dh``.

Only what the kernels cover takes this route (affine / inverse-affine / shift / RQ-spline
couplings with 4, 8 or 16 bins, ElementwiseAffine / ActNorm, permutations, no context);
anything else keeps the ATen composite graph.  ``TORCHFLOWS_AMD_TRAIN=0`` forces the latter.
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import List, Optional, Tuple

import torch

from torchflows_amd.utils import debug_switch

from torchflows_amd import native

FORWARD, INVERSE = 0, 1


def enabled() -> bool:
    return os.environ.get("TORCHFLOWS_AMD_TRAIN", "1") != "0"


_live = threading.local()


@contextlib.contextmanager
def live_route():
    """Inside: an evaluation WITHOUT gradients takes the training route too (the chain's per-layer launches, operands
    gathered from the live parameter tensors) instead of the packed flow programs.  A packed program is a copy of the
    weights, refreshed when a parameter's version counter moves -- which a replayed hipGraph never does; the launches of
    this route read the parameters themselves, so they can sit in a captured validation pass (Flow.fit)."""
    prev = getattr(_live, "on", False)
    _live.on = True
    try:
        yield
    finally:
        _live.on = prev


def live_forced() -> bool:
    return getattr(_live, "on", False)


def _flatten(layers, attr: str):
    from torchflows_amd.fused import _flatten as flat
    return flat(layers, attr)


def _step_kind(layer) -> Optional[str]:
    from torchflows_amd.bijections.finite.autoregressive.layers_base import (
        CouplingBijection, ElementwiseBijection, MaskedAutoregressiveBijection)
    from torchflows_amd.bijections.finite.matrix.permutation import PermutationMatrix
    if isinstance(layer, PermutationMatrix):
        return "perm"
    if isinstance(layer, MaskedAutoregressiveBijection):
        # the parallel pass (MAF density / IAF sampling): MADE on PyTorch-ROCm, the transformer's
        # forward and reverse-mode kernels with every position a target
        if layer.context_shape is not None:
            return None
        kind = layer.transformer.native_kind
        if kind in ("affine", "inverse_affine"):
            return "made"
        if kind == "rqs" and native.lib().tfk_rqs_coupling_bwd_supported(int(layer.transformer.n_bins)):
            return "made"
        if kind == "lrs" and int(layer.transformer.n_bins) in (4, 8):
            return "made"
        return None
    if isinstance(layer, ElementwiseBijection):
        if layer.transformer.native_kind not in ("affine", "inverse_affine"):
            return None
        # with a context the parameters come from a conditioner on it (layers_base.py:300-318): the affine coupling
        # kernels with every position a target, the conditioner (default: Linear) on PyTorch-ROCm
        return "elementwise" if layer.use_global_parameters else "elementwise_ctx"
    if isinstance(layer, CouplingBijection):
        # (a context only enters the conditioner, by concatenation: conditioning/context.py:38-64)
        kind = layer.transformer.native_kind
        if kind in ("affine", "inverse_affine", "shift"):
            return "coupling"
        if kind == "rqs" and native.lib().tfk_rqs_coupling_bwd_supported(int(layer.transformer.n_bins)):
            return "coupling"
        if kind == "lrs" and int(layer.transformer.n_bins) in (4, 8):
            return "coupling"
        if kind == "conv1x1" and int(layer.transformer.n_channels) <= 16:
            return "coupling"           # Glow's invertible 1x1 convolution (linear/convolution.py:33-64)
    return None


def training_plan(composition, direction: int):
    """[(layer, direction, kind)] if every layer has forward and reverse-mode kernels, else None."""
    # the steps depend on the layer objects and their configuration only: kept on the composition per direction, keyed
    # by the identity of its layers (dropped with every other ``_tfk_`` cache on train() / eval() / load_state_dict /
    # a move); 27 layers x isinstance chains cost 60 us of a 1.5 ms step
    from torchflows_amd import fused
    key = (fused._EPOCH[0], tuple(map(id, composition.layers)))
    cache = composition.__dict__.setdefault("_tfk_train_plan", {})
    hit = cache.get(direction)
    if hit is not None and hit[0] == key:
        return None if hit[1] is None else Plan(hit[1])
    order = composition.layers if direction == FORWARD else list(composition.layers)[::-1]
    flat = _flatten(order, "forward" if direction == FORWARD else "inverse")
    steps = None
    if flat is not None:
        steps = []
        for layer, d in flat:
            kind = _step_kind(layer)
            if kind is None or (kind == "made" and d == layer._sequential_when):
                steps = None    # (the element-by-element map is not differentiated on the HIP path)
                break
            steps.append((layer, d, kind))
    cache[direction] = (key, steps)
    return None if steps is None else Plan(steps)


class Plan(list):
    """[(layer, direction, kind)] + the context rows (N, *context_shape) of this call, or None; ``l2``: the
    {coefficient: [parameters]} of the L2 penalty the caller wants evaluated in the same autograd node
    (``Flow._base_batch_loss``), or None."""
    context = None
    l2 = None


def applicable(composition, x: torch.Tensor, context) -> bool:
    """Autograd is on, something needs a gradient, and everything is fp32 on one HIP device."""
    if not enabled() or not (torch.is_grad_enabled() or live_forced()):
        return False
    if x.device.type != "cuda" or x.dtype != torch.float32:
        return False
    if context is not None and (context.device != x.device or context.dtype != torch.float32 or context.requires_grad):
        return False            # (a gradient with respect to the context: the ATen graph)
    from torchflows_amd import fused
    if not fused.static_ok(composition):
        return False
    first = next(composition.parameters(), None)
    if first is not None and first.device != x.device:
        return False
    return x.requires_grad or fused.any_requires_grad(composition)


def _module_params(module) -> List[torch.Tensor]:
    """``list(module.parameters())`` without the module walk on every call (27 layers x forward + backward = 0.3 ms of a
    3 ms training step): the (submodule, name) slots are walked once and looked up by name afterwards, so a replaced
    Parameter is seen; dropped by fused.invalidate (train() / eval() / load_state_dict / invalidate_native_caches)."""
    slots = module.__dict__.get("_tfk_param_slots")
    if slots is None:
        seen, slots = set(), []
        for m in module.modules():
            for n, p in m._parameters.items():
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    slots.append((m, n))
        module.__dict__["_tfk_param_slots"] = slots
    return [m._parameters[n] for m, n in slots]


def _layer_params(layer, kind: str) -> List[torch.Tensor]:
    if kind == "elementwise":
        return [layer.value]
    if kind in ("coupling", "made", "elementwise_ctx"):
        return _module_params(layer.conditioner_transform)
    return []


def _conditioner(layer, x_in: torch.Tensor, context=None) -> torch.Tensor:
    """h (N, T*P) from the rows that enter the coupling (layers_base.py:117-143)."""
    N = x_in.shape[0]
    S = layer.coupling.source_event_size
    x_a = x_in[:, :S] if layer._source_is_head else x_in.index_select(1, layer._source_index)
    return layer.conditioner_transform(x_a.reshape(N, *layer.coupling.constant_shape), context=_layer_context(layer, context))


def _layer_context(layer, context):
    """The call's context rows for a layer that takes one, else None."""
    return context if (context is not None and layer.context_shape is not None) else None


def _keeps_graph(layer) -> bool:
    """Couplings whose conditioner is evaluated once per step with its autograd graph kept for the backward pass,
    instead of being re-evaluated there: the convolutional conditioners of the image flows."""
    from torchflows_amd.bijections.finite.multiscale.conditioning.classic import ConvNetConditioner
    return isinstance(layer.conditioner_transform, ConvNetConditioner)


def _plain_mlp(layer):
    """(Linear, Linear) when the conditioner is the default FeedForward: Linear, Tanh, Linear on
    x_A alone (transforms.py:274-307), no global parameters, unbounded output -- the case whose
    backward is written out below instead of being left to torch.autograd."""
    import math
    import torch.nn as nn
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import FeedForward
    ct = layer.conditioner_transform
    if type(ct) is not FeedForward or ct.n_global_parameters != 0 or ct.context_shape is not None:
        return None
    if ct.output_lower_bound != -math.inf or ct.output_upper_bound != math.inf:
        return None
    mods = list(ct.sequential)
    if (len(mods) == 4 and isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.Tanh)
            and isinstance(mods[2], nn.Linear) and isinstance(mods[3], nn.Unflatten)):
        return mods[0], mods[2]
    return None


class _TrainPack:
    """Index maps between a coupling layer's (W1, b1, W2, b2) and the operand / accumulator layouts
    of tfk_affine_coupling_train_bwd (csrc/tfk_bwd.hip), built once per (D, H, device).
    Lane l = (q = l >> 4, i = l & 15); hidden unit of D-row i: u(i) = 4 (i & 3) + (i >> 2)."""

    _cache = {}

    @classmethod
    def get(cls, D: int, H: int, device, D_log: Optional[int] = None, shift: bool = False) -> "_TrainPack":
        key = (D, H, str(device), D_log or D, shift)
        if key not in cls._cache:
            cls._cache[key] = cls(D, H, device, D_log or D, shift)
        return cls._cache[key]

    def __init__(self, D: int, H: int, device, D_log: Optional[int] = None, shift: bool = False):
        """``D``: the row width the kernels see; ``D_log`` <= D: the layer's event size when its rows are padded
        (training_layout: the source half at the head of plane A, the target half at the TAIL of plane B, zeros between) --
        the weights of the padding are the appended zero, its accumulators are not mapped back.
        ``shift``: a Shift coupling (NICE, affine.py:137-159) run as the affine coupling it is with a scale logit of
        zero -- alpha = exp(0 / 2 + c0) + 1e-10 = 1 exactly, log alpha = 0: the conditioner's one output per element is the
        shift row of GEMM 2, the logit rows are the appended zero, and their accumulators are not mapped back."""
        D_log = D_log or D
        half, EPL = D // 2, D // 8
        h = D_log // 2                                    # sources (HalfSplit: the first D // 2 elements, coupling_masks)
        ht = D_log - h                                    # targets: one more when the event size is odd
        pad = half - ht                                   # plane B: padding first, then the targets
        T2, T1 = EPL // 2, EPL // 4
        self.steps2 = (H + 3) // 4
        TP = ht if shift else 2 * ht                      # rows of the real W2 / b2
        off_b1 = H * h
        off_W2 = off_b1 + H
        off_b2 = off_W2 + TP * H
        Z = off_b2 + TP                                   # index of the appended zero
        ar = torch.arange
        W1idx = torch.full((16, half), Z, dtype=torch.long)
        W1idx[:H, :h] = (ar(H)[:, None] * h + ar(h)[None, :])
        b1idx = torch.full((16,), Z, dtype=torch.long)
        b1idx[:H] = off_b1 + ar(H)
        W2idx = torch.full((half, 2, 16), Z, dtype=torch.long)
        b2idx = torch.full((half, 2), Z, dtype=torch.long)
        if shift:
            W2idx[pad:, 1, :H] = off_W2 + (ar(ht)[:, None] * H + ar(H)[None, :])
            b2idx[pad:, 1] = off_b2 + ar(ht)
        else:
            W2idx[pad:, :, :H] = off_W2 + ((ar(ht)[:, None, None] * 2 + ar(2)[None, :, None]) * H + ar(H)[None, None, :])
            b2idx[pad:] = off_b2 + (ar(ht)[:, None] * 2 + ar(2)[None, :])
        lane = ar(64)
        ql, il = lane >> 4, lane & 15
        unit = 4 * (il & 3) + (il >> 2)
        q2, r2 = il >> 2, il & 3
        qq, rr = torch.meshgrid(ar(4), ar(4), indexing="ij")
        parts = [torch.stack([W1idx[unit, EPL * ql + s] for s in range(EPL)]).reshape(-1),        # A1
                 b1idx[4 * rr + qq].reshape(-1)]                                                  # b1[q][r]
        A2, b2m, A2T, A1T = [], [], [], []
        for t in range(T2):
            for r1 in range(self.steps2):
                A2.append(W2idx[EPL * q2 + 2 * t + (r2 >> 1), r2 & 1, 4 * r1 + ql])
            b2m.append(b2idx[EPL * qq + 2 * t + (rr >> 1), rr & 1].reshape(-1))
            for r in range(4):
                A2T.append(W2idx[EPL * ql + 2 * t + (r >> 1), r & 1, unit])
        for t in range(T1):
            for r in range(4):
                A1T.append(W1idx[4 * r + ql, EPL * (il >> 2) + 4 * t + (il & 3)])
        parts += [torch.stack(A2).reshape(-1), torch.stack(b2m).reshape(-1),
                  torch.stack(A2T).reshape(-1), torch.stack(A1T).reshape(-1)]
        self.param_index = torch.cat(parts).to(device)
        self.n_flat = Z + 1
        # accumulator layout -> [dW1 (H, half) | db1 (H) | dW2 (TP, H) | db2 (TP)]
        u = ar(H)
        e = ar(h)                                         # (physical = logical source element)
        off1 = T2 * 256
        dW1 = off1 + (((e // 16)[None, :] * 64 + 16 * ((e % 16) // 4)[None, :] + u[:, None]) * 4 + (e % 4)[None, :])
        db1 = off1 + T1 * 256 + 4 * (u % 4) + (u // 4)
        if shift:
            m, pbit = pad + ar(ht), torch.ones(ht, dtype=torch.long)  # (the shift rows only)
        else:
            m = (pad + ar(ht))[:, None].expand(ht, 2).reshape(-1)    # physical target element of logical target t
            pbit = ar(2)[None, :].expand(ht, 2).reshape(-1)
        q_m, rem = m // EPL, m % EPL
        T_m, r_m = rem // 2, 2 * (rem % 2) + pbit
        dW2 = ((T_m * 64 + 16 * q_m)[:, None] + u[None, :]) * 4 + r_m[:, None]
        db2 = (T_m * 64 + 16 * q_m + 15) * 4 + r_m
        self.grad_index = torch.cat([dW1.reshape(-1), db1, dW2.reshape(-1), db2]).to(device)
        self.sizes = (H * h, H, TP * H, TP)
        self.shapes = ((H, h), (H,), (TP, H), (TP,))
        self.zero = torch.zeros(1, dtype=torch.float32, device=device)
        n_out = int(native.lib().tfk_coupling_train_bwd_out_floats(D))
        self.n_out = n_out
        self.workspace = torch.empty(int(native.lib().tfk_coupling_train_bwd_workspace_bytes(D)) // 4,
                                     dtype=torch.float32, device=device)


class _RqsTrainPack:
    """Index maps for tfk_rqs_coupling_train_bwd (D = 64, 8 bins, hidden width H <= 16): operand
    block from (W1, b1, W2, b2), and the un-permutations of its accumulator-order outputs."""

    _cache = {}

    @classmethod
    def get(cls, H: int, device, D_log: int = 64) -> "_RqsTrainPack":
        key = (H, str(device), D_log)
        if key not in cls._cache:
            cls._cache[key] = cls(H, device, D_log)
        return cls._cache[key]

    def __init__(self, H: int, device, D_log: int = 64):
        """``D_log`` < 64: the layer's event size on rows in the padded training layout (its h = D_log / 2 sources at the
        head of plane A, its h targets at the tail of plane B); a padding element's 23 spline parameters are the appended
        zero -- equal bins, unit derivatives: the identity at its value 0 (a knot)."""
        half, EPL, P = 32, 8, 23
        h = D_log // 2                                    # sources
        ht = D_log - h                                    # targets (one more when the event size is odd)
        pad = half - ht
        self.H = H
        self.steps2 = (H + 3) // 4
        off_b1 = H * h
        off_W2 = off_b1 + H
        off_b2 = off_W2 + ht * P * H
        Z = off_b2 + ht * P
        ar = torch.arange
        W1idx = torch.full((16, half), Z, dtype=torch.long)
        W1idx[:H, :h] = ar(H)[:, None] * h + ar(h)[None, :]
        b1idx = torch.full((16,), Z, dtype=torch.long)
        b1idx[:H] = off_b1 + ar(H)
        W2idx = torch.full((half, 24, 16), Z, dtype=torch.long)
        W2idx[pad:, :P, :H] = off_W2 + ((ar(ht)[:, None, None] * P + ar(P)[None, :, None]) * H + ar(H)[None, None, :])
        b2idx = torch.full((half, 24), Z, dtype=torch.long)
        b2idx[pad:, :P] = off_b2 + ar(ht)[:, None] * P + ar(P)[None, :]
        lane = ar(64)
        ql, il = lane >> 4, lane & 15
        unit = 4 * (il & 3) + (il >> 2)
        q2, r2 = il >> 2, il & 3
        qq, rr = torch.meshgrid(ar(4), ar(4), indexing="ij")
        A1 = torch.stack([W1idx[unit, EPL * ql + s] for s in range(EPL)])
        b1m = b1idx[4 * rr + qq]
        A2, b2m, A2T = [], [], []
        for e in range(EPL):
            for c in range(6):
                for r1 in range(self.steps2):
                    A2.append(W2idx[EPL * q2 + e, 4 * c + r2, 4 * r1 + ql])
                b2m.append(b2idx[EPL * qq + e, 4 * c + rr].reshape(-1))
                for r in range(4):
                    A2T.append(W2idx[EPL * ql + e, 4 * c + r, unit])
        A1T = [W1idx[4 * r + ql, EPL * (il >> 2) + 4 * t + (il & 3)] for t in range(2) for r in range(4)]
        self.param_index = torch.cat([A1.reshape(-1), b1m.reshape(-1), torch.stack(A2).reshape(-1),
                                      torch.stack(b2m).reshape(-1), torch.stack(A2T).reshape(-1),
                                      torch.stack(A1T).reshape(-1)]).to(device)
        self.n_fwd = EPL * 64 + 16 + 48 * self.steps2 * 64 + 48 * 16
        # column of gh_perm that holds parameter p of (physical) target element m = pad + logical target
        m = (pad + ar(ht))[:, None]
        pp = ar(P)[None, :]
        e_, q_ = m % EPL, m // EPL
        self.gh_col = ((6 * e_ + pp // 4) * 16 + 4 * q_ + pp % 4).reshape(-1).to(device)     # (736,)
        u = ar(H)
        self.gpre_col = (4 * (u % 4) + u // 4).to(device)                                   # (H,)
        self.zero = torch.zeros(1, dtype=torch.float32, device=device)
        # tfk_rows_outer route (hidden width <= 15): three products in accumulator order, one buffer
        #   [gh_perm^T hid_perm (768 x 16) | x_A^T gpre_perm (32 x 16) | gpre_perm^T hid_perm (16 x 16)]
        # and ONE gather to [dW1 (H, 32) | db1 (H) | dW2 (736, H) | db2 (736)]
        self.n_acc = 768 * 16 + 32 * 16 + 16 * 16
        if H <= 15:
            slot = 4 * (u % 4) + u // 4                                   # column of hid_perm / gpre_perm that holds unit u
            c = ((6 * e_ + pp // 4) * 16 + 4 * q_ + pp % 4).reshape(-1)  # gh_perm column of (target m, parameter p)
            t1 = 4 * (c // 64) + (c % 64) % 4                             # its tile and M-index (include/tfk.h)
            i1 = (c % 64) // 4
            idx1 = lambda j: ((t1 * 64 + 16 * (i1 // 4))[:, None] + j[None, :]) * 4 + (i1 % 4)[:, None]
            dW2 = idx1(slot)                                              # (736, H)
            db2 = idx1(torch.tensor([15]))[:, 0]
            e = ar(h)
            base2 = 768 * 16
            dW1 = base2 + (((e // 16) * 64 + 16 * ((e % 16) // 4))[None, :] + slot[:, None]) * 4 + ((e % 16) % 4)[None, :]
            base3 = base2 + 32 * 16
            db1 = base3 + (16 * (slot // 4) + 15) * 4 + slot % 4
            self.acc_index = torch.cat([dW1.reshape(-1), db1, dW2.reshape(-1), db2]).to(device)
            self.acc_sizes = (H * h, H, ht * P * H, ht * P)
            self.acc_shapes = ((H, h), (H,), (ht * P, H), (ht * P,))

    def pack(self, lin1, lin2) -> torch.Tensor:
        flat = torch.cat([lin1.weight.detach().reshape(-1), lin1.bias.detach(),
                          lin2.weight.detach().reshape(-1), lin2.bias.detach(), self.zero])
        return flat[self.param_index]


def _fused_rqs_layer(layer, D: int):
    """(lin1, lin2) when the layer can use tfk_rqs_coupling_train_bwd (and the RQS flow-program op)."""
    if not fused_train_enabled() or layer.transformer.native_kind != "rqs":
        return None
    lib, K = native.lib(), int(layer.transformer.n_bins)
    if not (lib.tfk_rqs_coupling_train_bwd_supported(D, K)
            or (padded_train_enabled() and 3 <= D < 64 and lib.tfk_rqs_coupling_train_bwd_supported(64, K))):
        return None                           # (D < 64: on rows padded to 64 -- whether the plan may be is plan_width's call)
    c = layer.coupling
    if not (layer._source_is_head and layer._target_is_tail and c.source_event_size == D // 2
            and c.target_event_size == D - D // 2):
        return None
    mlp = _plain_mlp(layer)
    if mlp is None or mlp[0].out_features > 16:
        return None
    return mlp


def flat_enabled() -> bool:
    """Operands gathered from / gradients returned as slices of ONE buffer when the parameters live in one
    (torchflows_amd/flat_optim.py; TORCHFLOWS_AMD_DEBUG=flat=0: always the per-tensor route)."""
    return debug_switch("flat", "1") != "0"


def rows_outer_enabled() -> bool:
    """The weight-gradient products of the fused spline training step on tfk_rows_outer (TORCHFLOWS_AMD_DEBUG=rows_outer=0:
    split-K batched GEMMs on the GEMM library, as before round 3)."""
    return debug_switch("rows_outer", "1") != "0"


def fused_train_enabled() -> bool:
    return debug_switch("train_fused", "1") != "0"


def padded_train_enabled() -> bool:
    """Fused training launches for even event sizes below 128 that are not 64 / 128, on rows padded to the next of the
    two (TORCHFLOWS_AMD_DEBUG=train_pad=0: the layer-by-layer reverse mode, as before round 3)."""
    return debug_switch("train_pad", "1") != "0"


def train_width(D: int) -> Optional[int]:
    """Row width the fused training launches run an event size of ``D`` at, or None."""
    if native.lib().tfk_coupling_train_bwd_supported(D):
        return D
    if padded_train_enabled() and 3 <= D < 128:          # (odd sizes: D - D // 2 targets fit a plane of W / 2 as well)
        return 64 if D < 64 else 128
    return None


def _padding_capable(plan) -> bool:
    """Every step keeps the padded training layout (source half at the head of plane A, target half at the tail of
    plane B): reversals -- a logical reversal IS the physical one there --, global elementwise layers, and couplings
    with the fused launches."""
    D = plan[0][0].n_dim if len(plan) else 0
    for layer, d, kind in plan:
        if kind == "perm" and layer._is_reversal:
            continue
        if kind == "elementwise":
            continue
        if kind == "coupling" and _fused_bwd_layer(layer, D) is not None:
            continue
        if kind == "coupling" and D < 64 and _fused_rqs_layer(layer, D) is not None:
            continue                          # (the fused spline launches exist at 64 columns only)
        return False
    return True


def plan_width(plan, D: int) -> int:
    """The width of the rows ChainFunction runs this plan on: ``D``, or the padded width when D itself has no fused
    training launches and every step keeps the padded layout."""
    W = train_width(D)
    if W is None or W == D or not _padding_capable(plan):
        return D
    return W


def training_layout(D: int, W: int, device) -> torch.Tensor:
    """Physical column of logical element l on rows padded from D to W: l for the first D // 2 (the sources of a
    HalfSplit coupling), W - D + l for the others (its targets) -- for even D reversing the D logical columns is then
    reversing the W physical ones; for odd D it is not (the middle element is a target before AND after a reversal, so
    it has to change planes): those plans run their reversals as the general column gather ``reversal_index``."""
    key = (D, W, str(device))
    hit = _LAYOUTS.get(key)
    if hit is None:
        l = torch.arange(D, device=device)
        hit = _LAYOUTS[key] = torch.where(l < D // 2, l, l + (W - D))
    return hit


_LAYOUTS = {}


def reversal_index(D: int, W: int, device) -> torch.Tensor:
    """int32 (W,): out[:, c] = in[:, index[c]] reverses the D logical columns of rows in the padded training layout and
    leaves the padding where it is (an involution: the backward pass uses the same index)."""
    key = (D, W, str(device), "rev")
    hit = _LAYOUTS.get(key)
    if hit is None:
        phys = training_layout(D, W, "cpu")
        idx = torch.arange(W)
        idx[phys] = phys.flip(0)                     # the column of logical j takes the column of logical D - 1 - j
        hit = _LAYOUTS[key] = idx.to(torch.int32).to(device)
    return hit


def _fused_bwd_layer(layer, D: int):
    """(lin1, lin2) when the layer's whole backward can run as tfk_affine_coupling_train_bwd (at the width
    ``train_width(D)``: whether a plan may be padded is ``plan_width``'s call)."""
    if not fused_train_enabled() or train_width(D) is None:
        return None
    if layer.transformer.native_kind not in ("affine", "inverse_affine", "shift"):
        return None                           # (shift: the affine launches with a scale logit of zero, _TrainPack)
    c = layer.coupling
    if not (layer._source_is_head and layer._target_is_tail and c.source_event_size == D // 2
            and c.target_event_size == D - D // 2):
        return None
    mlp = _plain_mlp(layer)
    if mlp is None or mlp[0].out_features > 15:
        return None
    if not all(p.dtype == torch.float32 for p in (mlp[0].weight, mlp[0].bias, mlp[1].weight, mlp[1].bias)):
        return None
    return mlp


def _ew_block(layer, d: int, D: int, W: Optional[int] = None) -> torch.Tensor:
    """Parameter block of a fixed ElementwiseAffine / ActNorm op as tfk_flow_run reads it
    (alpha[W] | beta[W] | sum log alpha, pad[3] | 1/alpha[W] for the dividing form), cached on the
    layer until its value changes (ActNorm: once, at its data-dependent initialisation).  ``W`` > D: rows in the padded
    training layout -- the padding columns get alpha = 1, beta = 0."""
    W = W or D
    key = (layer.value._version, layer.value.data_ptr(), d, W)
    hit = layer.__dict__.get("_tfk_ew_block")
    if hit is not None and hit[0] == key:
        return hit[1]
    v = layer.value.detach().reshape(D, 2)
    alpha = layer.transformer.constrain_scale(v[:, 0])                     # affine.py:33-34
    beta = v[:, 1].contiguous()
    ldc = torch.log(alpha).sum().reshape(1)
    if W != D:
        phys = training_layout(D, W, v.device)
        alpha = torch.ones(W, dtype=alpha.dtype, device=v.device).index_copy_(0, phys, alpha)
        beta = torch.zeros(W, dtype=beta.dtype, device=v.device).index_copy_(0, phys, beta)
    inverse_form = _affine_form_is_inverse(layer, d)
    pad = torch.zeros(3, dtype=torch.float32, device=v.device)
    parts = [alpha, beta, -ldc if inverse_form else ldc, pad]
    if inverse_form:
        parts.append(1.0 / alpha)
    block = torch.cat(parts).float().contiguous()
    layer.__dict__["_tfk_ew_block"] = (key, block)
    return block


class _PlanPacks:
    """All fusable coupling layers of one plan: ONE concatenation + ONE gather packs every layer's
    weights into MFMA-operand order per step, and ONE gather maps all accumulator-layout gradient
    blocks back -- instead of three small launches per layer.

    With ``fold`` the fixed elementwise layer (ActNorm) and the reversal that FOLLOW a fusable
    coupling ride along: forward = one flow program [coupling, elementwise] with a reversed store,
    backward = the reversed / scaled load of tfk_affine_coupling_train_bwd."""

    def __init__(self, plan, D: int, device, fold: bool):
        self.layers = []                      # (plan index, lin1, lin2, pack)
        self.fold = []                        # per fusable layer: (ew step or None, reversal step or None)
        self.folded_steps = set()
        self.D = D
        # the rows' width: D, or -- event sizes without fused launches of their own, every step padding-capable -- the
        # next width that has them, the rows in the padded training layout (training_layout)
        self.W = W = plan_width(plan, D)
        self.phys = training_layout(D, W, device) if W != D else None
        # odd event sizes: a logical reversal is not the physical one -- no folding of reversals into the coupling
        # launches, the permutation steps gather with reversal_index
        self.rev_index = reversal_index(D, W, device) if (W != D and D % 2) else None
        fold_rev = self.rev_index is None
        fusable = bool(native.lib().tfk_coupling_train_bwd_supported(W))
        pidx, gidx, off_flat, off_out = [], [], 0, 0
        for i, (layer, d, kind) in enumerate(plan):
            if kind != "coupling":
                continue
            mlp = _fused_bwd_layer(layer, D) if fusable else None
            if mlp is None:
                continue
            pack = _TrainPack.get(W, mlp[0].out_features, device, D, shift=layer.transformer.native_kind == "shift")
            ew_step = rev_step = None
            j = i + 1
            if fold and j < len(plan) and plan[j][2] == "elementwise" and not plan[j][0].value.requires_grad:
                ew_step = j
                j += 1
            if fold and fold_rev and j < len(plan) and plan[j][2] == "perm" and plan[j][0]._is_reversal:
                rev_step = j
            self.layers.append((i, mlp[0], mlp[1], pack))
            self.fold.append((ew_step, rev_step))
            self.folded_steps.update(t for t in (ew_step, rev_step) if t is not None)
            n_train = pack.param_index.numel()
            idx = [pack.param_index + off_flat]
            n_ew = 0
            if ew_step is not None:           # the elementwise block follows the operand block
                n_ew = (3 * W + 4) if _affine_form_is_inverse(plan[ew_step][0], plan[ew_step][1]) else (2 * W + 4)
                idx.append(torch.arange(n_ew, device=device) + (off_flat + pack.n_flat))
            pidx.append(torch.cat(idx))
            gidx.append(pack.grad_index + off_out)
            off_flat += pack.n_flat + n_ew
            off_out += pack.n_out
        # RQ-spline couplings with the in-kernel conditioner: the same ride-along pattern
        self.rqs_fold = {}
        self.rqs_layers = []                  # (plan index, lin1, lin2, _RqsTrainPack)
        for i, (layer, d, kind) in enumerate(plan):
            if kind != "coupling" or self.rqs_layer(layer) is None:
                continue
            ew_step = rev_step = None
            j = i + 1
            if fold and j < len(plan) and plan[j][2] == "elementwise" and not plan[j][0].value.requires_grad:
                ew_step = j
                j += 1
            if fold and fold_rev and j < len(plan) and plan[j][2] == "perm" and plan[j][0]._is_reversal:
                rev_step = j
            self.rqs_fold[i] = (ew_step, rev_step)
            self.folded_steps.update(t for t in (ew_step, rev_step) if t is not None)
            lin1, lin2 = self.rqs_layer(layer)
            self.rqs_layers.append((i, lin1, lin2, _RqsTrainPack.get(lin1.out_features, device, D)))
        self.rqs_slot = {i: k for k, (i, _, _, _) in enumerate(self.rqs_layers)}
        self.slot = {i: k for k, (i, _, _, _) in enumerate(self.layers)}
        if not self.layers:
            self.n_out_total = 0
        self.plan = list(plan)        # (a plain copy: the call's context must not be kept alive by the cache)
        if self.layers:
            self.block_sizes = [int(t.numel()) for t in pidx]
            self.param_index = torch.cat(pidx)
            self.grad_index = torch.cat(gidx)
            self.n_out_total = off_out
            self.zero = self.layers[0][3].zero

    # ---- parameters that live in ONE buffer (torchflows_amd/flat_optim.py) --------------------------------------
    def flat_capable(self) -> bool:
        """Every step is a permutation, a global elementwise layer or a coupling with the fused training launches:
        all gradients of the plan then come out of libtfk accumulators and can leave as slices of one buffer."""
        if not self.layers and not self.rqs_layers:
            return False
        # ONE layer object (or one parameter tensor) twice in the chain (shared weights): grad_src maps a slot of the parameter buffer to exactly
        # one accumulator position, so the second use would overwrite the first instead of adding to it -- such a plan
        # keeps the per-tensor route, where autograd sums the per-step gradients
        ids = [id(p) for layer, _, kind in self.plan for p in _layer_params(layer, kind)]
        if len(ids) != len(set(ids)):
            return False
        outer = rows_outer_enabled()
        for i, (layer, d, kind) in enumerate(self.plan):
            if kind in ("perm", "elementwise") or (kind == "coupling" and i in self.slot):
                continue
            if kind == "coupling" and i in self.rqs_slot and outer and self.rqs_layers[self.rqs_slot[i]][1].out_features <= 15:
                continue                      # (spline couplings: their row-contracting products on tfk_rows_outer)
            return False
        return True

    def flat_maps(self, fb):
        """Index maps between a FlatParams buffer and this plan, built once per buffer:
        ``src_index``: position in torch.cat([fb.P, aux]) of every element of the packed operand blocks (aux = the blocks
        of the folded fixed elementwise layers); ``grad_src``: for every element of fb.P the position of its gradient in
        the backward pass's output buffer [coupling accumulators | trainable elementwise gradients | one zero];
        ``ret``: per plan parameter (in ChainFunction's argument order) the piece of fb.split_sizes and its shape."""
        hit = self.__dict__.get("_flat_maps")
        if hit is not None and hit[0]() is fb:
            return hit[1]
        import weakref
        dev = fb.P.device
        D, W = self.D, self.W
        phys = (self.phys if self.phys is not None else torch.arange(D)).cpu()
        slot_of = fb.slot_of
        cat_to_src, aux_layers, aux_off = [], [], fb.n
        for (i, lin1, lin2, pack), (ew_step, _) in zip(self.layers, self.fold):
            for t in (lin1.weight, lin1.bias, lin2.weight, lin2.bias):
                k = slot_of.get(id(t))
                if k is None:
                    return None                  # (a frozen conditioner weight: the general route)
                cat_to_src.append(torch.arange(fb.numel[k]) + fb.offset[k])
            cat_to_src.append(torch.tensor([fb.zero_slot]))
            if ew_step is not None:
                ew_layer, ew_d, _ = self.plan[ew_step]
                n_ew = (3 * W + 4) if _affine_form_is_inverse(ew_layer, ew_d) else (2 * W + 4)
                cat_to_src.append(torch.arange(n_ew) + aux_off)
                aux_layers.append((ew_layer, ew_d))
                aux_off += n_ew
        src_index = None
        if self.layers:
            cat_to_src = torch.cat(cat_to_src).to(dev)
            src_index = cat_to_src[self.param_index]
        # spline couplings: per layer the full operand block (what the backward kernel takes) and, when a fixed
        # elementwise layer rides along, a second segment [forward operands | that layer's block] for the forward program
        rqs_index, rqs_sizes, rqs_seg = [], [], []
        for (i, lin1, lin2, rp) in self.rqs_layers:
            pos = []
            for t in (lin1.weight, lin1.bias, lin2.weight, lin2.bias):
                k = slot_of.get(id(t))
                if k is None:
                    return None
                pos.append(torch.arange(fb.numel[k]) + fb.offset[k])
            pos = torch.cat(pos + [torch.tensor([fb.zero_slot])])
            full = pos[rp.param_index.cpu()]
            bwd_seg = len(rqs_sizes)
            rqs_index.append(full)
            rqs_sizes.append(int(full.numel()))
            fwd_seg = None
            ew_step = self.rqs_fold[i][0]
            if ew_step is not None:
                ew_layer, ew_d, _ = self.plan[ew_step]
                n_ew = (3 * W + 4) if _affine_form_is_inverse(ew_layer, ew_d) else (2 * W + 4)
                fwd_seg = len(rqs_sizes)
                rqs_index.append(torch.cat([full[:rp.n_fwd], torch.arange(n_ew) + aux_off]))
                rqs_sizes.append(rp.n_fwd + n_ew)
                aux_layers.append((ew_layer, ew_d))
                aux_off += n_ew
            rqs_seg.append((bwd_seg, fwd_seg))
        # gradients: accumulator layout of all layers | trainable elementwise (D, 2) blocks | zero
        grad_src = torch.full((fb.n,), -1, dtype=torch.long)
        lo = 0
        gi = self.grad_index.cpu() if self.layers else None
        for i, lin1, lin2, pack in self.layers:
            for t, n in zip((lin1.weight, lin1.bias, lin2.weight, lin2.bias), pack.sizes):
                k = slot_of[id(t)]
                grad_src[fb.offset[k]:fb.offset[k] + n] = gi[lo:lo + n]
                lo += n
        ext = self.n_out_total
        rqs_out = {}
        for (i, lin1, lin2, rp) in self.rqs_layers:       # the three tfk_rows_outer products of the layer, accumulator order
            rqs_out[i] = ext
            ai, lo_a = rp.acc_index.cpu(), 0
            for t, n in zip((lin1.weight, lin1.bias, lin2.weight, lin2.bias), rp.acc_sizes):
                k = slot_of[id(t)]
                grad_src[fb.offset[k]:fb.offset[k] + n] = ext + ai[lo_a:lo_a + n]
                lo_a += n
            ext += rp.n_acc
        ew_out = {}
        for i, (layer, d, kind) in enumerate(self.plan):
            if kind == "elementwise" and i not in self.folded_steps and layer.value.requires_grad:
                k = slot_of.get(id(layer.value))
                if k is None:
                    return None
                ew_out[i] = ext              # the kernel writes dL/dvalue as (W, 2): logical row l is physical row phys[l]
                grad_src[fb.offset[k]:fb.offset[k] + 2 * D] = (ext + 2 * phys[:, None] + torch.arange(2)[None, :]).reshape(-1)
                ext += 2 * W
        grad_src[grad_src < 0] = ext                 # padding, parameters outside the plan: the zero
        ret = []
        for layer, _, kind in self.plan:
            for t in _layer_params(layer, kind):
                k = slot_of.get(id(t)) if t.requires_grad else None
                ret.append(None if k is None else (fb.piece_of_slot[k], tuple(t.shape)))
        maps = dict(src_index=src_index, aux_layers=aux_layers, grad_src=grad_src.to(dev), ew_out=ew_out,
                    n_ext=ext + 1, ret=ret, l2={}, rqs_out=rqs_out, rqs_seg=rqs_seg, rqs_sizes=rqs_sizes,
                    rqs_index=torch.cat(rqs_index).to(dev) if rqs_index else None)
        self.__dict__["_flat_maps"] = (weakref.ref(fb), maps)
        return maps

    def pack_flat(self, fb, maps):
        """``pack()`` as ONE gather out of the parameter buffer (+ the cached blocks of the folded fixed layers)."""
        src = fb.P
        if maps["aux_layers"]:
            src = torch.cat([fb.P] + [_ew_block(l, d, self.D, self.W) for l, d in maps["aux_layers"]])
        affine = list(src.index_select(0, maps["src_index"]).split(self.block_sizes)) if self.layers else []
        rqs = (list(src.index_select(0, maps["rqs_index"]).split(maps["rqs_sizes"]))
               if maps["rqs_index"] is not None else [])
        return affine, rqs

    @staticmethod
    def l2_vector(fb, maps, l2):
        """coef at the positions of the L2-regularised parameters of ``l2`` = {coef: [parameters]}, 0 elsewhere
        (cached per parameter set); None when one of them is not in the buffer."""
        key = tuple((c, tuple(id(t) for t in ps)) for c, ps in sorted(l2.items()))
        vec = maps["l2"].get(key)
        if vec is None:
            v = torch.zeros(fb.n, dtype=torch.float32)
            for c, ps in l2.items():
                for t in ps:
                    k = fb.slot_of.get(id(t))
                    if k is None or fb.params[k] is not t:
                        return None
                    v[fb.offset[k]:fb.offset[k] + fb.numel[k]] = c
            if len(maps["l2"]) > 4:
                maps["l2"].clear()
            vec = maps["l2"][key] = v.to(fb.P.device)
        return vec

    def rqs_layer(self, layer):
        """(lin1, lin2) when this coupling runs as the fused spline launches at this plan's row width."""
        if not native.lib().tfk_rqs_coupling_train_bwd_supported(self.W, int(getattr(layer.transformer, "n_bins", 0) or 0)):
            return None
        return _fused_rqs_layer(layer, self.D)

    # ---- rows in the padded training layout (self.W > self.D) -----------------------------------------------------
    def pad_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """(N, D) -> (N, W): first half at the head of plane A, second half at the tail of plane B, zeros between."""
        h = self.D // 2                               # (odd sizes: h sources at the head, D - h targets at the tail)
        wide = rows.new_zeros(rows.shape[0], self.W)
        wide[:, :h] = rows[:, :h]
        wide[:, self.W - (self.D - h):] = rows[:, h:]
        return wide

    def unpad_rows(self, wide: torch.Tensor) -> torch.Tensor:
        h = self.D // 2
        return torch.cat([wide[:, :h], wide[:, self.W - (self.D - h):]], dim=1)

    def ew_value(self, layer) -> torch.Tensor:
        """``layer.value`` as the (W, 2) block the elementwise kernels read (padding rows: zeros = the identity)."""
        v = layer.value.detach().reshape(self.D, 2)
        if self.phys is None:
            return v.contiguous()
        return v.new_zeros(self.W, 2).index_copy_(0, self.phys, v)

    def pack(self):
        """[packed block of layer k: operands (+ the folded elementwise parameters)] for the
        current weights."""
        pieces = []
        for (_, lin1, lin2, pack), (ew_step, _) in zip(self.layers, self.fold):
            pieces += [lin1.weight.detach().reshape(-1), lin1.bias.detach(),
                       lin2.weight.detach().reshape(-1), lin2.bias.detach(), self.zero]
            if ew_step is not None:
                pieces.append(_ew_block(self.plan[ew_step][0], self.plan[ew_step][1], self.D, self.W))
        packed = torch.cat(pieces)[self.param_index]
        return list(packed.split(self.block_sizes))


def _plan_packs(plan, D: int, device, fold: bool) -> _PlanPacks:
    owner = plan[0][0]
    key = (tuple(id(l) for l, _, _ in plan), tuple(d for _, d, _ in plan), str(device),
           fused_train_enabled(), fold, padded_train_enabled(), rows_outer_enabled())
    cache = owner.__dict__.setdefault("_tfk_plan_packs", {})
    if key not in cache:
        if len(cache) > 4:
            cache.clear()
        cache[key] = _PlanPacks(plan, D, device, fold)
    return cache[key]


def fully_fused(plan, D: int) -> bool:
    """Every coupling of the plan runs as the fused launches (flow program forward, fused training
    backward): the step then consists of libtfk kernels and elementwise ATen ops only -- no GEMM
    library calls -- which is the configuration verified to survive hipGraph capture."""
    couplings = [layer for layer, _, kind in plan if kind == "coupling"]
    if not couplings or any(kind not in ("coupling", "perm", "elementwise") for _, _, kind in plan):
        return False
    if all(_fused_bwd_layer(layer, D) is not None for layer in couplings):
        # (an event size that would need padding in a plan that cannot keep it has no fused launches after all)
        return bool(native.lib().tfk_coupling_train_bwd_supported(plan_width(plan, D)))
    # RQ-spline couplings: the fused launch + the row-contracting products on tfk_rows_outer (hidden width <= 15)
    def spline_ok(layer):
        mlp = _fused_rqs_layer(layer, D)
        return mlp is not None and rows_outer_enabled() and mlp[0].out_features <= 15
    return plan_width(plan, D) == 64 and all(spline_ok(layer) for layer in couplings)


SPLIT_K_ROWS = 1024


def _outer_sum(a: torch.Tensor, b_aug: torch.Tensor) -> torch.Tensor:
    """``a^T @ b_aug`` for tall operands (N x m, N x k, N >> m, k): the contraction runs over the
    batch rows, so a plain GEMM has one tiny output tile and no parallelism (hipBLASLt: ~450 us at
    N = 2^18).  Split the rows into slabs of 1024, one batched GEMM over the slabs, then add the
    slab results (fixed order: deterministic)."""
    N, m = a.shape
    C = SPLIT_K_ROWS
    S = N // C
    out = None
    if S >= 2:
        head = torch.bmm(a[:S * C].view(S, C, m).transpose(1, 2), b_aug[:S * C].view(S, C, -1)).sum(dim=0)
        out = head
        if S * C < N:
            out = out + a[S * C:].t() @ b_aug[S * C:]
        return out
    return a.t() @ b_aug


def _mlp_backward(W1: torch.Tensor, W2: torch.Tensor, x_a: torch.Tensor, a1: torch.Tensor, gh: torch.Tensor):
    """Reverse mode of h = W2 tanh(W1 x_a + b1) + b2 given a1 = tanh(.) and gh = dL/dh.
    Returns (g_xa, dW1, db1, dW2, db2); the bias gradients ride along as a column of ones."""
    N = gh.shape[0]
    ones = torch.ones(N, 1, dtype=gh.dtype, device=gh.device)
    dW2b = _outer_sum(gh, torch.cat([a1, ones], dim=1))             # (TP, H + 1)
    g_pre = (gh @ W2) * (1.0 - a1 * a1)                              # tanh'
    dW1b = _outer_sum(g_pre, torch.cat([x_a, ones], dim=1))          # (H, S + 1)
    g_xa = g_pre @ W1
    return g_xa, dW1b[:, :-1], dW1b[:, -1], dW2b[:, :-1], dW2b[:, -1]


def _made_mlp(layer):
    """(MaskedLinear, MaskedLinear) when the conditioner is the default two-layer MADE (masked Linear,
    Tanh, masked Linear; transforms.py:184-267) without global parameters or output bounds."""
    import math
    import torch.nn as nn
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import MADE
    ct = layer.conditioner_transform
    if not isinstance(ct, MADE) or ct.n_global_parameters != 0 or ct.context_shape is not None:
        return None
    if ct.output_lower_bound != -math.inf or ct.output_upper_bound != math.inf:
        return None
    mods = list(ct.sequential)
    if (len(mods) == 3 and isinstance(mods[0], MADE.MaskedLinear) and isinstance(mods[1], nn.Tanh)
            and isinstance(mods[2], MADE.MaskedLinear) and mods[0].in_features == layer.n_dim):
        return mods[0], mods[2]
    return None


def _affine_form_is_inverse(layer, d: int) -> bool:
    """Does this step evaluate (x - beta) / alpha (True) or alpha x + beta (False)?"""
    return (d == INVERSE) != (layer.transformer.native_kind == "inverse_affine")


class ChainFunction(torch.autograd.Function):
    """rows (N, D) -> (rows (N, D), log-det (N,)) through a whole plan."""

    @staticmethod
    def forward(ctx, plan, rows: torch.Tensor, *params: torch.Tensor):
        from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm
        N, D = rows.shape
        logdet = torch.empty(N, dtype=torch.float32, device=rows.device)
        started = False
        cur = rows
        cur_is_saved = True          # never write into the caller's tensor
        saved: List[Optional[torch.Tensor]] = []
        # an ActNorm that still has to take its statistics from this batch cannot ride along
        needs_init = any(kind == "elementwise" and isinstance(layer, ActNorm) and d == FORWARD
                         and layer.training and layer.first_training_batch_pass for layer, d, kind in plan)
        packs = _plan_packs(plan, D, rows.device, fold=not needs_init)
        # parameters homed in one buffer (FlatAdamW): operands by one gather, gradients as slices of one buffer
        fb = maps = l2vec = None
        if flat_enabled() and packs.flat_capable():
            from torchflows_amd import flat_optim
            fb = flat_optim.lookup(params)
            maps = packs.flat_maps(fb) if fb is not None else None
            if maps is None:
                fb = None
        rqs_packed = None
        if fb is not None:
            packed, rqs_packed = packs.pack_flat(fb, maps)
            if plan.l2:
                l2vec = packs.l2_vector(fb, maps, plan.l2)
        else:
            packed = packs.pack() if packs.layers else []
        W = packs.W
        if W != D:                   # padded training layout: every kernel below sees (N, W) rows
            cur, cur_is_saved = packs.pad_rows(rows), False
        rqs_blocks = {}
        kept = {}                    # step -> (conditioner input, its output WITH graph) of the couplings that keep one
        for step, (layer, d, kind) in enumerate(plan):
            if step in packs.folded_steps:
                saved.append(None)          # ran inside the preceding coupling's flow program
                continue
            if kind == "perm":
                out = torch.empty_like(cur)
                perm = None if layer._is_reversal else (layer._fwd_index32 if d == FORWARD else layer._inv_index32)
                if packs.rev_index is not None:      # (padded rows of an odd event size: reversals only, as a gather)
                    perm = packs.rev_index
                native.permute(cur, perm, out)
                saved.append(None)
                cur, cur_is_saved = out, False
            elif kind == "elementwise":
                if isinstance(layer, ActNorm) and d == FORWARD and layer.training and layer.first_training_batch_pass:
                    layer._data_dependent_init((cur if W == D else packs.unpad_rows(cur)).view(N, *layer.event_shape))
                keep_input = layer.value.requires_grad
                out = cur if (not cur_is_saved and not keep_input) else torch.empty_like(cur)
                native.elementwise_affine(cur, packs.ew_value(layer), out, logdet,
                                          layer.transformer.native_kind == "inverse_affine",
                                          accumulate=started, inverse=(d == INVERSE))
                started = True
                saved.append(cur if keep_input else None)
                cur, cur_is_saved = out, False
            elif kind == "elementwise_ctx":
                h = layer.conditioner_transform(x=None, context=plan.context).reshape(N, -1).contiguous()
                out = torch.empty_like(cur)
                native.affine_coupling(cur, h, out, logdet, None, D, accumulate=started,
                                       inverse=_affine_form_is_inverse(layer, d))
                started = True
                saved.append(cur)
                cur, cur_is_saved = out, False
            elif kind == "made":
                h = layer.conditioner_transform(cur.view(N, *layer.event_shape), None).reshape(N, -1).contiguous()
                out = torch.empty_like(cur)
                tk, tr = layer.transformer.native_kind, layer.transformer
                if tk == "rqs":
                    native.rqs_coupling(cur, h, out, logdet, None, D, tr.n_bins, tr.boundary,
                                        accumulate=started, inverse=False)
                elif tk == "lrs":
                    native.lrs_coupling(cur, h, out, logdet, None, D, tr.n_bins, tr.boundary,
                                        accumulate=started, inverse=False)
                else:       # the parallel pass always applies transformer.forward (layers_base.py:196-199)
                    native.affine_coupling(cur, h, out, logdet, None, D, accumulate=started,
                                           inverse=(tk == "inverse_affine"))
                started = True
                saved.append(cur)
                cur, cur_is_saved = out, False
            elif step in packs.slot:
                # conditioner + transform (+ the fixed elementwise layer and the reversal that
                # follow) in one launch: a flow program on the matrix cores whose operand block is
                # the head of the layer's training pack
                k = packs.slot[step]
                pack = packs.layers[k][3]
                ew_step, rev_step = packs.fold[k]
                ops = [(3 if _affine_form_is_inverse(layer, d) else 2, 0, pack.steps2, 0)]   # TFK_OP_AFFINE_*
                if ew_step is not None:
                    ew_layer, ew_d, _ = plan[ew_step]
                    ops.append((1 if _affine_form_is_inverse(ew_layer, ew_d) else 0, 0, 0,
                                pack.param_index.numel()))                                # TFK_OP_EW_*
                out = torch.empty_like(cur)
                native.flow_run_mfma(cur, out, logdet, None, None, None, ops, packed[k], accumulate=started,
                                     reverse_out=rev_step is not None)
                started = True
                saved.append(cur)
                cur, cur_is_saved = out, False
            elif kind == "coupling" and packs.rqs_layer(layer) is not None:
                # conditioner + spline in one launch (single-op flow program); the operand block is
                # kept for the backward kernel
                lin1, lin2 = packs.rqs_layer(layer)
                rp = _RqsTrainPack.get(lin1.out_features, cur.device, D)
                seg = maps["rqs_seg"][packs.rqs_slot[step]] if fb is not None else None
                block = rqs_packed[seg[0]] if seg is not None else rp.pack(lin1, lin2)
                tr = layer.transformer
                import math
                op = (7 if d == INVERSE else 6, 0, rp.steps2, 0, 8, float(tr.boundary),
                      float(1.0 - tr.min_bin_size * tr.n_bins), float(math.log(math.expm1(1 - tr.min_delta))))
                ew_step, rev_step = packs.rqs_fold.get(step, (None, None))
                ops, prm = [op], block[:rp.n_fwd]
                if ew_step is not None:      # the fixed elementwise layer that follows rides along
                    ew_layer, ew_d, _ = plan[ew_step]
                    ops.append((1 if _affine_form_is_inverse(ew_layer, ew_d) else 0, 0, 0, rp.n_fwd))
                    prm = rqs_packed[seg[1]] if seg is not None else torch.cat([prm, _ew_block(ew_layer, ew_d, D, W)])
                out = torch.empty_like(cur)
                native.flow_run_mfma(cur, out, logdet, None, None, None, ops, prm, accumulate=started,
                                     reverse_out=rev_step is not None)
                started = True
                saved.append(cur)
                rqs_blocks[step] = block
                cur, cur_is_saved = out, False
            else:
                if _keeps_graph(layer) and any(ctx.needs_input_grad):
                    # a convolutional conditioner: evaluated ONCE, with its graph (the activations it saves are what a
                    # re-evaluation in the backward pass would produce again -- and a BatchNorm in training mode counts
                    # the batch once, as the reference's single forward does)
                    S = layer.coupling.source_event_size
                    x_a = cur[:, :S] if layer._source_is_head else cur.index_select(1, layer._source_index)
                    x_a = x_a.detach().requires_grad_(True)
                    with torch.enable_grad():
                        h2 = layer.conditioner_transform(x_a.reshape(N, *layer.coupling.constant_shape),
                                                         context=_layer_context(layer, plan.context)).reshape(N, -1)
                    kept[step] = (x_a, h2)
                    h = h2.detach().contiguous()
                else:
                    h = _conditioner(layer, cur, plan.context).reshape(N, -1).contiguous()
                out = torch.empty_like(cur)
                T = layer.coupling.target_event_size
                tgt = None if layer._target_is_tail else layer._target_index32
                tk = layer.transformer.native_kind
                if tk in ("affine", "inverse_affine"):
                    native.affine_coupling(cur, h, out, logdet, tgt, T, accumulate=started,
                                           inverse=_affine_form_is_inverse(layer, d))
                elif tk == "rqs":
                    native.rqs_coupling(cur, h, out, logdet, tgt, T, layer.transformer.n_bins,
                                        layer.transformer.boundary, accumulate=started, inverse=(d == INVERSE))
                elif tk == "lrs":
                    native.lrs_coupling(cur, h, out, logdet, tgt, T, layer.transformer.n_bins,
                                        layer.transformer.boundary, accumulate=started, inverse=(d == INVERSE))
                elif tk == "conv1x1":
                    native.conv1x1_coupling(cur, h, out, logdet, tgt, T, layer.transformer.n_channels,
                                            accumulate=started, inverse=(d == INVERSE))
                else:
                    native.shift_coupling(cur, h, out, logdet, tgt, T, accumulate=started, inverse=(d == INVERSE))
                started = True
                saved.append(cur)
                cur, cur_is_saved = out, False
        if not started:
            logdet.zero_()
        if cur is rows:
            cur = rows.clone()
        if W != D:
            cur = packs.unpad_rows(cur)
        ctx.plan = plan
        ctx.saved_rows = saved
        ctx.n_params = len(params)
        ctx.packs, ctx.packed = packs, packed
        ctx.rqs_blocks = rqs_blocks
        ctx.kept = kept
        ctx.flat, ctx.flat_maps, ctx.l2vec = fb, maps, l2vec
        if l2vec is not None:
            # sum_coef coef * sum_p ||p||^2 (layers_base.py:38-48) over the buffer: the gradient joins the flat one
            ctx.flat_version = fb.P._version
            reg = torch.dot(fb.P * l2vec, fb.P)
            return cur, logdet, reg
        return cur, logdet

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rows: Optional[torch.Tensor], g_logdet: Optional[torch.Tensor],
                 g_reg: Optional[torch.Tensor] = None):
        plan, saved = ctx.plan, ctx.saved_rows
        fb, maps = ctx.flat, ctx.flat_maps
        first = next(s for s in saved if s is not None) if any(s is not None for s in saved) else None
        ref = g_rows if g_rows is not None else (g_logdet if g_logdet is not None else first)
        device = ref.device
        N = (g_logdet.shape[0] if g_logdet is not None else
             (g_rows.shape[0] if g_rows is not None else first.shape[0]))     # (only the L2 output was differentiated)
        D = plan[0][0].n_dim
        W = ctx.packs.W                # (rows in the padded training layout when W > D)
        if W != D:
            g = (torch.zeros(N, W, dtype=torch.float32, device=device) if g_rows is None
                 else ctx.packs.pad_rows(g_rows))
        else:
            g = (torch.zeros(N, D, dtype=torch.float32, device=device) if g_rows is None
                 else g_rows.contiguous().clone())
        gld = (torch.zeros(N, dtype=torch.float32, device=device) if g_logdet is None
               else g_logdet.contiguous())
        grads_per_step: List[List[Optional[torch.Tensor]]] = [[] for _ in plan]
        packs = ctx.packs
        if fb is not None:
            out_all = torch.empty(maps["n_ext"], dtype=torch.float32, device=device)
            out_all[-1:].zero_()
        else:
            out_all = (torch.empty(packs.n_out_total, dtype=torch.float32, device=device)
                       if packs.layers else None)
        for i in range(len(plan) - 1, -1, -1):
            layer, d, kind = plan[i]
            x_in = saved[i]
            if i in packs.folded_steps:     # handled by the load of the coupling's backward kernel
                grads_per_step[i] = [None] * len(_layer_params(layer, kind))
                continue
            if kind == "perm":
                out = torch.empty_like(g)
                perm = None if layer._is_reversal else (layer._inv_index32 if d == FORWARD else layer._fwd_index32)
                if packs.rev_index is not None:      # (an involution: its own inverse)
                    perm = packs.rev_index
                native.permute(g, perm, out)
                g = out
            elif kind == "elementwise":
                want = layer.value.requires_grad
                lo = maps["ew_out"].get(i) if fb is not None else None
                gv = native.elementwise_affine_bwd(x_in, packs.ew_value(layer), g,
                                                   gld, want, inverse=_affine_form_is_inverse(layer, d),
                                                   out=None if lo is None else out_all[lo:lo + 2 * W])
                if want and W != D and fb is None:
                    gv = gv.index_select(0, packs.phys)          # (W, 2) -> the D logical rows
                grads_per_step[i] = [gv.view_as(layer.value) if (want and fb is None) else None]
            elif kind == "elementwise_ctx":
                cparams = _module_params(layer.conditioner_transform)
                with torch.enable_grad():
                    h2 = layer.conditioner_transform(x=None, context=plan.context).reshape(N, -1)
                hc = h2.detach().contiguous()
                gh = torch.empty_like(hc)
                native.affine_coupling_bwd(x_in, hc, g, gld, gh, None, D, inverse=_affine_form_is_inverse(layer, d))
                wanted = [p for p in cparams if p.requires_grad]
                outs = torch.autograd.grad(h2, wanted, gh, allow_unused=True) if wanted else ()
                it = iter(outs)
                grads_per_step[i] = [next(it) if p.requires_grad else None for p in cparams]
                grads_per_step[i] = [torch.zeros_like(p) if (gp is None and p.requires_grad) else gp
                                     for gp, p in zip(grads_per_step[i], cparams)]
            elif kind == "made":
                # h depends on every input position (through the masks): re-evaluate MADE with a graph,
                # the transformer's reverse-mode kernel gives dL/dh and the direct dL/dx, autograd the rest
                cparams = _module_params(layer.conditioner_transform)
                tk, tr = layer.transformer.native_kind, layer.transformer
                mlp = _made_mlp(layer)
                if mlp is not None:             # masked MLP written out (weight-gradient GEMMs split over rows)
                    lin1, lin2 = mlp
                    W1, W2 = lin1.weight.detach() * lin1.mask, lin2.weight.detach() * lin2.mask
                    a1 = torch.tanh(torch.addmm(lin1.bias.detach(), x_in, W1.t()))
                    hc = torch.addmm(lin2.bias.detach(), a1, W2.t())
                else:
                    x_req = x_in.detach().requires_grad_(True)
                    with torch.enable_grad():
                        h2 = layer.conditioner_transform(x_req.view(N, *layer.event_shape), None).reshape(N, -1)
                    hc = h2.detach().contiguous()
                gh = torch.empty_like(hc)
                if tk == "rqs":
                    native.rqs_coupling_bwd(x_in, hc, g, gld, gh, None, D, tr.n_bins, tr.boundary, inverse=False)
                elif tk == "lrs":
                    native.lrs_coupling_bwd(x_in, hc, g, gld, gh, None, D, tr.n_bins, tr.boundary, inverse=False)
                else:
                    native.affine_coupling_bwd(x_in, hc, g, gld, gh, None, D, inverse=(tk == "inverse_affine"))
                if mlp is not None:
                    g_x, dW1, db1, dW2, db2 = _mlp_backward(W1, W2, x_in, a1, gh)
                    g.add_(g_x)
                    by_param = {id(lin1.weight): dW1 * lin1.mask, id(lin1.bias): db1,
                                id(lin2.weight): dW2 * lin2.mask, id(lin2.bias): db2}
                    grads_per_step[i] = [
                        (by_param[id(p)] if id(p) in by_param else torch.zeros_like(p)) if p.requires_grad else None
                        for p in cparams]
                    continue
                wanted = [p for p in cparams if p.requires_grad]
                outs = torch.autograd.grad(h2, [x_req] + wanted, gh, allow_unused=True)
                g.add_(outs[0].reshape(N, D))
                it = iter(outs[1:])
                grads_per_step[i] = [next(it) if p.requires_grad else None for p in cparams]
                grads_per_step[i] = [torch.zeros_like(p) if (gp is None and p.requires_grad) else gp
                                     for gp, p in zip(grads_per_step[i], cparams)]
            else:
                T = layer.coupling.target_event_size
                S = layer.coupling.source_event_size
                tgt = None if layer._target_is_tail else layer._target_index32
                tk = layer.transformer.native_kind
                cparams = _module_params(layer.conditioner_transform)
                if i in packs.slot:         # one launch: conditioner, transform and MLP backward
                    k = packs.slot[i]
                    pack = packs.layers[k][3]
                    n_train = pack.param_index.numel()
                    ew_step, rev_step = packs.fold[k]
                    gscale = None
                    if ew_step is not None:     # d(alpha x + beta)/dx = alpha, d((x - beta)/alpha)/dx = 1/alpha
                        ew_layer, ew_d, _ = plan[ew_step]
                        lo = n_train + ((2 * W + 4) if _affine_form_is_inverse(ew_layer, ew_d) else 0)
                        gscale = ctx.packed[k][lo:lo + W]
                    native.affine_coupling_train_bwd(x_in, g, gld, ctx.packed[k][:n_train], pack.steps2,
                                                     out_all[k * pack.n_out:(k + 1) * pack.n_out], pack.workspace,
                                                     inverse_form=_affine_form_is_inverse(layer, d),
                                                     gscale=gscale, g_reversed=rev_step is not None)
                    continue
                if i in ctx.rqs_blocks:     # conditioner re-evaluated in the kernel, dL/dh written once
                    lin1, lin2 = packs.rqs_layer(layer)
                    rp = _RqsTrainPack.get(lin1.out_features, g.device, D)
                    gh_perm = torch.empty(N, 768, dtype=torch.float32, device=g.device)
                    gpre_perm = torch.empty(N, 16, dtype=torch.float32, device=g.device)
                    outer = rows_outer_enabled() and lin1.out_features <= 15
                    hid_perm = torch.empty(N, 16, dtype=torch.float32, device=g.device) if outer else None
                    ew_step, rev_step = packs.rqs_fold.get(i, (None, None))
                    gscale = None
                    if ew_step is not None:
                        ew_layer, ew_d, _ = plan[ew_step]
                        blk = _ew_block(ew_layer, ew_d, D, W)
                        lo = (2 * W + 4) if _affine_form_is_inverse(ew_layer, ew_d) else 0
                        gscale = blk[lo:lo + W]
                    native.rqs_coupling_train_bwd(x_in, g, gld, ctx.rqs_blocks[i], rp.steps2, gh_perm, gpre_perm,
                                                  layer.transformer.n_bins, layer.transformer.boundary,
                                                  inverse=(d == INVERSE), gscale=gscale,
                                                  g_reversed=rev_step is not None, hid_perm=hid_perm)
                    if outer:
                        # the three products that contract over the batch rows, on the matrix cores without a GEMM-library
                        # call (tfk_rows_outer: deterministic, capturable), un-permuted by ONE gather -- per layer, or, with
                        # the parameters in one buffer, by the gather that maps every accumulator of the plan at once
                        lo_acc = maps["rqs_out"][i] if fb is not None else None
                        acc = (out_all[lo_acc:lo_acc + rp.n_acc] if fb is not None
                               else torch.empty(rp.n_acc, dtype=torch.float32, device=g.device))
                        native.rows_outer(gh_perm, 768, hid_perm, acc[:768 * 16])
                        native.rows_outer(x_in, 32, gpre_perm, acc[768 * 16:768 * 16 + 32 * 16])
                        native.rows_outer(gpre_perm, 16, hid_perm, acc[768 * 16 + 32 * 16:])
                        if fb is not None:
                            continue
                        dW1, db1, dW2, db2 = (t.view(shp) for t, shp in
                                              zip(acc.index_select(0, rp.acc_index).split(rp.acc_sizes), rp.acc_shapes))
                        by_param = {id(lin1.weight): dW1, id(lin1.bias): db1, id(lin2.weight): dW2, id(lin2.bias): db2}
                        grads_per_step[i] = [
                            (by_param[id(p)] if id(p) in by_param else torch.zeros_like(p)) if p.requires_grad else None
                            for p in cparams]
                        continue
                    x_a = x_in[:, :S]
                    a1 = torch.tanh(torch.addmm(lin1.bias, x_a, lin1.weight.t()))
                    ones = torch.ones(N, 1, dtype=torch.float32, device=g.device)
                    dW2b = _outer_sum(gh_perm, torch.cat([a1, ones], dim=1)).index_select(0, rp.gh_col)
                    g_pre = gpre_perm.index_select(1, rp.gpre_col)
                    dW1b = _outer_sum(g_pre, torch.cat([x_a, ones], dim=1))
                    by_param = {id(lin1.weight): dW1b[:, :-1], id(lin1.bias): dW1b[:, -1],
                                id(lin2.weight): dW2b[:, :-1], id(lin2.bias): dW2b[:, -1]}
                    grads_per_step[i] = [
                        (by_param[id(p)] if id(p) in by_param else torch.zeros_like(p)) if p.requires_grad else None
                        for p in cparams]
                    continue
                x_a = x_in[:, :S] if layer._source_is_head else x_in.index_select(1, layer._source_index)
                mlp = _plain_mlp(layer)
                if mlp is not None:            # re-evaluate h; keep the hidden activations
                    lin1, lin2 = mlp
                    a1 = torch.tanh(torch.addmm(lin1.bias, x_a, lin1.weight.t()))
                    hc = torch.addmm(lin2.bias, a1, lin2.weight.t())
                elif ctx.kept.get(i) is not None:      # the forward pass kept the conditioner's graph (used once)
                    x_a, h2 = ctx.kept[i]
                    ctx.kept[i] = None
                    hc = h2.detach().contiguous()
                else:                          # any other conditioner: re-evaluate with a graph
                    x_a = x_a.detach().requires_grad_(True)
                    from torchflows_amd import convnet_train
                    with torch.enable_grad(), convnet_train.recomputing():      # (same batch again: BatchNorm's running
                        h2 = layer.conditioner_transform(x_a.reshape(N, *layer.coupling.constant_shape),   # statistics stay)
                                                         context=_layer_context(layer, plan.context)).reshape(N, -1)
                    hc = h2.detach().contiguous()
                gh = torch.empty_like(hc)
                if tk in ("affine", "inverse_affine"):
                    native.affine_coupling_bwd(x_in, hc, g, gld, gh, tgt, T,
                                               inverse=_affine_form_is_inverse(layer, d))
                elif tk == "rqs":
                    native.rqs_coupling_bwd(x_in, hc, g, gld, gh, tgt, T, layer.transformer.n_bins,
                                            layer.transformer.boundary, inverse=(d == INVERSE))
                elif tk == "lrs":
                    native.lrs_coupling_bwd(x_in, hc, g, gld, gh, tgt, T, layer.transformer.n_bins,
                                            layer.transformer.boundary, inverse=(d == INVERSE))
                elif tk == "conv1x1":
                    native.conv1x1_coupling_bwd(x_in, hc, g, gld, gh, tgt, T, layer.transformer.n_channels,
                                                inverse=(d == INVERSE))
                else:
                    native.shift_coupling_bwd(g, gh, tgt, T, inverse=(d == INVERSE))
                if mlp is not None:
                    g_xa, dW1, db1, dW2, db2 = _mlp_backward(lin1.weight, lin2.weight, x_a, a1, gh)
                    by_param = {id(lin1.weight): dW1, id(lin1.bias): db1, id(lin2.weight): dW2, id(lin2.bias): db2}
                    grads_per_step[i] = [by_param.get(id(p)) if p.requires_grad else None for p in cparams]
                    # (global_theta_flat is empty here: its gradient is an empty tensor)
                    grads_per_step[i] = [torch.zeros_like(p) if (gp is None and p.requires_grad) else gp
                                         for gp, p in zip(grads_per_step[i], cparams)]
                else:
                    wanted = [p for p in cparams if p.requires_grad]
                    outs = torch.autograd.grad(h2, [x_a] + wanted, gh, allow_unused=True)
                    g_xa = outs[0].reshape(N, S)
                    it = iter(outs[1:])
                    grads_per_step[i] = [next(it) if p.requires_grad else None for p in cparams]
                if layer._source_is_head:
                    g[:, :S].add_(g_xa)
                else:
                    g.index_add_(1, layer._source_index, g_xa)
        if fb is not None:
            # ONE gather: accumulator layout of every layer -> the layout of the parameter buffer; the gradients leave
            # as slices of it (what FlatAdamW.step looks for)
            G = out_all.index_select(0, maps["grad_src"])
            if ctx.l2vec is not None and g_reg is not None:
                if fb.P._version != ctx.flat_version:
                    raise RuntimeError("the parameter buffer was modified between the forward and the backward pass")
                G.addcmul_(fb.P, ctx.l2vec * g_reg, value=2.0)
            fb.last_grad = G
            pieces = G.split_with_sizes(fb.split_sizes)
            flat = [None if r is None else pieces[r[0]].view(r[1]) for r in maps["ret"]]
            assert len(flat) == ctx.n_params
            g_in = None if not ctx.needs_input_grad[1] else (g if W == D else packs.unpad_rows(g))
            return (None, g_in, *flat)
        if packs.layers:                    # accumulator layout -> parameter layout, all layers at once
            pieces = out_all[packs.grad_index]
            lo = 0
            for i, lin1, lin2, pack in packs.layers:
                dW1, db1, dW2, db2 = (t.view(shp) for t, shp in
                                      zip(pieces[lo:lo + sum(pack.sizes)].split(pack.sizes), pack.shapes))
                lo += sum(pack.sizes)
                by_param = {id(lin1.weight): dW1, id(lin1.bias): db1, id(lin2.weight): dW2, id(lin2.bias): db2}
                cparams = _module_params(plan[i][0].conditioner_transform)
                grads_per_step[i] = [
                    (by_param[id(p)] if id(p) in by_param else torch.zeros_like(p)) if p.requires_grad else None
                    for p in cparams]
        flat: List[Optional[torch.Tensor]] = []
        for gs in grads_per_step:
            flat.extend(gs)
        assert len(flat) == ctx.n_params
        g_in = None if not ctx.needs_input_grad[1] else (g if W == D else packs.unpad_rows(g))
        return (None, g_in, *flat)


class GaussLogProbFunction(torch.autograd.Function):
    """``DiagonalGaussian.log_prob(rows) [+ log_det]`` (gaussian.py:46-54, flows.py:647-648) with
    fixed loc / scale: one forward and one reverse-mode launch."""

    @staticmethod
    def forward(ctx, rows, loc, log_scale, log_det):
        out = torch.empty(rows.shape[0], dtype=torch.float32, device=rows.device)
        native.diag_gauss_logprob(rows, loc, log_scale, log_det, out)
        ctx.save_for_backward(rows, loc, log_scale)
        ctx.has_ld = log_det is not None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, glp):
        rows, loc, log_scale = ctx.saved_tensors
        glp = glp.contiguous()
        g = None
        if ctx.needs_input_grad[0]:
            g = torch.empty_like(rows)
            native.diag_gauss_logprob_bwd(rows, loc, log_scale, glp, g)
        return g, None, None, (glp if ctx.has_ld and ctx.needs_input_grad[3] else None)


def run(composition, plan, x: torch.Tensor, context=None) -> Tuple[torch.Tensor, torch.Tensor]:
    from torchflows_amd.utils import as_rows
    rows, batch = as_rows(x, composition.event_shape)
    if context is not None:
        # (a hand-built composition takes its context_shape from its FIRST layer, which may have none: the layers that
        # do take a context reshape the flat rows themselves)
        cs = composition.context_shape
        plan.context = (context.reshape(rows.shape[0], *cs) if cs is not None
                        else context.reshape(rows.shape[0], -1)).contiguous()
    params: List[torch.Tensor] = []
    for layer, _, kind in plan:
        params.extend(_layer_params(layer, kind))
    outs = ChainFunction.apply(plan, rows, *params)
    out, ld = outs[0], outs[1]
    if plan.l2:                      # the caller asked for the L2 penalty in the same node: hand it over (or None)
        composition.__dict__["_tfk_l2_out"] = outs[2] if len(outs) == 3 else None
    return out.view(x.shape), ld.view(batch)
