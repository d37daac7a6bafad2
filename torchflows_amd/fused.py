"""Host-side compiler from a ``BijectiveComposition`` to libtfk *flow programs*.

``tfk_flow_run`` (csrc/tfk_flow.hip) keeps each row in registers and applies a list of ops
to it, with the conditioner MLP evaluated in-kernel -- one launch for a whole chain of
layers.  This module turns the reference-shaped layer objects into that
list:

* permutation layers emit nothing: the compiler tracks ``pos[l]`` = physical position of
  logical element ``l`` and packs every later layer's parameters in physical order;
* ElementwiseAffine / ActNorm become ``alpha*x+beta`` / ``(x-beta)/alpha`` ops with
  ``alpha = exp(u/2 + log(1-1e-10)) + 1e-10`` and ``sum log alpha`` evaluated once per
  compilation (they are batch-constant; the reference recomputes them per row);
* Affine / Shift couplings on the HalfSplit mask whose conditioner is the default
  ``FeedForward`` (Linear, Tanh, Linear) become coupling ops; after the folded reversals
  the conditioner's input must still be exactly one half ("plane") of the physical row.

* RQ-spline couplings (8 bins) and the parallel map of MADE-based affine / RQ-spline layers have
  ops of their own on the matrix-core kernel (``tfk_flow_run_mfma``: event sizes 64 / 128 / 256);
* other event sizes ride on the same kernel padded: even sizes with each half at the head of its
  plane, odd sizes (HalfSplit moves one element across the halves at every reversal) with every
  element owning an index in both planes and ``OP_PLANE_SWAP`` ops in front of the couplings --
  padding elements see zero weights, i.e. the identity with log-det 0.

Anything else (other masks, context, deeper / non-tanh conditioners, ActNorm that still has to
initialise itself from data) makes ``compile_chain`` return ``None`` and the composition runs
layer by layer (bijections/base.py).  Programs are cached per (direction, device) and rebuilt
when any parameter version changes.
"""
from __future__ import annotations

import os
import warnings
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from torchflows_amd.utils import debug_switch

from torchflows_amd import native

OP_EW_MULADD, OP_EW_SUBDIV, OP_AFFINE_FWD, OP_AFFINE_INV, OP_SHIFT_FWD, OP_SHIFT_INV, \
    OP_RQS_FWD, OP_RQS_INV, OP_MADE_FWD, OP_MADE_INV, OP_MADE_RQS, OP_PLANE_SWAP, \
    OP_AFFINE_FWD_LEAN, OP_AFFINE_INV_LEAN, OP_SHIFT_FWD_LEAN, OP_SHIFT_INV_LEAN, OP_EW_FMA, \
    OP_RQS_FWD_LEAN, OP_RQS_INV_LEAN, OP_EWC_MULADD, OP_EWC_SUBDIV, OP_MADE_FWD_LEAN, OP_MADE_INV_LEAN, \
    OP_LRS_FWD_LEAN, OP_LRS_INV_LEAN, OP_MADE_RQS_FWD_LEAN, OP_MADE_RQS_INV_LEAN, OP_MADE_LRS_FWD_LEAN, \
    OP_MADE_LRS_INV_LEAN = range(29)
MAX_CONTEXT = 16      # context elements a flow program takes (4 k-steps of 4, csrc/tfk_flow_mfma.h: kCtxSteps)
LOG2E = 1.4426950408889634
AFF_C0 = -1.000000082790371e-10       # float32(log(1 - 1e-10)), affine.py:19-23
RQS_PAD = 24          # 23 spline parameters per element, padded to 6 float4
MAX_HIDDEN_RQS = 32
FORWARD, INVERSE = 0, 1

# parameters staged in LDS per launch; 40 KB keeps 4 workgroups (16 waves) per CU and holds
# the whole RealNVP(64, n_layers=8) program (35.8 KB)
MAX_PARAM_BYTES = 40 * 1024
# the matrix-core layout stores MFMA A-operands per lane (zero-padded to 16 hidden units):
# RealNVP(64, n_layers=8) is 49 KB; its kernel runs 512-thread workgroups, 2 per CU (register-bound)
MAX_PARAM_BYTES_MFMA = 52 * 1024
# wider rows hold more registers per lane, so fewer workgroups fit a CU whatever the LDS use:
# D = 128 (120 VGPRs) runs 2 x 512 threads per CU, D = 256 (179 VGPRs) one -- their launches may
# stage more of the program (each extra launch re-reads and re-writes all the rows)
MFMA_BUDGET_WIDE = {128: 76 * 1024, 256: 150 * 1024}
LEAN_BUDGET_64 = 76 * 1024        # lean affine / shift chains at D = 64 (two 512-thread workgroups per CU either way)
MAX_OPS = 96


def enabled() -> bool:
    return os.environ.get("TORCHFLOWS_AMD_FUSED", "1") != "0"


def mfma_enabled() -> bool:
    return os.environ.get("TORCHFLOWS_AMD_MFMA", "1") != "0"


@dataclass
class Segment:
    ops: List[tuple]                          # (kind, src_plane, H, offset[, K, boundary, scale, c])
    params: torch.Tensor                      # fp32 device block, numel % 4 == 0
    mfma: bool = False                        # packed for tfk_flow_run_mfma (matrix-core conditioner)

    def packed_ops(self):
        """The op records as the ctypes array the C-ABI takes, built once per segment."""
        arr = self.__dict__.get("_ops_c")
        if arr is None:
            arr = self.__dict__["_ops_c"] = native._pack_ops(self.ops)
        return arr


@dataclass
class CompiledChain:
    D: int                     # width of the rows the kernels see (padded: 64 / 128 / 256)
    segments: List[Segment]
    pos: torch.Tensor          # int64 (D_log,): physical position of logical element l at the end
    identity_out: bool         # rows come out in logical order
    version: int
    D_log: int = 0             # event size; < D when the two halves are padded to a supported width
    pos_in: Optional[torch.Tensor] = None     # padded chains: physical position of logical element l on entry
    ctx_width: Optional[int] = None           # context programs: elements per context row the packed weights expect

    def reversed_out(self) -> bool:
        """The rows come out exactly reversed (an odd number of ReversePermutationMatrix layers behind the last fold) and
        the last launch is a matrix-core one: its store writes column c to D - 1 - c (flag bit 1 of tfk_flow_run_mfma)
        instead of a tfk_permute pass over the rows afterwards."""
        hit = self.__dict__.get("_reversed_out")
        if hit is None:
            hit = (not self.identity_out and self.pos_in is None and bool(self.segments) and self.segments[-1].mfma
                   and self.pos.numel() == self.D
                   and bool(torch.equal(self.pos, torch.arange(self.D - 1, -1, -1, device=self.pos.device))))
            self.__dict__["_reversed_out"] = hit
        return hit


def _pad4(t: torch.Tensor) -> torch.Tensor:
    r = (-t.numel()) % 4
    return t if r == 0 else torch.cat([t, t.new_zeros(r)])


def _flatten(layers, attr: str):
    """[(layer, direction)] with nested compositions expanded; None if a layer's bound
    forward/inverse is not one of the tagged implementations."""
    from torchflows_amd.bijections.base import BijectiveComposition, method_direction
    out = []
    for layer in layers:
        d = method_direction(getattr(layer, attr))
        if d is None:
            return None
        if isinstance(layer, BijectiveComposition):
            inner = layer.layers if d == FORWARD else list(layer.layers)[::-1]
            sub = _flatten(inner, "forward" if d == FORWARD else "inverse")
            if sub is None:
                return None
            out.extend(sub)
        else:
            out.append((layer, d))
    return out


def _tensor_slots(module: nn.Module):
    """[(owner dict, name, tensor)] for every parameter and buffer below ``module``."""
    slots = []
    for m in module.modules():
        for owner in (m._parameters, m._buffers):
            for k, t in owner.items():
                if t is not None:
                    slots.append((owner, k, t))
    return slots


# Bumped whenever ANY module of this package has its tensors moved or converted (``_apply``: .to() / .cuda() /
# .double() ...) and whenever a tensor slot is found replaced: mixed into every version below, so that a cache entry built
# before such an event -- under any key, on the moved module or on an ancestor whose program packs its weights -- can never
# match again.  (A move keeps the tensor objects and their version counters; only the storage changes.)
_EPOCH = [0]


def _params_version(module: nn.Module) -> int:
    """Changes whenever a parameter / buffer below ``module`` is modified in place (version counter) or replaced by
    another tensor (slot identity).  Moves and conversions (``.to()``, ``.cuda()``, ``.double()``: the tensor object
    stays, its storage changes) are caught by ``Bijection._apply`` / ``BaseFlow._apply``, which drop the caches -- so
    the per-call check is ONE pass over a flat list reading ``_version`` (the ``data_ptr()`` walk this replaces cost
    ~35 us of the ~57 us a small-batch ``log_prob`` call spent on the host)."""
    slots = module.__dict__.get("_tfk_slots")
    if slots is None:
        slots = module.__dict__["_tfk_slots"] = _tensor_slots(module)
    for owner, k, t in slots:
        if owner.get(k) is not t:                     # a tensor was replaced: re-walk the tree, retire every older entry
            slots = module.__dict__["_tfk_slots"] = _tensor_slots(module)
            module.__dict__.pop("_tfk_static_ok", None)
            _EPOCH[0] += 1
            break
    v = len(slots) + 1000003 * _EPOCH[0]
    for _, _, t in slots:
        v = v * 1000003 + t._version
    return v & 0xFFFFFFFFFFFFFFF


def _layout_version(module: nn.Module) -> int:
    """Kept for callers that key on the layout alone: the identity of the slot list (rebuilt when a tensor is replaced;
    dropped by ``_apply`` when tensors move)."""
    slots = module.__dict__.get("_tfk_slots")
    if slots is None:
        slots = module.__dict__["_tfk_slots"] = _tensor_slots(module)
    return id(slots)


def static_ok(module: nn.Module) -> bool:
    """Every floating-point parameter / buffer below ``module`` is fp32 on one HIP device (cached until a tensor moves:
    ``_apply`` drops the cache)."""
    hit = module.__dict__.get("_tfk_static_ok")
    if hit is not None and hit[0] == _EPOCH[0]:
        return hit[1]
    slots = module.__dict__.get("_tfk_slots")
    if slots is None:
        slots = module.__dict__["_tfk_slots"] = _tensor_slots(module)
    devices = set()
    ok = True
    for _, _, t in slots:
        if t.is_floating_point():
            devices.add(t.device)
            ok = ok and t.device.type == "cuda" and t.dtype == torch.float32
    ok = ok and len(devices) <= 1
    module.__dict__["_tfk_static_ok"] = (_EPOCH[0], ok)
    return ok


def tensors_moved(module: nn.Module) -> None:
    """Called from ``_apply`` (``.to()`` / ``.cuda()`` / ``.float()`` ...): every cache that depends on where the
    tensors live or on their values is dropped, on this module and -- because a parent's program packs this module's
    weights -- this is also invoked for every ancestor that is itself being moved (Module._apply recurses)."""
    _EPOCH[0] += 1                      # ancestors that are NOT being moved hold programs packed from these tensors too
    for m in module.modules():          # (submodules that are not bijections -- conditioner blocks -- keep caches too)
        d = m.__dict__
        for k in [k for k in d if k.startswith("_tfk_") and k != "_tfk_declined_warned"]:
            del d[k]


def any_requires_grad(module: nn.Module) -> bool:
    slots = module.__dict__.get("_tfk_slots")
    if slots is None:
        _layout_version(module)
        slots = module.__dict__["_tfk_slots"]
    return any(t.requires_grad for _, _, t in slots)


def _elementwise_op(layer, d: int, pos: torch.Tensor, D: int, Dp: Optional[int] = None):
    from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm
    kind = layer.transformer.native_kind
    if kind not in ("affine", "inverse_affine") or not layer.use_global_parameters:
        return None
    if isinstance(layer, ActNorm) and layer.training and layer.first_training_batch_pass:
        return None                       # must see a batch first (layers.py:58-68)
    value = layer.value.detach().reshape(D, 2)
    alpha = layer.transformer.constrain_scale(value[:, 0])
    beta = value[:, 1]
    subdiv = (d == INVERSE) != (kind == "inverse_affine")
    ld = torch.log(alpha).sum()
    ld = -ld if subdiv else ld
    Dp = D if Dp is None else Dp
    alpha_p = alpha.new_ones(Dp)               # padding elements: alpha = 1, beta = 0 (identity, log alpha = 0)
    beta_p = beta.new_zeros(Dp)
    alpha_p[pos] = alpha
    beta_p[pos] = beta
    parts = [alpha_p, beta_p, ld.reshape(1), ld.new_zeros(3)]
    if subdiv:
        parts.append(1.0 / alpha_p)           # correctly rounded reciprocal for the in-kernel quotient
    return (OP_EW_SUBDIV if subdiv else OP_EW_MULADD, 0, 0), torch.cat(parts)


def _pack_mfma(kind: str, d: int, plane: int, H: int, D: int, W1t, b1, W2p, b2p):
    """Parameter block of one affine / shift coupling for tfk_flow_run_mfma
    (csrc/tfk_flow_mfma.hip): A1[EPL][HT][64] | b1[HT][4][4] | A2[T2][steps2][64] | b2[T2][4][4]
    (HT = 1, 2, 4 or 8 tiles of 16 hidden units; RQS: HT = 1).
    Lane l = (q = l >> 4, i = l & 15).  GEMM 1: D-row i <-> hidden unit 4*(i & 3) + (i >> 2),
    k-step s of lane-group q <-> physical source element EPL*q + s.  GEMM 2, tile t: D-row
    i = 4*q2 + r <-> parameter (r & 1) of target element EPL*q2 + 2t + (r >> 1) (affine) or
    target element EPL*q2 + 4t + r (shift); k-step r' of lane-group q <-> hidden unit 4r' + q."""
    half, EPL = D // 2, D // 8
    dev, dt = W1t.device, W1t.dtype
    P = W2p.shape[1]
    steps2 = (H + 3) // 4
    HT = 1 if steps2 <= 4 else (2 if steps2 <= 8 else (4 if steps2 <= 16 else 8))       # 16-unit tiles of the hidden layer
    W1pad = torch.zeros(16 * HT, half, dtype=dt, device=dev)
    W1pad[:H] = W1t
    b1pad = torch.zeros(16 * HT, dtype=dt, device=dev)
    b1pad[:H] = b1
    W2pad = torch.zeros(half, P, 16 * HT, dtype=dt, device=dev)
    W2pad[:, :, :H] = W2p
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    unit1 = 4 * (il & 3) + (il >> 2)
    # A1[s][t][lane]: tile t holds hidden units 16t .. 16t+15 (D-row i <-> unit 16t + unit1(i))
    A1 = torch.stack([W1pad[16 * t + unit1, EPL * ql + s] for s in range(EPL) for t in range(HT)])
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    b1m = torch.stack([b1pad[16 * t + 4 * rr + qq] for t in range(HT)])           # [t][q][r]
    q2, r2 = il >> 2, il & 3
    if kind == "rqs":
        # 23 spline parameters per element padded to 24: tile 6e + c holds parameters 4c .. 4c+3
        # of target element EPL*q2 + e
        W2q = torch.zeros(half, 24, 16, dtype=dt, device=dev)
        W2q[:, :P] = W2pad
        b2q = torch.zeros(half, 24, dtype=dt, device=dev)
        b2q[:, :P] = b2p
        A2, b2m = [], []
        for e in range(EPL):
            for c in range(6):
                for r1 in range(steps2):
                    A2.append(W2q[EPL * q2 + e, 4 * c + r2, 4 * r1 + ql])
                b2m.append(b2q[EPL * qq + e, 4 * c + rr])
        block = torch.cat([A1.reshape(-1), b1m.reshape(-1), torch.stack(A2).reshape(-1),
                           torch.stack(b2m).reshape(-1)])
        return (OP_RQS_FWD if d == FORWARD else OP_RQS_INV, plane, steps2), block
    T2 = EPL // 2 if P == 2 else EPL // 4
    A2, b2m = [], []
    for t in range(T2):
        if P == 2:
            m_l, p_l = EPL * q2 + 2 * t + (r2 >> 1), r2 & 1
            m_b, p_b = EPL * qq + 2 * t + (rr >> 1), rr & 1
        else:
            m_l, p_l = EPL * q2 + 4 * t + r2, torch.zeros_like(r2)
            m_b, p_b = EPL * qq + 4 * t + rr, torch.zeros_like(rr)
        for r1 in range(steps2):
            A2.append(W2pad[m_l, p_l, 4 * r1 + ql])                               # (64,)
        b2m.append(b2p[m_b, p_b])                                                  # (4, 4)
    block = torch.cat([A1.reshape(-1), b1m.reshape(-1), torch.stack(A2).reshape(-1),
                       torch.stack(b2m).reshape(-1)])
    if kind == "shift":
        op = OP_SHIFT_FWD if d == FORWARD else OP_SHIFT_INV
    else:
        op = OP_AFFINE_FWD if (d == FORWARD) != (kind == "inverse_affine") else OP_AFFINE_INV
    return (op, plane, steps2), block


def _coupling_op(layer, d: int, pos: torch.Tensor, D: int, mfma: bool = False, Dp: Optional[int] = None):
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import FeedForward
    from torchflows_amd.utils import event_size as _esize
    kind = layer.transformer.native_kind
    if kind not in ("affine", "inverse_affine", "shift", "rqs"):
        return None
    C = _esize(layer.context_shape) if layer.context_shape is not None else 0
    if C and (not mfma or C > MAX_CONTEXT):
        return None                       # a context: matrix-core interpreter only (tfk_flow_run_mfma_ctx)
    if kind == "rqs" and layer.transformer.n_bins != 8:
        return None                       # the fused spline op is built for the default 8 bins
    half = D // 2
    c = layer.coupling
    S, T = c.source_event_size, c.target_event_size        # HalfSplit: D // 2 and D - D // 2 (coupling_masks.py:78-81)
    if not (c.source_is_head and c.target_is_tail and S == half and S + T == D):
        return None
    if T != S and (Dp is None or Dp == D or not mfma):
        return None                           # odd event sizes: padded matrix-core programs only
    ct = layer.conditioner_transform
    if type(ct) is not FeedForward or ct.n_global_parameters != 0:
        return None
    if ct.output_lower_bound != float("-inf") or ct.output_upper_bound != float("inf"):
        return None
    mods = list(ct.sequential)
    if not (len(mods) == 4 and isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.Tanh)
            and isinstance(mods[2], nn.Linear) and isinstance(mods[3], nn.Unflatten)):
        return None
    P = {"shift": 1, "rqs": 23, "lrs": 32}.get(kind, 2)
    W1, b1 = mods[0].weight.detach(), mods[0].bias.detach()          # (H, S), (H,)
    W2, b2 = mods[2].weight.detach(), mods[2].bias.detach()          # (T*P, H), (T*P,)
    H = W1.shape[0]
    if W1.shape[1] != S + C or W2.shape[0] != T * P or W2.shape[1] != H:
        return None
    W1c = W1[:, S:] if C else None            # [x_A || context]: the context's columns (context.py:46-60)
    W1 = W1[:, :S]
    if C and H > 16:
        return None
    src_pos, tgt_pos = pos[:S], pos[S:]
    Dp = D if Dp is None else Dp
    hp = Dp // 2                              # plane width the kernel sees (> half when padded)
    if Dp != D and not mfma:
        return None                           # padded planes: matrix-core path only
    # (a padded RQ-spline element has all-zero parameters: equal bins, unit derivatives -- the identity at its
    # value 0, log-det 0 up to the 1 ulp of softplus(c) + 1e-5 vs 1)
    plane = int(src_pos[0].item()) // hp
    if not bool(((src_pos // hp) == plane).all()) or not bool(((tgt_pos // hp) == 1 - plane).all()):
        return None
    # padding elements: zero weights => h = 0 => alpha = exp(c0) + 1e-10 = 1, beta = 0, log alpha = 0 exactly
    W1t = W1.new_zeros(H, hp)
    W1t[:, src_pos - plane * hp] = W1
    m_t = tgt_pos - (1 - plane) * hp
    W2p = torch.zeros(hp, P, H, dtype=W2.dtype, device=W2.device)
    W2p[m_t] = W2.reshape(T, P, H)
    b2p = torch.zeros(hp, P, dtype=b2.dtype, device=b2.device)
    b2p[m_t] = b2.reshape(T, P)
    if mfma:
        if H > (16 if kind == "rqs" else 128) or (kind == "rqs" and Dp > 128):
            return None
        head, block = _pack_mfma(kind, d, plane, H, Dp, W1t, b1, W2p, b2p)
        if C:                                 # further GEMM-1 k-steps: A1c[cs][64], lane (q, i) <-> context element 4 s + q
            cs = (C + 3) // 4
            W1cp = W1c.new_zeros(16, 4 * cs)
            W1cp[:H, :C] = W1c
            lane = torch.arange(64, device=W1c.device)
            ql, il = lane >> 4, lane & 15
            unit1 = 4 * (il & 3) + (il >> 2)
            A1c = torch.stack([W1cp[unit1, 4 * s_ + ql] for s_ in range(cs)])
            block = torch.cat([block, A1c.reshape(-1).to(block.dtype)])
            head = (head[0], head[1] | (cs << 4)) + tuple(head[2:])
        if kind == "rqs":
            import math
            import numpy as np
            tr = layer.transformer
            head = head + (8, float(tr.boundary), float(np.float32(1.0 - tr.min_bin_size * tr.n_bins)),
                           float(np.float32(math.log(math.expm1(1 - tr.min_delta)))))
        return head, block
    if kind == "rqs":
        if H > MAX_HIDDEN_RQS:
            return None
        G = D // 8
        # physical target m = 4*j + e  ->  [k][e][j][24] (23 parameters + 1 pad)
        W2q = torch.zeros(half, RQS_PAD, H, dtype=W2.dtype, device=W2.device)
        W2q[:, :P] = W2p
        b2q = torch.zeros(half, RQS_PAD, dtype=b2.dtype, device=b2.device)
        b2q[:, :P] = b2p
        W2t = W2q.reshape(G, 4, RQS_PAD, H).permute(3, 1, 0, 2).reshape(-1)
        b2t = b2q.reshape(G, 4, RQS_PAD).permute(1, 0, 2).reshape(-1)
        block = torch.cat([W1t.reshape(-1), _pad4(b1), W2t, b2t])
        tr = layer.transformer
        op = OP_RQS_FWD if d == FORWARD else OP_RQS_INV
        import math
        import numpy as np
        scale = float(np.float32(1.0 - tr.min_bin_size * tr.n_bins))
        c = float(np.float32(math.log(math.expm1(1 - tr.min_delta))))
        return (op, plane, H, 8, float(tr.boundary), scale, c), block
    block = torch.cat([W1t.reshape(-1), _pad4(b1), W2p.permute(2, 0, 1).reshape(-1), b2p.reshape(-1)])
    if kind == "shift":
        op = OP_SHIFT_FWD if d == FORWARD else OP_SHIFT_INV
    else:
        op = OP_AFFINE_FWD if (d == FORWARD) != (kind == "inverse_affine") else OP_AFFINE_INV
    return (op, plane, H), block


def _elementwise_ctx_op(layer, d: int, pos: torch.Tensor, D: int, Dp: int, lean: bool = False):
    """An elementwise affine layer whose parameters come from the CONTEXT through the default Linear conditioner
    (layers_base.py:300-318) as a TFK_OP_EWC_* op: Ac[EPL][cs][64] | bc[EPL][4][4] (csrc/tfk_flow_mfma.h: ewc_m)."""
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import Linear
    from torchflows_amd.utils import event_size as _esize
    kind = layer.transformer.native_kind
    ct = layer.conditioner_transform
    if kind not in ("affine", "inverse_affine") or layer.use_global_parameters or type(ct) is not Linear:
        return None
    if ct.n_global_parameters != 0 or ct.output_lower_bound != float("-inf") or ct.output_upper_bound != float("inf"):
        return None
    lin = ct.sequential[0]
    C = _esize(layer.context_shape)
    if not isinstance(lin, nn.Linear) or lin.in_features != C or lin.out_features != 2 * D or C > MAX_CONTEXT:
        return None
    cs = (C + 3) // 4
    EPL, hp = Dp // 8, Dp // 2
    dev = lin.weight.device
    W = torch.zeros(Dp, 2, 4 * cs, dtype=lin.weight.dtype, device=dev)      # physical element, parameter, context column
    W[pos, :, :C] = lin.weight.detach().view(D, 2, C)
    bq = torch.zeros(Dp, 2, dtype=lin.bias.dtype, device=dev)               # padding: zero logits = the identity
    bq[pos] = lin.bias.detach().view(D, 2)
    if lean:      # inside a lean program (csrc/tfk_flow_chain.h: ewc_lean): alpha = exp2(.) + 1e-10, as the lean couplings
        W, bq = W.double(), bq.double()
        W[:, 0] = W[:, 0] * (0.5 * LOG2E)
        bq[:, 0] = (bq[:, 0] * 0.5 + AFF_C0) * LOG2E
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    q2, r2 = il >> 2, il & 3
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    A, bm = [], []
    for t in range(EPL):
        plane, tt = (0, t) if t < EPL // 2 else (1, t - EPL // 2)
        m_l = plane * hp + EPL * q2 + 2 * tt + (r2 >> 1)
        for s_ in range(cs):
            A.append(W[m_l, r2 & 1, 4 * s_ + ql])
        bm.append(bq[plane * hp + EPL * qq + 2 * tt + (rr >> 1), rr & 1])
    block = torch.cat([torch.stack(A).reshape(-1), torch.stack(bm).reshape(-1)])
    subdiv = (d == INVERSE) != (kind == "inverse_affine")
    return (OP_EWC_SUBDIV if subdiv else OP_EWC_MULADD, cs << 4, 0), block


def _plane_swap(layer, pos: torch.Tensor, Dp: int):
    """Odd event sizes run in a layout where logical element l owns index l of BOTH planes (Dp >= 2 D), so an
    element changes planes by a swap with the zero that sits at its index in the other plane.  Returns
    ``(item or None, new pos)`` such that the coupling's source elements share one plane and its targets the
    other (the assignment that needs fewer swaps)."""
    hp = Dp // 2
    S = layer.coupling.source_event_size
    plane = pos // hp
    src, tgt = plane[:S], plane[S:]
    cost0 = int((src != 0).sum() + (tgt != 1).sum())        # sources in plane 0, targets in plane 1
    cost1 = int((src != 1).sum() + (tgt != 0).sum())
    want_src = 0 if cost0 <= cost1 else 1
    want = torch.cat([torch.full_like(src, want_src), torch.full_like(tgt, 1 - want_src)])
    wrong = plane != want
    if not bool(wrong.any()):
        return None, pos
    slot = pos % hp
    mask = torch.zeros(hp, dtype=torch.float32, device=pos.device)
    mask[slot[wrong]] = 1.0
    new_pos = torch.where(wrong, want * hp + slot, pos)
    return ((OP_PLANE_SWAP, 0, 0), mask), new_pos


def _made_op(layer, d: int, pos: torch.Tensor, D: int, Dp: Optional[int] = None):
    """A MADE-based affine layer's PARALLEL map as one matrix-core op (tfk_flow_mfma.hip: made_m).
    None for the sequential map, other transformers, deeper MADEs, a context."""
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import MADE
    if d == layer._sequential_when or layer.context_shape is not None:
        return None
    kind = layer.transformer.native_kind
    ct = layer.conditioner_transform
    if kind not in ("affine", "inverse_affine", "rqs") or ct.n_global_parameters != 0:
        return None
    if ct.output_lower_bound != float("-inf") or ct.output_upper_bound != float("inf"):
        return None
    mods = list(ct.sequential)
    if not (len(mods) == 3 and isinstance(mods[0], MADE.MaskedLinear) and isinstance(mods[1], nn.Tanh)
            and isinstance(mods[2], MADE.MaskedLinear)):
        return None
    H = mods[0].out_features
    Dp = D if Dp is None else Dp              # padded row width: padding elements get zero weights (identity)
    if mods[0].in_features != D or H > 128 or (Dp == 256 and H > 16):
        return None
    if kind == "rqs":
        return _made_rqs_op(layer, mods, pos, D, H, Dp)
    half, EPL = Dp // 2, Dp // 8
    T2 = EPL // 2
    steps2 = (H + 3) // 4
    HT = 1 if steps2 <= 4 else (2 if steps2 <= 8 else (4 if steps2 <= 16 else 8))
    W1 = (mods[0].weight * mods[0].mask).detach()                       # (H, D) logical columns
    W2 = (mods[2].weight * mods[2].mask).detach().view(D, 2, H)        # logical element, parameter, unit
    dev, dt = W1.device, W1.dtype
    W1p = torch.zeros(16 * HT, Dp, dtype=dt, device=dev)
    W1p[:H, pos] = W1                                                   # physical columns
    b1p = torch.zeros(16 * HT, dtype=dt, device=dev)
    b1p[:H] = mods[0].bias.detach()
    W2p = torch.zeros(Dp, 2, 16 * HT, dtype=dt, device=dev)
    W2p[pos, :, :H] = W2                                                # physical elements
    b2p = torch.zeros(Dp, 2, dtype=dt, device=dev)
    b2p[pos] = mods[2].bias.detach().view(D, 2)
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    unit1 = 4 * (il & 3) + (il >> 2)
    q2, r2 = il >> 2, il & 3
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    A1 = torch.stack([W1p[16 * t + unit1, plane * half + EPL * ql + s]
                      for plane in range(2) for s in range(EPL) for t in range(HT)])
    b1m = torch.stack([b1p[16 * t + 4 * rr + qq] for t in range(HT)])
    A2, b2m = [], []
    for plane in range(2):
        for t in range(T2):
            for r1 in range(steps2):
                A2.append(W2p[plane * half + EPL * q2 + 2 * t + (r2 >> 1), r2 & 1, 4 * r1 + ql])
            b2m.append(b2p[plane * half + EPL * qq + 2 * t + (rr >> 1), rr & 1])
    block = torch.cat([A1.reshape(-1), b1m.reshape(-1), torch.stack(A2).reshape(-1), torch.stack(b2m).reshape(-1)])
    divide = (kind == "inverse_affine")                # the parallel map uses transformer.forward
    return (OP_MADE_INV if divide else OP_MADE_FWD, 0, steps2), block


def _made_rqs_op(layer, mods, pos: torch.Tensor, D: int, H: int, Dp: Optional[int] = None):
    """The parallel map of a MADE-based RQ-spline layer (8 bins, hidden <= 16, D <= 128) as one matrix-core
    op (tfk_flow_mfma.h: made_rqs_m): A1[2 EPL][64] | b1[4][4] | A2[2 EPL 6][steps2][64] | b2[2 EPL 6][4][4]."""
    import math
    import numpy as np
    tr = layer.transformer
    Dp = D if Dp is None else Dp
    if tr.n_bins != 8 or H > 16 or Dp > 128:
        return None
    half, EPL = Dp // 2, Dp // 8
    steps2 = (H + 3) // 4
    W1 = (mods[0].weight * mods[0].mask).detach()                       # (H, D) logical columns
    W2 = (mods[2].weight * mods[2].mask).detach().view(D, 23, H)       # logical element, parameter, unit
    dev, dt = W1.device, W1.dtype
    W1p = torch.zeros(16, Dp, dtype=dt, device=dev)
    W1p[:H, pos] = W1                                                   # physical columns
    b1p = torch.zeros(16, dtype=dt, device=dev)
    b1p[:H] = mods[0].bias.detach()
    W2p = torch.zeros(Dp, 24, 16, dtype=dt, device=dev)
    W2p[pos, :23, :H] = W2                                              # physical elements, 23 + 1 pad
    b2p = torch.zeros(Dp, 24, dtype=dt, device=dev)
    b2p[pos, :23] = mods[2].bias.detach().view(D, 23)
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    unit1 = 4 * (il & 3) + (il >> 2)
    q2, r2 = il >> 2, il & 3
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    A1 = torch.stack([W1p[unit1, plane * half + EPL * ql + s] for plane in range(2) for s in range(EPL)])
    b1m = b1p[4 * rr + qq]
    A2, b2m = [], []
    for plane in range(2):
        for e in range(EPL):
            for c in range(6):
                for r1 in range(steps2):
                    A2.append(W2p[plane * half + EPL * q2 + e, 4 * c + r2, 4 * r1 + ql])
                b2m.append(b2p[plane * half + EPL * qq + e, 4 * c + rr])
    block = torch.cat([A1.reshape(-1), b1m.reshape(-1), torch.stack(A2).reshape(-1), torch.stack(b2m).reshape(-1)])
    head = (OP_MADE_RQS, 0, steps2, 8, float(tr.boundary), float(np.float32(1.0 - tr.min_bin_size * tr.n_bins)),
            float(np.float32(math.log(math.expm1(1 - tr.min_delta)))))
    return head, block


# ---------------------------------------------------------------------------------------------
# lean chains (csrc/tfk_flow_chain.h): couplings of one kind, elementwise layers deferred
# ---------------------------------------------------------------------------------------------
def lean_enabled() -> bool:
    return debug_switch("lean", "1") != "0"


def _lean_elementwise(layer, d: int, D: int):
    """(alpha, beta, divide) of an ElementwiseAffine / ActNorm with global parameters in LOGICAL order, fp64 --
    ``z = alpha x + beta`` or, with ``divide``, ``z = (x - beta) / alpha`` -- or None."""
    from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm
    kind = layer.transformer.native_kind
    if kind not in ("affine", "inverse_affine") or not layer.use_global_parameters:
        return None
    if isinstance(layer, ActNorm) and layer.training and layer.first_training_batch_pass:
        return None
    value = layer.value.detach().reshape(D, 2)
    alpha = layer.transformer.constrain_scale(value[:, 0]).double()      # the fp32 scale the reference uses
    beta = value[:, 1].double()
    return alpha, beta, (d == INVERSE) != (kind == "inverse_affine")


def _lean_coupling(layer, d: int, pos: torch.Tensor, D: int, Dp: int, allow_ctx: bool = False):
    """Physical-order conditioner weights of an affine / shift HalfSplit coupling with the default FeedForward(tanh)
    conditioner of hidden width <= 16: (lean kind 0..3, source plane, H, W1t (H, hp), b1, W2p (hp, P, H), b2p (hp, P))
    in fp64, or None (the checks of ``_coupling_op``)."""
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import FeedForward
    from torchflows_amd.utils import event_size as _esize
    kind = layer.transformer.native_kind
    if kind not in ("affine", "inverse_affine", "shift", "rqs", "lrs"):
        return None
    C = _esize(layer.context_shape) if layer.context_shape is not None else 0
    if C and not (allow_ctx and C <= MAX_CONTEXT):
        return None                                           # (a context: further GEMM-1 k-steps, lean context programs)
    if kind in ("rqs", "lrs") and layer.transformer.n_bins != 8:
        return None
    half = D // 2
    c = layer.coupling
    S, T = c.source_event_size, c.target_event_size
    if not (c.source_is_head and c.target_is_tail and S == half and S + T == D and T - S in (0, 1)):
        return None                                           # (T = S + 1: odd event sizes, _compile_lean(odd=True))
    ct = layer.conditioner_transform
    if type(ct) is not FeedForward or ct.n_global_parameters != 0:
        return None
    if ct.output_lower_bound != float("-inf") or ct.output_upper_bound != float("inf"):
        return None
    mods = list(ct.sequential)
    if not (len(mods) == 4 and isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.Tanh)
            and isinstance(mods[2], nn.Linear) and isinstance(mods[3], nn.Unflatten)):
        return None
    P = {"shift": 1, "rqs": 23, "lrs": 32}.get(kind, 2)
    W1, b1 = mods[0].weight.detach().double(), mods[0].bias.detach().double()
    W2, b2 = mods[2].weight.detach().double(), mods[2].bias.detach().double()
    H = W1.shape[0]
    if H > (31 if kind in ("rqs", "lrs") else 16) or W1.shape[1] != S + C or W2.shape[0] != T * P or W2.shape[1] != H:
        return None
    W1c = W1[:, S:] if C else None                            # [x_A || context] (conditioning/context.py:46-60)
    W1 = W1[:, :S]
    hp = Dp // 2
    src_pos, tgt_pos = pos[:S], pos[S:]
    plane = int(src_pos[0].item()) // hp
    if not bool(((src_pos // hp) == plane).all()) or not bool(((tgt_pos // hp) == 1 - plane).all()):
        return None
    W1t = W1.new_zeros(H, hp)
    W1t[:, src_pos - plane * hp] = W1
    m_t = tgt_pos - (1 - plane) * hp
    W2p = W2.new_zeros(hp, P, H)
    W2p[m_t] = W2.reshape(T, P, H)
    b2p = b2.new_zeros(hp, P)
    b2p[m_t] = b2.reshape(T, P)
    if kind == "shift":
        lk = 2 if d == FORWARD else 3
    elif kind == "rqs":
        lk = 4 if d == FORWARD else 5
    elif kind == "lrs":
        lk = 8 if d == FORWARD else 9
    else:
        lk = 0 if (d == FORWARD) != (kind == "inverse_affine") else 1
    if allow_ctx:
        return lk, plane, H, W1t, b1, W2p, b2p, W1c
    return lk, plane, H, W1t, b1, W2p, b2p


def lean_bf16x3_enabled(Dp: int = 64) -> bool:
    """bf16 x 3 operands for GEMM 2 of affine / shift chains (TORCHFLOWS_AMD_DEBUG=lean_bf16x3=0 / 1; default "auto").
    D = 64: OFF unless forced -- measured on RealNVP-64 (2^20 rows): 252.0 us per launch against 255.8 us with fp32
    operands.  The 12 f32 MFMAs it removes (384 matrix cycles per wave-layer) come back as 12 bf16 MFMAs (~192) + 22 vector
    instructions for the split + a 1024-thread workgroup per CU (85 KB of operands).
    D = 256 (streamed operands): ON -- round 4, RealNVP-256 on 2^19 rows, same box, alternating: 617 -> 545 us per launch
    (8.25 -> 9.34e8 evals/s, +13 %): 48 f32 MFMAs per wave-layer (1 536 matrix cycles of the layer's 2 560) become 48 bf16
    ones (~770) for the same 22-instruction split; parity unchanged (tests/test_gpu_fused.py)."""
    mode = debug_switch("lean_bf16x3", "auto")
    return mode == "1" or (mode == "auto" and Dp == 256)


def _pack_lean(lk: int, H: int, Dp: int, W1t, b1, W2p, b2p, pre_s, pre_t, bf16x3: bool = False, W1c=None) -> torch.Tensor:
    """Parameter block of a lean coupling op (csrc/tfk_flow_chain.h): lane-major MFMA A-operands
    A1[EPL/4][64][4] | b1[4][4] | A2[nA2/4][64][4] | b2[T2][4][4] | pre_s[hp] | pre_t[hp], with W1 / b1 multiplied by
    2 log2(e) and (affine) the scale-logit rows of W2 / b2 by log2(e) / 2, b2 += c0 log2(e).  Inputs fp64, physical
    order, the pending elementwise maps of the source plane already folded into W1t / b1."""
    hp, EPL = Dp // 2, Dp // 8
    dev = W1t.device
    P = W2p.shape[1]
    steps2 = (H + 3) // 4
    W1pad = torch.zeros(16, hp, dtype=torch.float64, device=dev)
    W1pad[:H] = W1t * (2.0 * LOG2E)
    b1pad = torch.zeros(16, dtype=torch.float64, device=dev)
    b1pad[:H] = b1 * (2.0 * LOG2E)
    W2pad = torch.zeros(hp, P, 16, dtype=torch.float64, device=dev)
    W2pad[:, :, :H] = W2p
    b2q = b2p.clone()
    if lk < 2:                                   # alpha = exp(u / 2 + c0) + 1e-10 = exp2((u / 2 + c0) log2 e) + 1e-10
        W2pad[:, 0, :] *= 0.5 * LOG2E
        b2q[:, 0] = (b2q[:, 0] * 0.5 + AFF_C0) * LOG2E
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    unit1 = 4 * (il & 3) + (il >> 2)
    A1 = torch.stack([W1pad[unit1, EPL * ql + s] for s in range(EPL)])                 # (EPL, 64)
    VW = min(EPL, 4)                                                                   # (16-wide rows: A1[64][2])
    A1 = A1.reshape(EPL // VW, VW, 64).permute(0, 2, 1)                                # lane-major groups of 4 k-steps
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    b1m = b1pad[4 * rr + qq]                                                           # [q][r]
    q2, r2 = il >> 2, il & 3
    T2 = EPL // 2 if P == 2 else (EPL + 3) // 4
    A2, b2m = [], []
    for t in range(T2):
        ok_l = ok_b = None
        if P == 2:
            m_l, p_l = EPL * q2 + 2 * t + (r2 >> 1), r2 & 1
            m_b, p_b = EPL * qq + 2 * t + (rr >> 1), rr & 1
        else:
            m_l, p_l = EPL * q2 + 4 * t + r2, torch.zeros_like(r2)
            m_b, p_b = EPL * qq + 4 * t + rr, torch.zeros_like(rr)
            if EPL < 4:                              # 16-wide rows: rows r >= EPL of every group belong to no element
                ok_l, ok_b = (r2 < EPL).to(torch.float64), (rr < EPL).to(torch.float64)
                m_l, m_b = m_l.clamp(max=hp - 1), m_b.clamp(max=hp - 1)
        for r1 in range(steps2):
            A2.append(W2pad[m_l, p_l, 4 * r1 + ql] if ok_l is None else W2pad[m_l, p_l, 4 * r1 + ql] * ok_l)
        b2m.append(b2q[m_b, p_b] if ok_b is None else b2q[m_b, p_b] * ok_b)
    if bf16x3:
        # GEMM 2 in the bf16 x 3 operand format (csrc/tfk_flow_chain.h: couple_lean3): per tile and lane
        # [W_hi | W_mid] and [W_lo | W_hi], 4 bf16 each (slot i <-> hidden unit 4 i + q); b2 = the weight of unit 15
        W2b = W2pad.clone()
        W2b[:, :, 15] = b2q
        Af = []
        for t in range(T2):
            if P == 2:
                m_l, p_l = EPL * q2 + 2 * t + (r2 >> 1), r2 & 1
            else:
                m_l, p_l = EPL * q2 + 4 * t + r2, torch.zeros_like(r2)
            Af.append(torch.stack([W2b[m_l, p_l, 4 * i + ql] for i in range(4)], dim=1))      # (64, 4)
        hi, mid, lo = _bf16_pieces(torch.stack(Af))                                            # (T2, 64, 4)
        pack2 = lambda v: (v[..., 0::2] | (v[..., 1::2] << 16))
        A23 = torch.stack([torch.cat([pack2(hi), pack2(mid)], dim=-1),
                           torch.cat([pack2(lo), pack2(hi)], dim=-1)], dim=1).to(torch.int32)  # (T2, 2, 64, 4)
        return torch.cat([A1.reshape(-1).float(), b1m.reshape(-1).float(), A23.reshape(-1).view(torch.float32),
                          pre_s.float(), pre_t.float()])
    nA2 = (T2 * steps2 + 3) & ~3
    A2 = torch.stack(A2 + [torch.zeros(64, dtype=torch.float64, device=dev)] * (nA2 - T2 * steps2))
    A2 = A2.reshape(nA2 // 4, 4, 64).permute(0, 2, 1)
    tail_ = []
    if W1c is not None:            # the context's columns of W1: A1c[lane][k] = W1c[unit1(i), context element 4 k + q]
        W1cp = torch.zeros(16, 16, dtype=torch.float64, device=dev)
        W1cp[:H, :W1c.shape[1]] = W1c * (2.0 * LOG2E)
        kk = torch.arange(4, device=dev)
        tail_.append(W1cp[unit1.view(64, 1), 4 * kk.view(1, 4) + ql.view(64, 1)].reshape(-1))
    return torch.cat([A1.reshape(-1), b1m.reshape(-1), A2.reshape(-1), torch.stack(b2m).reshape(-1),
                      pre_s, pre_t] + tail_).float()


def rqs_bf16x3_enabled() -> bool:
    return debug_switch("rqs_bf16x3", "1") != "0"


def _bf16_pieces(w: torch.Tensor):
    """fp32 -> three bf16 pieces by truncation (hi = w & 0xffff0000, mid and lo likewise from the exact remainders),
    each returned as the int32 holding its 16 significant bits in the LOW half."""
    w = w.float().contiguous()
    mask = torch.tensor(-65536, dtype=torch.int32, device=w.device)             # 0xffff0000
    hi = (w.view(torch.int32) & mask).view(torch.float32)
    r1 = w - hi
    mid = (r1.view(torch.int32) & mask).view(torch.float32)
    r2 = (r1 - mid).contiguous()
    lo = (r2.view(torch.int32) & mask).view(torch.float32)
    return [(t.contiguous().view(torch.int32) >> 16) & 0xFFFF for t in (hi, mid, lo)]


def _pack_lean_rqs(H: int, Dp: int, W1t, b1, W2p, b2p, pre_s, pre_t, c_delta: float, bf16x3: bool = False,
                   lrs: bool = False, W1c=None) -> torch.Tensor:
    """Parameter block of a lean RQ-spline coupling op (csrc/tfk_flow_rqs_chain.h), fp64 in, fp32 out.
    fp32 operands (hidden width <= 16): head A1[EPL/4][64][4] | b1[4][4] | pre_s[hp] | pre_t[hp], then EPL/8 chunks
    A2[48][64][4] | b2[48][4][4].  bf16 x 3 operands (hidden width <= 15, or <= 31 with HT = 2 hidden tiles): head
    A1[EPL/4][HT][64][4] | b1[HT][4][4] | pre_s | pre_t, then EPL HT / 4 chunks A[4/HT][6][HT][2][64][4 dwords].
    Per target element 24 parameters, all times log2(e): [0, 8) u_x, [8, 16) u_x + u_y / 1000 (the reference's height
    logits, rational_quadratic.py:76), [16, 23) c + u_d / 1000 (:77, c = boundary_u_delta), pad.
    ``lrs`` (bf16 x 3 only): linear rational spline, 32 parameters = 8 tiles per element (linear_rational.py:49-58):
    [0, 8) u_x, [8, 16) u_x + u_y / 100, [16, 24) -u_lambda, [24, 31) c + u_d / 100, [31] u_w0, times log2(e)."""
    hp, EPL = Dp // 2, Dp // 8
    dev = W1t.device
    HT = 2 if (bf16x3 and H > 15) else 1
    HU = 16 * HT                                                   # hidden units the kernel sees
    W1pad = torch.zeros(HU, hp, dtype=torch.float64, device=dev)
    W1pad[:H] = W1t * (2.0 * LOG2E)
    b1pad = torch.zeros(HU, dtype=torch.float64, device=dev)
    b1pad[:H] = b1 * (2.0 * LOG2E)
    TPE = 8 if lrs else 6                                          # tiles of 4 parameters per element
    Q = torch.zeros(hp, 4 * TPE, HU, dtype=torch.float64, device=dev)
    bq = torch.zeros(hp, 4 * TPE, dtype=torch.float64, device=dev)
    if lrs:
        assert bf16x3
        for dst, w in ((Q[:, :, :H], W2p), (bq, b2p)):
            dst[:, 0:8] = w[:, 0:8] * LOG2E
            dst[:, 8:16] = (w[:, 0:8] + w[:, 8:16] / 100.0) * LOG2E
            dst[:, 16:24] = -w[:, 16:24] * LOG2E
            dst[:, 24:31] = w[:, 24:31] / 100.0 * LOG2E
            dst[:, 31] = w[:, 31] * LOG2E
        bq[:, 24:31] += c_delta * LOG2E
    else:
        Q[:, 0:8, :H] = W2p[:, 0:8] * LOG2E
        Q[:, 8:16, :H] = (W2p[:, 0:8] + W2p[:, 8:16] / 1000.0) * LOG2E
        Q[:, 16:23, :H] = W2p[:, 16:23] / 1000.0 * LOG2E
        bq[:, 0:8] = b2p[:, 0:8] * LOG2E
        bq[:, 8:16] = (b2p[:, 0:8] + b2p[:, 8:16] / 1000.0) * LOG2E
        bq[:, 16:23] = (c_delta + b2p[:, 16:23] / 1000.0) * LOG2E
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    unit1 = 4 * (il & 3) + (il >> 2)
    # A1[g][t][lane][k]: source k-step 4 g + k, hidden tile t, D-row i <-> unit 16 t + unit1(i)
    A1 = torch.stack([torch.stack([W1pad[16 * t + unit1, EPL * ql + s_] for t in range(HT)]) for s_ in range(EPL)])
    A1 = A1.reshape(EPL // 4, 4, HT, 64).permute(0, 2, 3, 1)                          # (EPL/4, HT, 64, 4)
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    b1m = torch.stack([b1pad[16 * t + 4 * rr + qq] for t in range(HT)])                # [t][q][r]
    q2, r2 = il >> 2, il & 3
    parts = [A1.reshape(-1), b1m.reshape(-1), pre_s, pre_t]
    if W1c is not None:            # the context's columns of W1: A1c[t][lane][k] = W1c[unit 16 t + unit1(i), context 4 k + q]
        assert bf16x3
        W1cp = torch.zeros(HU, 16, dtype=torch.float64, device=dev)
        W1cp[:H, :W1c.shape[1]] = W1c * (2.0 * LOG2E)
        kk = torch.arange(4, device=dev)
        A1c = torch.stack([W1cp[(16 * t + unit1).view(64, 1), 4 * kk.view(1, 4) + ql.view(64, 1)] for t in range(HT)])
        parts.append(A1c.reshape(-1))
    e_all = torch.arange(EPL, device=dev).view(EPL, 1, 1, 1, 1)                        # the lane-group's element index
    c = torch.arange(TPE, device=dev).view(1, TPE, 1, 1, 1)
    m = EPL * q2.view(1, 1, 1, 64, 1) + e_all                                          # physical target element
    if bf16x3:
        Q[:, :, HU - 1] = bq                                   # the bias as the weight of the last hidden unit (= 1)
        th = torch.arange(HT, device=dev).view(1, 1, HT, 1, 1)
        i4 = torch.arange(4, device=dev).view(1, 1, 1, 1, 4)
        Wf = Q[m, 4 * c + r2.view(1, 1, 1, 64, 1), 16 * th + 4 * i4 + ql.view(1, 1, 1, 64, 1)]   # (EPL, TPE, HT, 64, 4)
        hi, mid, lo = _bf16_pieces(Wf)
        pack2 = lambda v: (v[..., 0::2] | (v[..., 1::2] << 16))
        a1 = torch.cat([pack2(hi), pack2(mid)], dim=-1)
        a2 = torch.cat([pack2(lo), pack2(hi)], dim=-1)
        A3 = torch.stack([a1, a2], dim=3).to(torch.int32)                              # (EPL, TPE, HT, 2, 64, 4)
        return torch.cat([torch.cat(parts).float(), A3.reshape(-1).view(torch.float32)])
    r1 = torch.arange(4, device=dev).view(1, 1, 1, 1, 4)
    A2 = Q[m, 4 * c + r2.view(1, 1, 1, 64, 1), 4 * r1 + ql.view(1, 1, 1, 64, 1)][:, :, 0]   # (EPL, 6, 64, 4)
    mb = EPL * qq.view(1, 1, 4, 4) + torch.arange(EPL, device=dev).view(EPL, 1, 1, 1)
    b2 = bq[mb, 4 * torch.arange(6, device=dev).view(1, 6, 1, 1) + rr.view(1, 1, 4, 4)]       # (EPL, 6, 4, 4)
    for k in range(EPL // 8):
        parts += [A2[8 * k:8 * k + 8].reshape(-1), b2[8 * k:8 * k + 8].reshape(-1)]
    return torch.cat(parts).float()


def _lean_made(layer, d: int, pos: torch.Tensor, D: int, Dp: int):
    """Physical-order weights of a MADE-based affine layer's PARALLEL map (the checks of ``_made_op``), fp64:
    (divide, H, W1p (H, Dp), b1, W2p (Dp, 2, H), b2p (Dp, 2)) or None."""
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import MADE
    if d == layer._sequential_when or layer.context_shape is not None:
        return None
    kind = layer.transformer.native_kind
    ct = layer.conditioner_transform
    if kind not in ("affine", "inverse_affine", "rqs", "lrs") or ct.n_global_parameters != 0:
        return None
    if kind in ("rqs", "lrs") and layer.transformer.n_bins != 8:
        return None
    P = {"rqs": 23, "lrs": 32}.get(kind, 2)
    if ct.output_lower_bound != float("-inf") or ct.output_upper_bound != float("inf"):
        return None
    mods = list(ct.sequential)
    if not (len(mods) == 3 and isinstance(mods[0], MADE.MaskedLinear) and isinstance(mods[1], nn.Tanh)
            and isinstance(mods[2], MADE.MaskedLinear)):
        return None
    H = mods[0].out_features
    if mods[0].in_features != D or H > (15 if P != 2 else 16):
        return None
    W1 = (mods[0].weight * mods[0].mask).detach().double()                     # (H, D) logical columns
    W2 = (mods[2].weight * mods[2].mask).detach().double().view(D, P, H)       # logical element, parameter, unit
    W1p = W1.new_zeros(H, Dp)
    W1p[:, pos] = W1
    W2p = W2.new_zeros(Dp, P, H)
    W2p[pos] = W2
    b2p = W2.new_zeros(Dp, P)
    b2p[pos] = mods[2].bias.detach().double().view(D, P)
    return (kind if P != 2 else kind == "inverse_affine"), H, W1p, mods[0].bias.detach().double(), W2p, b2p


def _pack_lean_made_spline(H: int, Dp: int, W1f, b1f, W2p, b2p, pre_s, pre_t, c_delta: float, lrs: bool) -> torch.Tensor:
    """Parameter block of a lean MADE spline op (csrc/tfk_flow_rqs_chain.h: rqs_made_layer3), bf16 x 3 operands, hidden
    width <= 15: head A1[2 EPL / 4][64][4] (plane A's k-steps, then plane B's) | b1[4][4] | pre_s[Dp] | pre_t[Dp], then the
    chunks of plane A's elements and of plane B's, each exactly as a lean spline coupling's (``_pack_lean_rqs``)."""
    hp, EPL = Dp // 2, Dp // 8
    dev = W1f.device
    W1pad = torch.zeros(16, Dp, dtype=torch.float64, device=dev)
    W1pad[:H] = W1f * (2.0 * LOG2E)
    b1pad = torch.zeros(16, dtype=torch.float64, device=dev)
    b1pad[:H] = b1f * (2.0 * LOG2E)
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    unit1 = 4 * (il & 3) + (il >> 2)
    A1 = torch.stack([W1pad[unit1, (s_ // EPL) * hp + EPL * ql + (s_ % EPL)] for s_ in range(2 * EPL)])   # (2 EPL, 64)
    A1 = A1.reshape(2 * EPL // 4, 4, 64).permute(0, 2, 1)
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    b1m = b1pad[4 * rr + qq]
    head_len = EPL * 64 + 16 + 2 * hp                    # a coupling block's head (dropped below)
    zero_w1 = torch.zeros(H, hp, dtype=torch.float64, device=dev)
    chunks = []
    for plane in (0, 1):
        sl = slice(plane * hp, (plane + 1) * hp)
        blk = _pack_lean_rqs(H, Dp, zero_w1, b1f, W2p[sl], b2p[sl], pre_s[sl], pre_t[sl], c_delta, bf16x3=True, lrs=lrs)
        chunks.append(blk[head_len:])
    head = torch.cat([A1.reshape(-1), b1m.reshape(-1), pre_s, pre_t]).float()
    return torch.cat([head] + chunks)


def _pack_lean_made(H: int, Dp: int, W1f, b1f, W2p, b2p, pre_s, pre_t) -> torch.Tensor:
    """Parameter block of a lean MADE op (csrc/tfk_flow_chain.h: made_lean): A1[2 EPL / 4][64][4] (plane A's k-steps,
    then plane B's) | b1[4][4] | A2[nA2 / 4][64][4] | b2[EPL][4][4] | pre_s[Dp] | pre_t[Dp], weights pre-scaled as
    for the lean couplings.  Inputs fp64, physical order, pending elementwise maps already folded into W1f / b1f."""
    hp, EPL = Dp // 2, Dp // 8
    dev = W1f.device
    steps2 = (H + 3) // 4
    W1pad = torch.zeros(16, Dp, dtype=torch.float64, device=dev)
    W1pad[:H] = W1f * (2.0 * LOG2E)
    b1pad = torch.zeros(16, dtype=torch.float64, device=dev)
    b1pad[:H] = b1f * (2.0 * LOG2E)
    W2pad = torch.zeros(Dp, 2, 16, dtype=torch.float64, device=dev)
    W2pad[:, :, :H] = W2p
    W2pad[:, 0, :] *= 0.5 * LOG2E
    b2q = b2p.clone()
    b2q[:, 0] = (b2q[:, 0] * 0.5 + AFF_C0) * LOG2E
    lane = torch.arange(64, device=dev)
    ql, il = lane >> 4, lane & 15
    unit1 = 4 * (il & 3) + (il >> 2)
    A1 = torch.stack([W1pad[unit1, plane * hp + EPL * ql + s_] for plane in range(2) for s_ in range(EPL)])
    A1 = A1.reshape(2 * EPL // 4, 4, 64).permute(0, 2, 1)
    qq, rr = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    b1m = b1pad[4 * rr + qq]
    q2, r2 = il >> 2, il & 3
    A2, b2m = [], []
    for t in range(EPL):
        plane, tt = (0, t) if t < EPL // 2 else (1, t - EPL // 2)
        for r1 in range(steps2):
            A2.append(W2pad[plane * hp + EPL * q2 + 2 * tt + (r2 >> 1), r2 & 1, 4 * r1 + ql])
        b2m.append(b2q[plane * hp + EPL * qq + 2 * tt + (rr >> 1), rr & 1])
    nA2 = (EPL * steps2 + 3) & ~3
    A2 = torch.stack(A2 + [torch.zeros(64, dtype=torch.float64, device=dev)] * (nA2 - EPL * steps2))
    A2 = A2.reshape(nA2 // 4, 4, 64).permute(0, 2, 1)
    return torch.cat([A1.reshape(-1), b1m.reshape(-1), A2.reshape(-1), torch.stack(b2m).reshape(-1), pre_s, pre_t]).float()


def _compile_lean(composition, plan, device, D: int, Dp: int, pos: torch.Tensor, pos_in: torch.Tensor,
                  context: bool = False, odd: bool = False, allow_aff3: bool = True):
    """The chain as LEAN flow programs (csrc/tfk_flow_chain.h), or None: elementwise layers with global parameters,
    folded reversals and affine / shift couplings of one kind and one hidden width <= 16 whose source plane
    alternates -- every RealNVP / NICE preset.  The elementwise layers are deferred: physical column c carries a
    pending map x -> s[c] x + t[c] (fp64 on the host) that is folded into W1 / b1 where c feeds a conditioner,
    applied by the coupling that transforms c (its pre-affine), and flushed by one TFK_OP_EW_FMA at the end together
    with the sum of the constant log-dets.
    ``context`` (conditional flows; spline chains only): the couplings' conditioners read [x_A | context] -- the
    context's columns of W1 are further k-steps of GEMM 1 --, and the elementwise layers whose parameters are a Linear map
    of the context (the presets' first and second-to-last layer) run as interpreter ops (TFK_OP_EWC_*) in a launch of
    their own before / after the lean chain: 3 launches for a conditional CouplingRQNSF instead of one per coupling."""
    from torchflows_amd.bijections.finite.autoregressive.layers_base import (
        CouplingBijection, ElementwiseBijection, MaskedAutoregressiveBijection)
    from torchflows_amd.bijections.finite.matrix.permutation import PermutationMatrix
    hp = Dp // 2
    pos_arg = pos                                            # (the walk below rebinds ``pos`` at every permutation)
    if odd:
        # ODD event sizes (affine / shift chains): HalfSplit has one target more than sources, and with the reversals the
        # MIDDLE element is a target of every coupling -- it has to sit in whichever plane is being transformed.  Both
        # planes reserve their last column for it (the caller's layout puts it there); before a coupling that finds it in
        # its SOURCE plane the kernel takes it over (bit 2 of the op's src_plane, csrc/tfk_flow_chain.h: move_middle), and
        # its pending elementwise map moves with it here.
        if context or (D + 1) // 2 > hp:
            return None
        pos = pos.clone()
    s = torch.ones(Dp, dtype=torch.float64, device=device)
    t = torch.zeros(Dp, dtype=torch.float64, device=device)
    ld_const = torch.zeros((), dtype=torch.float64, device=device)
    items = []          # (lean kind, plane, steps2, block)
    head, tail, closed_flush = [], [], None                  # (context) interpreter ops around the chain
    ctx_bits = 0
    kind0 = steps0 = None
    # bf16 x 3 operands for affine / shift chains: D = 64 only, and the whole chain must stay ONE launch (10.6 KB of
    # operands per coupling instead of 5.7): decided after a dry count of the couplings
    n_couplings = sum(isinstance(layer, CouplingBijection) for layer, _ in plan)
    aff3 = allow_aff3 and lean_bf16x3_enabled(Dp) and not odd and ((Dp == 64 and 0 < n_couplings <= 13)
                                                    or (Dp == 256 and not context and stream_chain_enabled()))
    # (D = 256, round 4: the streamed chain takes the bf16 x 3 format too -- 42 KB per coupling, two blocks resident)
    # (context) if a context-conditioned elementwise layer precedes the first coupling, every elementwise layer before
    # that coupling is an interpreter op of the launch in front of the chain (the inverse direction starts with a
    # constant ActNorm followed by the context-conditioned layer)
    first_c = next((i for i, (layer, _) in enumerate(plan) if isinstance(layer, (CouplingBijection, MaskedAutoregressiveBijection))), len(plan))
    head_mode = context and any(isinstance(layer, ElementwiseBijection) and not layer.use_global_parameters
                                for layer, _ in plan[:first_c])
    # affine / shift chains take those elementwise layers INSIDE the lean launch (csrc/tfk_flow_chain.h: side_op): in front
    # of the couplings and behind the closing TFK_OP_EW_FMA, up to 3 ops per side -- one launch per conditional log_prob
    kind_c = (plan[first_c][0].transformer.native_kind
              if first_c < len(plan) and isinstance(plan[first_c][0], CouplingBijection) else None)
    inline = (context and (kind_c in ("affine", "inverse_affine", "shift")
                           or (kind_c in ("rqs", "lrs") and Dp <= 128 and rqs_bf16x3_enabled()))
              )
    if inline and Dp == 256 and kind_c not in ("rqs", "lrs"):
        # the context variant of the 256-wide chain kernel spills (660 B of scratch at the 256-VGPR cap): measured
        # 1.8e8 evals/s against the interpreter's 3.5e8 on a conditional RealNVP(256) -- the interpreter keeps that size
        return None
    pre_items, post_items = [], []

    def pending_block():                                     # the pending maps as an EW_FMA block, then reset
        nonlocal ld_const
        blk = torch.cat([s, t, ld_const.reshape(1), ld_const.new_zeros(3)]).float()
        s.fill_(1.0)
        t.fill_(0.0)
        ld_const = torch.zeros((), dtype=torch.float64, device=device)
        return blk

    def pending_is_identity():
        return bool((s == 1).all()) and bool((t == 0).all()) and float(ld_const) == 0.0
    with torch.no_grad():
        for li, (layer, d) in enumerate(plan):
            if isinstance(layer, PermutationMatrix):
                perm = (layer._fwd_index if d == FORWARD else layer._inv_index).to(device)
                pos = pos[perm]
            elif inline and isinstance(layer, ElementwiseBijection) and not layer.use_global_parameters:
                item = _elementwise_ctx_op(layer, d, pos, D, Dp, lean=True)
                if item is None:
                    return None
                ewc = (item[0][0], item[0][1], 0, item[1].float(), ())
                if not items and closed_flush is None:       # in front of the couplings: what is pending goes first
                    if not pending_is_identity():
                        pre_items.append((OP_EW_FMA, 0, 0, pending_block(), ()))
                    pre_items.append(ewc)
                else:                                        # behind them: close the chain (once), then the op
                    if closed_flush is None:
                        closed_flush = pending_block()
                    elif not pending_is_identity():
                        post_items.append((OP_EW_FMA, 0, 0, pending_block(), ()))
                    post_items.append(ewc)
                if len(pre_items) > 3 or len(post_items) > 2:
                    return None
            elif not inline and isinstance(layer, ElementwiseBijection) and head_mode and li < first_c:
                item = (_elementwise_ctx_op(layer, d, pos, D, Dp) if not layer.use_global_parameters
                        else _elementwise_op(layer, d, pos, D, Dp))
                if item is None:
                    return None
                head.append(item)
            elif isinstance(layer, ElementwiseBijection):
                if context and not layer.use_global_parameters:
                    item = _elementwise_ctx_op(layer, d, pos, D, Dp)
                    if item is None:
                        return None
                    if not items and closed_flush is None:   # before the chain: nothing may be pending
                        if not (bool((s == 1).all()) and bool((t == 0).all())):
                            return None
                        head.append(item)
                    else:                                    # after it: the chain ends here, pending maps flushed
                        if closed_flush is None:
                            closed_flush = torch.cat([s, t, ld_const.reshape(1), ld_const.new_zeros(3)]).float()
                        tail.append(item)
                    continue
                if closed_flush is not None and not inline:  # constant elementwise layers behind the chain: interpreter ops
                    item = _elementwise_op(layer, d, pos, D, Dp)
                    if item is None:
                        return None
                    tail.append(item)
                    continue
                got = _lean_elementwise(layer, d, D)
                if got is None:
                    return None
                alpha, beta, divide = got
                if divide:
                    s[pos] = s[pos] / alpha
                    t[pos] = (t[pos] - beta) / alpha
                    ld_const = ld_const - torch.log(alpha).sum()
                else:
                    s[pos] = alpha * s[pos]
                    t[pos] = alpha * t[pos] + beta
                    ld_const = ld_const + torch.log(alpha).sum()
            elif isinstance(layer, MaskedAutoregressiveBijection):
                if context or Dp < 32:
                    return None                              # (odd event sizes: any layout serves a MADE layer -- the
                                                             # conditioner reads and transforms every element; no moves)
                got = _lean_made(layer, d, pos, D, Dp)       # MAF density / IAF sampling: the parallel map
                if got is None:
                    return None
                divide, H, W1p, b1, W2p, b2p = got
                if divide in ("rqs", "lrs"):                 # MADE spline layers: the single-launch spline chain kernel
                    if not rqs_bf16x3_enabled() or Dp not in (64, 128):
                        return None
                    tr = layer.transformer
                    lrs_ = divide == "lrs"
                    lk, steps2 = (12 if lrs_ else 10), (H + 1 + 3) // 4
                    if lrs_:
                        extra = (8 + 256, float(tr.boundary), float(np.float32(1.0 - tr.min_bin_width * tr.n_bins)),
                                 float(np.float32(tr.const)))
                        c_delta = float(np.float32(tr.const))
                    else:
                        extra = (8 + 256, float(tr.boundary), float(np.float32(1.0 - tr.min_bin_size * tr.n_bins)),
                                 float(np.float32(tr.boundary_u_delta)))
                        c_delta = float(np.float32(tr.boundary_u_delta))
                    if kind0 is None:
                        kind0, steps0 = lk, steps2
                    elif (lk, steps2) != (kind0, steps0) or items[-1][4] != extra:
                        return None
                    block = _pack_lean_made_spline(H, Dp, W1p * s, b1 + W1p @ t, W2p, b2p, s.clone(), t.clone(), c_delta, lrs_)
                    s.fill_(1.0)
                    t.fill_(0.0)
                    items.append((OP_MADE_LRS_FWD_LEAN if lrs_ else OP_MADE_RQS_FWD_LEAN, 0, steps2, block, extra))
                    continue
                lk, steps2 = (7 if divide else 6), (H + 3) // 4
                if kind0 is None:
                    kind0, steps0 = lk, steps2
                elif (lk, steps2) != (kind0, steps0):
                    return None
                block = _pack_lean_made(H, Dp, W1p * s, b1 + W1p @ t, W2p, b2p, s.clone(), t.clone())
                s.fill_(1.0)
                t.fill_(0.0)
                items.append((OP_MADE_INV_LEAN if divide else OP_MADE_FWD_LEAN, 0, steps2, block, ()))
            elif isinstance(layer, CouplingBijection):
                if closed_flush is not None:
                    return None                               # (a coupling behind a context-conditioned elementwise layer)
                moved = 0
                if odd:
                    S_ = layer.coupling.source_event_size
                    src_plane = int(pos[0].item()) // hp
                    wrong = ((pos[S_:] // hp) == src_plane).nonzero().flatten()
                    if wrong.numel() > 1:
                        return None
                    if wrong.numel() == 1:                    # the middle element sits in the source plane: take it over
                        l_mid = S_ + int(wrong[0].item())
                        old, new = int(pos[l_mid].item()), (1 - src_plane) * hp + hp - 1
                        if old != src_plane * hp + hp - 1:
                            return None
                        pos[l_mid] = new
                        s[new], t[new] = s[old].clone(), t[old].clone()
                        s[old], t[old] = 1.0, 0.0
                        moved = 4
                got = _lean_coupling(layer, d, pos, D, Dp, allow_ctx=context)
                if got is None:
                    return None
                W1c = None
                if context:
                    lk, plane, H, W1t, b1, W2p, b2p, W1c = got
                    if lk in (4, 5, 8, 9) and not rqs_bf16x3_enabled():
                        return None                           # (lean context spline programs: bf16 x 3 operands)
                    cs_l = 0 if W1c is None else (W1c.shape[1] + 3) // 4
                    if items and (cs_l << 4) != ctx_bits:
                        return None
                    ctx_bits = cs_l << 4
                else:
                    lk, plane, H, W1t, b1, W2p, b2p = got
                if kind0 is not None and kind0 in (6, 7, 10, 12):
                    return None                               # (couplings and MADE layers do not share a program)
                steps2 = (H + 3) // 4
                if Dp < 32 and lk >= 4:
                    return None                               # (16-wide rows: affine / shift chains only)
                if odd and lk >= 4 and not rqs_bf16x3_enabled():
                    return None                               # (odd event sizes of spline chains: the bf16 x 3 format)
                if lk >= 8 and not rqs_bf16x3_enabled():
                    return None                               # (linear rational splines: bf16 x 3 operands only)
                if lk >= 4 and rqs_bf16x3_enabled():
                    steps2 = (H + 1 + 3) // 4                 # bf16 x 3 operands: counts the bias unit; > 4 = two hidden tiles
                if kind0 is None:
                    kind0, steps0 = lk, steps2
                elif (lk, steps2) != (kind0, steps0) or plane != 1 - (items[-1][1] & 1):
                    return None

                src = torch.arange(plane * hp, (plane + 1) * hp, device=device)
                tgt = torch.arange((1 - plane) * hp, (2 - plane) * hp, device=device)
                b1f = b1 + W1t @ t[src]                      # W1 (s x + t) + b1 = (W1 s) x + (W1 t + b1)
                W1f = W1t * s[src]
                if lk >= 8:                                  # linear rational spline: as the RQ spline, 32 parameters
                    tr = layer.transformer
                    extra = (8 + 256, float(tr.boundary), float(np.float32(1.0 - tr.min_bin_width * tr.n_bins)),
                             float(np.float32(tr.const)))
                    if items and items[-1][4] != extra:
                        return None
                    block = _pack_lean_rqs(H, Dp, W1f, b1f, W2p, b2p, s[tgt].clone(), t[tgt].clone(),
                                           float(np.float32(tr.const)), bf16x3=True, lrs=True, W1c=W1c)
                    items.append((OP_LRS_FWD_LEAN + lk - 8, plane | moved, steps2, block, extra))
                elif lk >= 4:                                # RQ spline: one launch for the chain, operands streamed
                    tr = layer.transformer
                    fmt3 = rqs_bf16x3_enabled()                       # (the last hidden unit carries the bias: H <= 31)
                    if not fmt3 and (H > 16 or Dp < 64):
                        return None                                   # (fp32 operands: chunks of 8 elements per lane group)

                    extra = (8 + (256 if fmt3 else 0), float(tr.boundary),
                             float(np.float32(1.0 - tr.min_bin_size * tr.n_bins)), float(np.float32(tr.boundary_u_delta)))
                    if items and items[-1][4] != extra:
                        return None
                    block = _pack_lean_rqs(H, Dp, W1f, b1f, W2p, b2p, s[tgt].clone(), t[tgt].clone(),
                                           float(np.float32(tr.boundary_u_delta)), bf16x3=fmt3, W1c=W1c)
                    items.append((OP_RQS_FWD_LEAN + lk - 4, plane | moved, steps2, block, extra))
                else:
                    use3 = aff3 and H <= 15 and not context
                    if items and bool(items[-1][4]) != use3:
                        return None
                    block = _pack_lean(lk, H, Dp, W1f, b1f, W2p, b2p, s[tgt].clone(), t[tgt].clone(), bf16x3=use3, W1c=W1c)
                    items.append((OP_AFFINE_FWD_LEAN + lk, plane | moved, steps2, block, (256,) if use3 else ()))
                s[tgt] = 1.0
                t[tgt] = 0.0
            else:
                return None
    if inline and closed_flush is not None and not pending_is_identity():
        post_items.append((OP_EW_FMA, 0, 0, pending_block(), ()))        # constant layers behind the last context op
    flush = closed_flush if closed_flush is not None else torch.cat([s, t, ld_const.reshape(1), ld_const.new_zeros(3)]).float()
    streamed = kind0 in (4, 5, 8, 9, 10, 12)             # spline chains read their operands from global memory
    if (not items and streamed) or (context and not items):
        return None
    items.append((OP_EW_FMA, 0, 0, flush, ()))
    n_pre = len(pre_items)
    items = pre_items + items + post_items
    budget = max(MAX_PARAM_BYTES_MFMA, MFMA_BUDGET_WIDE.get(Dp, 0))
    if any(extra == (256,) for *_, extra in items):
        budget = 150 * 1024          # bf16 x 3 operands: one 1024-thread workgroup per CU holds the whole chain
    if Dp == 64 and kind0 in (0, 1, 2, 3):
        # the 64-wide chain kernel is register-bound at 4 waves per SIMD = TWO 512-thread workgroups per CU: up to
        # 76 KB of operands per launch cost no occupancy (12 couplings, or 8 with their context columns)
        budget = max(budget, LEAN_BUDGET_64)
    if Dp == 128:
        # the 128-wide chain kernel runs one 768-thread workgroup per CU whatever the block's size: RealNVP(128, 8 layers)
        # = 90.6 KB is ONE launch (at the interpreter's 76 KB budget it was two, and the rows' trip through HBM between
        # them -- 0.9 GB per launch at 2.8 TB/s -- bound both: 2 x 316 us)
        budget = 150 * 1024
    # affine / shift chains whose blocks do not fit the LDS together (D = 256: 22 KB per coupling): ONE launch with the
    # operands streamed block by block (csrc/tfk_flow_chain.h: chain_layers_stream) instead of one launch per LDS-full
    total = sum(block.numel() for _, _, _, block, _ in items) * 4
    fmt3_all = all(extra == (256,) for k_, _, _, _, extra in items if k_ != OP_EW_FMA)
    if (not streamed and not context and kind0 in (0, 1, 2, 3) and Dp >= 128 and stream_chain_enabled() and len(items) <= 61
            and total > 158 * 1024 and not odd
            and (not any(extra for *_, extra in items) or (Dp == 256 and fmt3_all))):
        streamed = True
    elif Dp == 256 and any(extra == (256,) for *_, extra in items):
        # bf16 x 3 blocks at D = 256 exist as STREAMED operands only: a chain short enough to keep its blocks resident
        # (three couplings) is packed again in the fp32 format
        return _compile_lean(composition, plan, device, D, Dp, pos_arg, pos_in, context=context, odd=odd, allow_aff3=False)
    segments: List[Segment] = []
    ops, blocks, used = [], [], 0
    for kind, plane, steps2, block, extra in items:
        n = block.numel()
        if not streamed:
            if n * 4 > 150 * 1024:
                return None
            over = (used + n) * 4 > budget
            if ops and kind not in (OP_EW_FMA, OP_EWC_MULADD, OP_EWC_SUBDIV) and len(ops) > n_pre and (over or len(ops) >= 60):
                segments.append(Segment(ops, torch.cat(blocks).contiguous(), True))
                ops, blocks, used = [], [], 0
        ops.append((kind, plane | (ctx_bits if kind not in (OP_EW_FMA, OP_EWC_MULADD, OP_EWC_SUBDIV) else 0), steps2, used)
                   + tuple(extra))
        blocks.append(block)
        used += n
    segments.append(Segment(ops, torch.cat(blocks).contiguous(), True))

    def small_segment(its):                              # a few interpreter ops (elementwise layers) as one launch
        o, b, u = [], [], 0
        for head_, block_ in its:
            o.append((head_[0], head_[1], head_[2], u) + tuple(head_[3:]))
            b.append(block_.float())
            u += block_.numel()
        return Segment(o, torch.cat(b).contiguous(), True)
    if head:
        segments.insert(0, small_segment(head))
    if tail:
        segments.append(small_segment(tail))
    identity = bool(torch.equal(pos, pos_in))
    return CompiledChain(Dp, segments, pos, identity, _params_version(composition), D_log=D,
                         pos_in=pos_in if Dp != D else None)


def compile_chain(composition, direction: int, device: torch.device,
                  mfma: Optional[bool] = None, context: bool = False) -> Optional[CompiledChain]:
    """Flow programs for ``composition.forward`` (direction 0) or ``.inverse`` (1), or None.
    ``mfma`` None: use the matrix-core kernel when the chain qualifies (D in {64, 128}, affine /
    shift couplings, hidden width <= 16), else the vector-ALU one."""
    from torchflows_amd.bijections.finite.autoregressive.layers_base import (
        CouplingBijection, ElementwiseBijection, MaskedAutoregressiveBijection)
    from torchflows_amd.bijections.finite.matrix.permutation import PermutationMatrix

    D = composition.n_dim
    if not enabled():
        return None
    # even event sizes that are not 64 / 128 / 256: both halves are padded to the next supported plane width
    # (the rows are padded on the way in, run_chain) -- matrix-core programs with affine / shift couplings only
    Dp = D
    if not native.lib().tfk_flow_mfma_supported(D) and D % 2 == 0 and 4 <= D < 256:
        Dp = 64 if D < 64 else (128 if D < 128 else 256)
    # odd event sizes: HalfSplit moves one element across the halves at every reversal, so every element gets
    # its own index in both planes (plane width >= D) and changes planes by TFK_OP_PLANE_SWAP
    slots = D % 2 == 1 and 3 <= D <= 64          # (the swap op is not built for 256-wide rows)
    if slots:
        Dp = 64 if D <= 32 else 128
    # odd event sizes above 64: the straight-line kernels only (the middle element changes planes there, _compile_lean)
    odd_wide = D % 2 == 1 and 64 < D < 256
    if odd_wide:
        Dp = 128 if D < 128 else 256
    if mfma is None:
        if mfma_enabled() and native.lib().tfk_flow_mfma_supported(Dp) and padded_enabled(D, Dp):
            chain = compile_chain(composition, direction, device, mfma=True, context=context)
            if chain is not None:
                return chain
        mfma = False
    if context and not mfma:
        return None                              # context-conditioned programs: matrix-core interpreter only
    if not mfma:
        Dp = D
    elif Dp != D and not padded_enabled(D, Dp):
        return None
    if not native.lib().tfk_flow_supported(Dp):
        return None
    order = composition.layers if direction == FORWARD else list(composition.layers)[::-1]
    plan = _flatten(order, "forward" if direction == FORWARD else "inverse")
    if plan is None:
        return None
    def planes(width):                           # logical element -> physical column of a row of `width`
        p_ = torch.arange(D, device=device)
        if width != D and not slots:             # second half of the row starts at the padded plane boundary
            p_ = torch.where(p_ < D // 2, p_, p_ - D // 2 + width // 2)
        return p_
    if mfma and not slots and lean_enabled():
        # event sizes <= 32: the straight-line kernels exist at row width 32 as well (half the work of a 64-wide row)
        # (conditional flows: the elementwise layers around the chain are interpreter launches -- 64 columns at least)
        narrow = D % 2 == 0 and narrow_rows_enabled() and not context
        widths = (([16] if (narrow and 4 <= D <= 16 and rows16_enabled()) else [])
                  + ([32] if (narrow and 4 <= D <= 32) else []) + [Dp])
        for w in widths:
            if w != D and not padded_enabled(D, w):
                continue
            chain = _compile_lean(composition, plan, device, D, w, planes(w), planes(w), context=context)
            if chain is not None:
                return chain
    if mfma and D % 2 == 1 and 3 <= D < 256 and lean_enabled() and odd_lean_enabled() and not context:
        # odd event sizes on the straight-line kernels (affine / shift chains): sources at the head of plane A, the
        # targets behind the middle element at the head of plane B, the middle element in plane B's last column
        h_ = D // 2
        for w in (16, 32, 64, 128, 256):
            if (D + 1) // 2 > w // 2 or (w == 16 and not rows16_enabled()) or (w == 32 and not narrow_rows_enabled()) \
                    or not padded_enabled(D, w):
                continue
            l_ = torch.arange(D, device=device)
            lay = torch.where(l_ < h_, l_, torch.where(l_ == h_, torch.full_like(l_, w - 1), w // 2 + l_ - h_ - 1))
            chain = _compile_lean(composition, plan, device, D, w, lay, lay.clone(), context=False, odd=True)
            if chain is not None:
                return chain
            if w >= 32:
                break                            # (a wider row would not make a chain lean that is not lean here; 16-wide
                                                 # rows are for affine / shift chains only: a spline chain gets 32)
    if odd_wide:
        return None                              # (no interpreter route at these sizes)
    pos = planes(Dp)
    pos_in = pos.clone()                         # (odd sizes: the whole row enters in plane 0, element l at index l)
    items = []                                   # [(op triple, block)]
    with torch.no_grad():
        for layer, d in plan:
            if isinstance(layer, PermutationMatrix):
                perm = (layer._fwd_index if d == FORWARD else layer._inv_index).to(device)
                pos = pos[perm]                  # new logical j = old logical perm[j]
                continue
            if isinstance(layer, ElementwiseBijection):
                if context and not layer.use_global_parameters:
                    item = _elementwise_ctx_op(layer, d, pos, D, Dp)
                else:
                    item = _elementwise_op(layer, d, pos, D, Dp)
            elif isinstance(layer, CouplingBijection):
                if slots and Dp != D:
                    swap, pos = _plane_swap(layer, pos, Dp)
                    if swap is not None:
                        items.append(swap)
                item = _coupling_op(layer, d, pos, D, mfma=mfma, Dp=Dp)
            elif isinstance(layer, MaskedAutoregressiveBijection):
                item = _made_op(layer, d, pos, D, Dp) if mfma else None
            else:
                item = None
            if item is None:
                return None
            items.append(item)
    # pack into launches whose parameter block fits the LDS budget
    segments: List[Segment] = []
    ops, blocks, used = [], [], 0
    for head, block in items:
        kind, plane, H = head[:3]
        extra = tuple(head[3:])
        n = block.numel()
        if n * 4 > 150 * 1024:
            return None                          # a single op larger than LDS: not fusable here
        # a launch holds as many ops as fit the LDS budget; a coupling op too big for the budget
        # gets a launch of its own, and the small elementwise ops around it ride along
        small = kind in (OP_EW_MULADD, OP_EW_SUBDIV, OP_PLANE_SWAP, OP_EWC_MULADD, OP_EWC_SUBDIV)
        budget = max(MAX_PARAM_BYTES_MFMA, MFMA_BUDGET_WIDE.get(Dp, 0)) if mfma else MAX_PARAM_BYTES
        over = (used + n) * 4 > budget
        if ops and ((over and not (small and (used + n) * 4 <= 150 * 1024)) or len(ops) == MAX_OPS):
            segments.append(Segment(ops, torch.cat(blocks).contiguous(), mfma))
            ops, blocks, used = [], [], 0
        ops.append((kind, plane, H, used) + extra)
        blocks.append(block.float())
        used += n
    if ops:
        segments.append(Segment(ops, torch.cat(blocks).contiguous(), mfma))
    identity = bool(torch.equal(pos, pos_in))
    if Dp != D and not segments:
        return None
    return CompiledChain(Dp, segments, pos, identity, _params_version(composition), D_log=D,
                         pos_in=pos_in if Dp != D else None)


def narrow_enabled() -> bool:
    return debug_switch("narrow_in", "1") != "0"


def stream_chain_enabled() -> bool:
    """One launch with streamed operands for affine / shift chains that do not fit the LDS (TORCHFLOWS_AMD_DEBUG=stream_chain=0:
    one launch per LDS-full of couplings, as before)."""
    return debug_switch("stream_chain", "1") != "0"


def odd_lean_enabled() -> bool:
    """Odd event sizes on the straight-line chain kernels (TORCHFLOWS_AMD_DEBUG=odd_lean=0: the interpreter with a plane per
    element, as before round 3)."""
    return debug_switch("odd_lean", "1") != "0"


def rows16_enabled() -> bool:
    """Row width 16 for affine / shift chains on event sizes <= 16 (TORCHFLOWS_AMD_DEBUG=rows16=0: width 32 as before)."""
    return debug_switch("rows16", "1") != "0"


def narrow_rows_enabled() -> bool:
    """Row width 32 for event sizes <= 32 (always since round 4; the switch that padded them to 64 instead is retired)."""
    return True


def padded_enabled(D: int, Dp: int) -> bool:
    return Dp == D or debug_switch("fused_pad", "1") != "0"


def sample_ready(chain: Optional[CompiledChain]) -> bool:
    """One matrix-core launch: the base density of the incoming rows can ride along (flag bit 2)."""
    return (chain is not None and len(chain.segments) == 1 and chain.segments[0].mfma
            and chain.pos_in is None)


def invalidate(module: nn.Module, compiled_only: bool = False) -> None:
    """Drop EVERY cached packing below ``module``: compiled flow programs (``_tfk_compiled``), MADE weight packs,
    elementwise blocks, BatchNorm scale / shift, image programs, training packs, the flat tensor lists.  The caches are
    keyed on the tensors' autograd version counters, on slot identity (a replaced Parameter) and on a process-wide epoch
    that every ``.to()`` / ``.cuda()`` / dtype conversion of a module of this package bumps -- NOT on data pointers.  An
    in-place edit through ``.data`` (``p.data.mul_(0.5)``, manual weight averaging or clamping), an assignment
    ``p.data = other`` or a replayed hipGraph moves none of these.  After such an
    edit call this (``Bijection.invalidate_native_caches()`` / ``Flow.invalidate_native_caches()``); ``train()`` /
    ``eval()`` and ``load_state_dict`` call it themselves.
    ``compiled_only``: just the compiled flow programs -- what ``Flow.fit`` drops after an epoch of hipGraph replays
    (the captured graph keeps READING the other packs' tensors on every replay, so those must stay allocated)."""
    for m in module.modules():
        d = m.__dict__
        if compiled_only:
            d.pop("_tfk_compiled", None)
            continue
        for k in [k for k in d if k.startswith("_tfk_") and k not in _STRUCTURAL_CACHES]:
            del d[k]


# caches that hold no parameter VALUES -- index maps of the training packs (keyed on layer identity, direction and
# device), the flat tensor-slot list and the dtype / device check -- and therefore never go stale through a value
# edit.  They must survive invalidate(): a captured hipGraph of the training step (Flow.fit, TORCHFLOWS_AMD_GRAPH=1)
# reads the packs' index tensors on every replay, and fit() invalidates after every epoch of replays.
_STRUCTURAL_CACHES = ("_tfk_plan_packs", "_tfk_slots", "_tfk_declined_warned")
# (the (module, name) slot lists of the L2 term and of the autograd node -- "_tfk_l2_slots", "_tfk_param_slots" -- hold no
# tensors and are rebuilt in microseconds: invalidate() drops them like any other cache)


_CACHE_CHECK = int(debug_switch("cache_check", "0") or 0)
_cache_calls = 0

# Edits through ``.data`` (``p.data.mul_(...)``) move no version counter: a compiled program would go on serving the old
# weights in silence -- a hazard the reference does not have (it reads the live tensors on every call).  Default guard,
# WITHOUT a host synchronisation on the hit path: every GUARD_EVERY-th hit of a compiled program on a HIP device enqueues a
# checksum of the live tensors (two multi-tensor kernels), its comparison with the checksum taken when the program was
# packed, and a copy of the verdict into pinned host memory; the verdict is READ on a later hit, once its event has
# completed.  A mismatch drops every packed copy below the composition, warns, and recompiles in the same call.  A stale
# program can therefore serve at most ~2 GUARD_EVERY calls after such an edit (TORCHFLOWS_AMD_DEBUG=guard_every=N; 0 = off);
# ``invalidate_native_caches()`` right after the edit stays the way to serve none.
GUARD_EVERY = int(debug_switch("guard_every", "64") or 0)


class StaleProgramWarning(UserWarning):
    """Parameters changed without their version counters moving (an edit through ``.data``); the packed copies were
    dropped and rebuilt."""


def _device_checksum(module: nn.Module) -> Optional[torch.Tensor]:
    """0-dim fp64 device tensor: a position-weighted checksum of every fp32 parameter / buffer below ``module``."""
    ts = [t.detach() for t in list(module.parameters()) + list(module.buffers())
          if t.is_floating_point() and t.numel() and t.is_cuda]
    if not ts:
        return None
    n1 = torch.stack(torch._foreach_norm(ts, 1)).double()          # sum |x|
    n2 = torch.stack(torch._foreach_norm(ts, 2)).double()          # sqrt(sum x^2)
    w = torch.arange(1, len(ts) + 1, dtype=torch.float64, device=n1.device)
    return (n1 * w).sum() + 3.0 * (n2 * w).sum()


class _Guard:
    __slots__ = ("packed", "hits", "flag", "event")

    def __init__(self, module: nn.Module):
        self.packed = _device_checksum(module)
        self.hits, self.flag, self.event = 0, None, None

    def stale(self, module: nn.Module) -> bool:
        """Called on every cache hit.  True when an EARLIER check has come back with a mismatch."""
        if self.packed is None or GUARD_EVERY <= 0:
            return False
        if self.event is not None and self.event.query():
            bad = bool(self.flag.item())                              # (pinned host memory, the copy has completed)
            self.event = None
            if bad:
                return True
        self.hits += 1
        if self.hits % GUARD_EVERY == 0 and self.event is None and not torch.cuda.is_current_stream_capturing():
            live = _device_checksum(module)
            if live is not None and live.device == self.packed.device:
                if self.flag is None:
                    self.flag = torch.zeros((), dtype=torch.bool).pin_memory()
                self.flag.copy_(live != self.packed, non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record()
        return False


def _live_checksum(module: nn.Module) -> float:
    """fp64 sum of all floating-point tensors below ``module`` (one host sync: debug only)."""
    acc = 0.0
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_floating_point() and t.numel():
            acc += float(t.detach().double().sum()) + 3.0 * float(t.detach().double().abs().sum())
    return acc


def _context_width(composition) -> Optional[int]:
    """Elements per context row the packed weights expect.  A hand-built composition takes its ``context_shape`` from
    its FIRST layer (bijections/base.py:203-209), which is None when an ActNorm or a permutation stands in front of the
    context-conditioned couplings: the width comes from the first layer that does define one (None if none does)."""
    from torchflows_amd.utils import event_size
    if composition.context_shape is not None:
        return event_size(composition.context_shape)
    for m in composition.modules():
        cs = getattr(m, "context_shape", None)
        if m is not composition and cs is not None:
            return event_size(cs)
    return None


def get_compiled(composition, direction: int, device: torch.device, context: bool = False) -> Optional[CompiledChain]:
    """Cached ``compile_chain``; recompiles when a parameter / buffer was modified in place."""
    cache = composition.__dict__.setdefault("_tfk_compiled", {})
    key = (direction, str(device), bool(composition.training), bool(context))
    hit = cache.get(key)
    version = _params_version(composition)
    if (hit is not None and hit[0] == version and device.type == "cuda" and len(hit) > 3 and hit[3] is not None
            and hit[3].stale(composition)):
        warnings.warn("torchflows_amd: parameters below this composition changed without their version counters moving "
                      "(an in-place edit through .data?); the packed copies were served stale for up to "
                      f"{2 * GUARD_EVERY} calls and are rebuilt now -- call invalidate_native_caches() right after such "
                      "edits", StaleProgramWarning, stacklevel=3)
        invalidate(composition)
        cache = composition.__dict__.setdefault("_tfk_compiled", {})
        hit = None
    if hit is not None and hit[0] == version:
        if _CACHE_CHECK:
            # opt-in debug check (TORCHFLOWS_AMD_DEBUG=cache_check=N): every N-th hit compares a checksum of the LIVE
            # tensors with the one taken when the program was packed -- catches edits through ``.data``
            global _cache_calls
            _cache_calls += 1
            if _cache_calls % _CACHE_CHECK == 0 and hit[2] is not None and hit[2] != _live_checksum(composition):
                raise RuntimeError(
                    "torchflows_amd: parameters below this composition changed without their version counters "
                    "moving (an in-place edit through .data?): call invalidate_native_caches() after such edits")
        return hit[1]
    chain = compile_chain(composition, direction, device, context=context)
    if chain is not None and context:
        chain.ctx_width = _context_width(composition)
    guard = _Guard(composition) if (device.type == "cuda" and chain is not None and GUARD_EVERY > 0) else None
    cache[key] = (version, chain, _live_checksum(composition) if _CACHE_CHECK else None, guard)
    if chain is None:
        warn_declined(composition, direction)
    return chain


def warn_declined(composition, direction: int) -> None:
    """Never silent: the layer-by-layer route (one libtfk kernel per reference layer + PyTorch-ROCm conditioner
    GEMMs) is ~10x slower than a flow program.  Said once per composition."""
    if not enabled() or composition.__dict__.get("_tfk_declined_warned"):
        return
    composition.__dict__["_tfk_declined_warned"] = True
    if len(tuple(composition.event_shape)) >= 3:
        return          # image blocks (ConvNet conditioners): one kernel per layer IS their route, nothing was lost
    from torchflows_amd.bijections.finite.autoregressive.layers_base import MaskedAutoregressiveBijection
    order = composition.layers if direction == FORWARD else list(composition.layers)[::-1]
    plan = _flatten(order, "forward" if direction == FORWARD else "inverse") or []
    if any(isinstance(layer, MaskedAutoregressiveBijection) and d == layer._sequential_when for layer, d in plan):
        return          # the element-by-element map of a MADE layer runs as ONE launch per layer (tfk_made_*_sequential)
    warnings.warn("torchflows_amd: this composition is not compiled to a flow program and runs layer by layer on "
                  "the HIP kernels (about 10x slower): " + _why_declined(composition, direction),
                  NativeRouteWarning, stacklevel=4)


class NativeRouteWarning(UserWarning):
    """A composition runs on the slower layer-by-layer route (see ``get_compiled``)."""


def _why_declined(composition, direction: int) -> str:
    """First layer of the chain the flow-program compiler has no op for (best effort, for the warning)."""
    from torchflows_amd.bijections.finite.autoregressive.layers_base import (
        CouplingBijection, ElementwiseBijection, MaskedAutoregressiveBijection)
    from torchflows_amd.bijections.finite.matrix.permutation import PermutationMatrix
    D = composition.n_dim
    order = composition.layers if direction == FORWARD else list(composition.layers)[::-1]
    plan = _flatten(order, "forward" if direction == FORWARD else "inverse")
    if plan is None:
        return "a layer's forward / inverse is not one of this package's implementations"
    if not (3 <= D <= 512):
        return f"event size {D} is outside the supported range"
    for layer, _ in plan:
        if isinstance(layer, PermutationMatrix):
            continue
        name = type(layer).__name__
        if isinstance(layer, (CouplingBijection, MaskedAutoregressiveBijection)):
            ct = layer.conditioner_transform
            what = (f"{name} (transformer {layer.transformer.native_kind or type(layer.transformer).__name__}, "
                    f"conditioner {type(ct).__name__}")
            if layer.context_shape is not None:
                if type(ct).__name__ != "FeedForward" or D > 256:
                    return what + f", context_shape {tuple(layer.context_shape)})"
            seq = getattr(ct, "sequential", None)
            hidden = getattr(seq[0], "out_features", None) if seq is not None and len(seq) else None
            kind = layer.transformer.native_kind
            plain = type(ct).__name__ in ("FeedForward", "MADE") and hidden is not None
            if plain and kind in ("affine", "inverse_affine", "shift") and hidden <= 128:
                continue
            if plain and kind == "rqs" and hidden <= 16 and getattr(layer.transformer, "n_bins", 8) == 8 and D <= 128:
                continue
            return what + (f", hidden width {hidden}" if hidden else "") + f", event size {D})"
        if isinstance(layer, ElementwiseBijection):
            if getattr(layer, "first_training_batch_pass", False) and layer.training:
                return f"{name} still has to initialise itself from a batch (train mode)"
            if not layer.use_global_parameters:
                if type(layer.conditioner_transform).__name__ != "Linear":
                    return f"{name} takes its parameters from the context through a {type(layer.conditioner_transform).__name__}"
                continue
            if layer.transformer.native_kind not in ("affine", "inverse_affine"):
                return f"{name} (transformer {type(layer.transformer).__name__})"
            continue
        return f"{name} has no flow-program op"
    return f"event size {D} / layer mix not covered by one kernel"


_LEAN_KINDS = frozenset((OP_AFFINE_FWD_LEAN, OP_AFFINE_INV_LEAN, OP_SHIFT_FWD_LEAN, OP_SHIFT_INV_LEAN, OP_EW_FMA,
                         OP_RQS_FWD_LEAN, OP_RQS_INV_LEAN, OP_MADE_FWD_LEAN, OP_MADE_INV_LEAN,
                         OP_LRS_FWD_LEAN, OP_LRS_INV_LEAN, OP_MADE_RQS_FWD_LEAN, OP_MADE_LRS_FWD_LEAN))


def sum_ready(chain: Optional[CompiledChain]) -> bool:
    """One LEAN launch: the fp64 sum of the log-probabilities can ride along (tfk_flow_run_mfma_sum).  Opt-in
    (TORCHFLOWS_AMD_DEBUG=sum_in_kernel=1): measured on RealNVP-64, 2^20 rows, it removes the reduction's two launches (~8 us
    of a 261 us step) and adds ~6 us to the kernel (one agent-scope ticket per workgroup, 3 072 of them): 263.7 against
    261.2 us per step over 20 steps, 239.3 against 241.3 us at the median of 100 -- a wash."""
    return (chain is not None and len(chain.segments) == 1 and chain.segments[0].mfma
            and chain.segments[0].ops[0][0] in _LEAN_KINDS
            and debug_switch("sum_in_kernel", "0") == "1")


def run_chain(chain: CompiledChain, rows: torch.Tensor, want_rows: bool, base=None, base_of_input: bool = False,
              context: Optional[torch.Tensor] = None, sum_out: Optional[torch.Tensor] = None):
    """Apply the compiled chain to ``rows`` (N, D).  Returns ``(out_rows or None, logdet or
    None, logprob or None)``; with ``base`` (loc, log_scale in logical order) the final launch
    also evaluates the diagonal-Gaussian log-density and adds the log-det (flows.py:647-648).
    ``base_of_input`` (single-launch matrix-core programs only, see ``sample_ready``): the density is that
    of the rows as they come in -- ``Flow.sample``'s ``base_log_prob(z) + log_det`` in the same launch.
    ``sum_out`` (1-element float64, ``sum_ready`` chains with ``base``): receives the fp64 sum of the log-probabilities
    from the same launch."""
    assert sum_out is None or (sum_ready(chain) and base is not None and not base_of_input and context is None)
    if context is not None and chain.ctx_width is not None and (context.dim() != 2 or context.shape[1] != chain.ctx_width):
        # the kernels only count 4-wide k-steps: a context of another width would be truncated or zero-padded in
        # silence where the reference's Linear layer raises a shape error (conditioning/context.py:46-60)
        raise ValueError(f"context rows have {tuple(context.shape[1:])} elements, the flow was built for "
                         f"context_shape with {chain.ctx_width}")
    if base_of_input:
        assert sample_ready(chain) and base is not None
        N, D = rows.shape
        logprob = torch.empty(N, dtype=torch.float32, device=rows.device)
        out = torch.empty_like(rows)
        seg = chain.segments[0]
        native.flow_run_mfma(rows, out, None, base[0], base[1], logprob, seg.packed_ops(), seg.params,
                             base_of_input=True, reverse_out=chain.reversed_out())
        if not chain.identity_out and not chain.reversed_out():
            cur, out = out, torch.empty_like(out)
            native.permute(cur, chain.pos.to(torch.int32), out)
        return out, None, logprob
    N, D = rows.shape
    dev = rows.device
    n_seg = len(chain.segments)
    padded = chain.pos_in is not None
    # lean programs read narrower rows themselves (tfk_flow_run_mfma_in): no padding pass over the rows
    # (odd event sizes: the lean affine / shift chains read them in place too -- the interpreter's plane-per-element programs
    # start with an op below OP_AFFINE_FWD_LEAN and take the padding pass)
    narrow_in = (padded and n_seg > 0 and chain.segments[0].mfma and context is None
                 and (chain.D_log % 2 == 0 or OP_AFFINE_FWD_LEAN <= chain.segments[0].ops[0][0] <= OP_SHIFT_INV_LEAN
                      or chain.segments[0].ops[0][0] in (OP_RQS_FWD_LEAN, OP_RQS_INV_LEAN, OP_LRS_FWD_LEAN, OP_LRS_INV_LEAN,
                                                         OP_MADE_FWD_LEAN, OP_MADE_INV_LEAN))
                 and (OP_AFFINE_FWD_LEAN <= chain.segments[0].ops[0][0] <= OP_RQS_INV_LEAN
                      or chain.segments[0].ops[0][0] in (OP_LRS_FWD_LEAN, OP_LRS_INV_LEAN)) and narrow_enabled())
    if padded and not narrow_in:                 # (N, D_log) -> (N, D): each half at the head of its plane
        wide = rows.new_zeros(N, chain.D)
        if chain.D_log % 2 == 0:                 # each half at the head of its plane: two strided copies
            half, hp = chain.D_log // 2, chain.D // 2
            wide[:, :half] = rows[:, :half]
            wide[:, hp:hp + half] = rows[:, half:]
        else:                                    # odd sizes: logical element l at column pos_in[l]
            wide.index_copy_(1, chain.pos_in, rows)
        rows = wide
    logprob = torch.empty(N, dtype=torch.float32, device=dev) if base is not None else None
    logdet = torch.empty(N, dtype=torch.float32, device=dev) if (base is None or n_seg > 1) else None
    cur = rows
    buf = None
    if n_seg == 0:                               # only permutations: nothing to launch
        out = rows[:, chain.pos] if want_rows else None
        ld = torch.zeros(N, dtype=torch.float32, device=dev)
        lp = None
        if base is not None:
            native.diag_gauss_logprob(rows[:, chain.pos].contiguous(), base[0], base[1], ld, logprob)
            lp = logprob
        return out, ld, lp
    loc_p = ls_p = None
    if base is not None:                          # base parameters in physical order, cached
        key = (base[0].data_ptr(), base[0]._version, base[1].data_ptr(), base[1]._version)
        hit = chain.__dict__.get("_base_cache")
        if hit is None or hit[0] != key:
            # padding elements hold 0 throughout: loc 0 and log_scale = -0.5 log(2 pi) make their density term
            # -(0 + 0.5 log(2 pi) + log_scale) vanish exactly
            loc_p = base[0].new_zeros(chain.D)
            ls_p = base[1].new_full((chain.D,), -0.9189385332046727)
            loc_p[chain.pos] = base[0]
            ls_p[chain.pos] = base[1]
            chain.__dict__["_base_cache"] = hit = (key, loc_p, ls_p)
        loc_p, ls_p = hit[1], hit[2]
    for i, seg in enumerate(chain.segments):
        last = i == n_seg - 1
        need_rows = (not last) or want_rows
        out = None
        if need_rows:
            if buf is None:
                buf = torch.empty(N, chain.D, dtype=rows.dtype, device=dev)
            out = buf                             # in place from the second segment on
        run = native.flow_run_mfma if seg.mfma else native.flow_run
        kw = dict(D=chain.D) if (narrow_in and i == 0) else {}
        if context is not None:
            kw["context"] = context
        if sum_out is not None:
            kw["sum_out"] = sum_out
        if last and need_rows and chain.reversed_out():
            kw["reverse_out"] = True              # (the reversal behind the program, folded into its store)
        run(cur, out, None if (last and base is not None and n_seg == 1) else logdet,
            loc_p if last else None, ls_p if last else None,
            logprob if last else None, seg.packed_ops(), seg.params, accumulate=(i > 0), **kw)
        if need_rows:
            cur = out
    out_rows = None
    if want_rows and padded:
        out_rows = cur.index_select(1, chain.pos)            # logical order, padding dropped
    elif want_rows:
        if chain.identity_out or chain.reversed_out():
            out_rows = cur
        else:                                     # logical l <- physical pos[l]
            out_rows = torch.empty_like(cur)
            native.permute(cur, chain.pos.to(torch.int32), out_rows)
    return out_rows, logdet, logprob
