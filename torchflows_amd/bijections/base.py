"""The plugin surface: ``Bijection``, ``invert`` and ``BijectiveComposition``.

Mirrors the contract of the reference's ``torchflows/bijections/base.py`` (``Bijection``
:11-156, ``invert`` :159-167, ``BijectiveComposition`` :170-243): ``forward(x, context=)``
and ``inverse(z, context=)`` return ``(tensor, log_det)`` with ``log_det.shape ==
batch_shape``; inputs are never mutated.

What is new here is *how* a composition runs on an MI355X: when every tensor is fp32 on a
HIP device and autograd is off, the composition drives the layers' ``_native_step`` hooks
-- each one enqueues a libtfk kernel that transforms a shared ``(N, D)`` row buffer and
adds its log-det into ONE running ``(N,)`` accumulator (the ``accumulate`` flag of the
C-ABI) -- instead of materialising a log-det tensor per layer and adding them up.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from torchflows_amd import native
from torchflows_amd.utils import as_rows, event_size, get_batch_shape

FORWARD, INVERSE = 0, 1


# ``forward`` / ``inverse`` implementations are tagged with the direction they compute, so
# the composition can ask a *bound method* which way it goes.  ``Bijection.invert()`` swaps
# the bound methods on the instance (reference base.py:155-156), hence the dispatch must look
# at the method object, not at a cached flag.
def _mark(direction_id: int):
    def deco(fn):
        fn._tfk_direction = direction_id
        return fn
    return deco


forward_method = _mark(FORWARD)
inverse_method = _mark(INVERSE)


def method_direction(bound) -> Optional[int]:
    return getattr(getattr(bound, "__func__", bound), "_tfk_direction", None)


def _drop_native_caches(module, incompatible_keys=None) -> None:
    from torchflows_amd import fused
    fused.invalidate(module)


class Bijection(nn.Module):
    """Invertible map with a tractable log|det J| (reference bijections/base.py:11-156)."""

    def __init__(self, event_shape: Sequence[int], context_shape: Optional[Sequence[int]] = None,
                 **kwargs):
        super().__init__()
        self.event_shape = event_shape
        self.n_dim = event_size(event_shape)
        self.context_shape = context_shape
        self.register_load_state_dict_post_hook(_drop_native_caches)

    # -- packed-weight caches of the HIP path (new; no reference counterpart) ---
    def invalidate_native_caches(self) -> None:
        """Drop every packed copy of this module's parameters kept for the HIP kernels (flow programs, MADE packs,
        elementwise blocks, BatchNorm scale / shift).  They are refreshed automatically when a parameter is modified
        through autograd-visible in-place ops, moved or replaced, on ``train()`` / ``eval()`` and on
        ``load_state_dict``; an edit through ``.data`` is invisible to those checks and needs this call."""
        from torchflows_amd import fused
        fused.invalidate(self)

    def train(self, mode: bool = True):
        self.invalidate_native_caches()
        return super().train(mode)

    def __getstate__(self):
        # copy.deepcopy / pickle: the packed-weight caches (device tensors, ctypes arrays, weak references) stay behind;
        # the copy rebuilds its own on first use
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_tfk_")}

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() ...: the tensors keep their identity and version counters but move -- the packed
        # copies (and the dtype / device check) are dropped here instead of comparing data pointers on every call
        out = super()._apply(fn, *args, **kwargs)
        from torchflows_amd import fused
        fused.tensors_moved(self)
        return out

    # -- the two maps ------------------------------------------------------
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    # -- chunked application (reference :58-121, without the DataLoader detour) ---
    def batch_apply(self, fn: Callable, batch_size: int, x: torch.Tensor,
                    context: torch.Tensor = None, **kwargs) -> Tuple[torch.Tensor, ...]:
        n_batch = x.dim() - len(self.event_shape)
        xf = x.flatten(0, n_batch - 1)
        cf = None if context is None else context.flatten(0, n_batch - 1)
        pieces: List[Tuple[torch.Tensor, ...]] = []
        for lo in range(0, xf.shape[0], batch_size):
            args = (xf[lo:lo + batch_size],) if cf is None else (xf[lo:lo + batch_size], cf[lo:lo + batch_size])
            pieces.append(tuple(fn(*args, **kwargs)))
        return tuple(torch.cat(col, dim=0) for col in zip(*pieces))

    def batch_forward(self, x, batch_size: int, context=None, **kwargs):
        return self.batch_apply(self.forward, batch_size, x, context, **kwargs)

    def batch_inverse(self, x, batch_size: int, context=None, **kwargs):
        return self.batch_apply(self.inverse, batch_size, x, context, **kwargs)

    # -- regularisation hooks (reference :123-153) ----------------------------
    def sq_norm_param(self) -> torch.Tensor:
        terms = [p.square().sum() for p in self.parameters() if p.requires_grad]
        return sum(terms) if terms else torch.tensor(0.0)

    def regularization(self, *aux: Any) -> torch.Tensor:
        return torch.tensor(0.0)

    def invert(self):
        """Swap the two maps on this instance (reference :155-156)."""
        self.forward, self.inverse = self.inverse, self.forward
        from torchflows_amd import fused
        fused._EPOCH[0] += 1             # plans and programs that walked through this layer's old direction retire


def invert(bijection: Bijection) -> Bijection:
    """Functional form of :meth:`Bijection.invert` (reference :159-167)."""
    bijection.forward, bijection.inverse = bijection.inverse, bijection.forward
    return bijection


class RowState:
    """What a native step works on: the current rows ``(N, D)``, the running log-det
    ``(N,)`` and whether ``rows`` is a private buffer (may be overwritten in place)."""

    __slots__ = ("rows", "logdet", "owned", "spare", "batch_shape", "started")

    def __init__(self, rows: torch.Tensor, batch_shape: torch.Size):
        self.rows = rows
        self.batch_shape = batch_shape
        self.logdet = torch.empty(rows.shape[0], dtype=torch.float32, device=rows.device)
        self.owned = False      # rows still aliases the caller's tensor
        self.spare = None       # second buffer for out-of-place steps
        self.started = False    # logdet not written yet: first writer stores, later ones add

    def out_buffer(self) -> torch.Tensor:
        """A buffer a step may write its full output to (never the caller's tensor)."""
        if self.spare is None:
            self.spare = torch.empty_like(self.rows)
        return self.spare

    def commit(self, new_rows: torch.Tensor) -> None:
        if new_rows is not self.rows:
            old = self.rows
            self.spare = old if self.owned else None
            self.rows = new_rows
            self.owned = True

    def zero_logdet_if_unwritten(self) -> None:
        if not self.started:
            self.logdet.zero_()
            self.started = True


def _live_forced() -> bool:
    from torchflows_amd import autograd as hip_autograd
    return hip_autograd.live_forced()


class BijectiveComposition(Bijection):
    """Layers applied in order, log-dets summed in layer order (reference :170-243)."""

    def __init__(self, layers: List[Bijection], **kwargs):
        super().__init__(event_shape=layers[0].event_shape, context_shape=layers[0].context_shape)
        self.layers = nn.ModuleList(layers)

    def freeze_after(self, index: int):
        for i, layer in enumerate(self.layers):
            if i > index:
                layer.requires_grad_(False)

    def unfreeze_all_layers(self):
        for layer in self.layers:
            layer.requires_grad_(True)

    # -- HIP path ------------------------------------------------------------
    def _native_plan(self, order, attr: str):
        """[(layer, direction)] if every layer can run as a native step, else None."""
        plan = []
        for layer in order:
            d = method_direction(getattr(layer, attr))
            if d is None or not hasattr(layer, "_native_step"):
                return None
            supported = getattr(layer, "_native_supported", None)
            if supported is not None and not supported():
                return None
            plan.append((layer, d))
        return plan

    def _run_fused(self, x: torch.Tensor, d: int, context=None):
        """Whole chain as flow programs (conditioner in-kernel); None if not compilable."""
        from torchflows_amd import fused
        chain = fused.get_compiled(self, d, x.device, context=context is not None)
        if chain is None:
            return None
        rows, batch = as_rows(x, self.event_shape)
        crows = None if context is None else context.reshape(rows.shape[0], -1).contiguous()
        out, ld, _ = fused.run_chain(chain, rows, want_rows=True, context=crows)
        return out.view(x.shape), ld.view(batch)

    def _run_native(self, plan, x: torch.Tensor, context):
        rows, batch = as_rows(x, self.event_shape)
        state = RowState(rows, batch)
        for layer, d in plan:
            layer._native_step(state, context, d)
        state.zero_logdet_if_unwritten()
        out = state.rows if state.owned else state.rows.clone()
        return out.view(x.shape), state.logdet.view(batch)

    # -- public maps -----------------------------------------------------------
    def _run_trainable(self, x: torch.Tensor, context, d: int):
        """Autograd on the HIP path: the whole chain as one autograd node (autograd.py)."""
        from torchflows_amd import autograd as hip_autograd
        if not hip_autograd.applicable(self, x, context):
            return None
        plan = hip_autograd.training_plan(self, d)
        if plan is None:
            return None
        plan.l2 = self.__dict__.pop("_tfk_l2_request", None)
        return hip_autograd.run(self, plan, x, context)

    def _request_l2(self) -> bool:
        """Ask the next differentiable pass through this composition to evaluate ``regularization()`` inside its own
        autograd node (Flow._base_batch_loss: the penalty's gradient then joins the chain's in one buffer instead of
        meeting it in 32 small additions).  Only when the penalty is the plain L2 rule on every layer; the result --
        or None when the pass could not fold it -- is left in ``_tfk_l2_out``."""
        self.__dict__.pop("_tfk_l2_out", None)
        by_coef = {}
        for layer in self.layers:
            terms = getattr(layer, "_l2_terms", None)
            got = terms() if terms is not None else None
            if got is None:
                if type(layer).regularization is Bijection.regularization:
                    continue                     # (the base class's constant zero)
                from torchflows_amd.bijections.finite.matrix.base import InvertibleMatrix
                if isinstance(layer, InvertibleMatrix) and type(layer).regularization is InvertibleMatrix.regularization:
                    # permutations, triangular / orthogonal factors: sum_p ||p||^2 when switched on (matrix/base.py:44-49)
                    ps = [p for p in layer.parameters() if p.requires_grad] if layer.l2_regularization else []
                    if ps:
                        by_coef.setdefault(1.0, []).extend(ps)
                    continue
                return False
            coef, params = got
            if params:
                by_coef.setdefault(coef, []).extend(params)
        if not by_coef:
            return False
        self.__dict__["_tfk_l2_request"] = by_coef
        return True

    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None, **kwargs):
        if x.numel() == 0:          # no rows: nothing to launch (the reference's reshapes reject this)
            return x.clone(), x.new_zeros(get_batch_shape(x, self.event_shape))
        if not kwargs and (torch.is_grad_enabled() or _live_forced()):
            trained = self._run_trainable(x, context, FORWARD)
            if trained is not None:
                return trained
        if not kwargs and native.eligible(x, context) and _params_ok(self):
            fused_out = self._run_fused(x, FORWARD, context)
            if fused_out is not None:
                return fused_out
            plan = self._native_plan(self.layers, "forward")
            if plan is not None:
                return self._run_native(plan, x, context)
        log_det = torch.zeros(get_batch_shape(x, self.event_shape), dtype=x.dtype, device=x.device)
        for layer in self.layers:
            x, ld = layer.forward(x, context=context, **kwargs)
            log_det = log_det + ld
        return x, log_det

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None, **kwargs):
        order = list(self.layers)[::-1]
        if z.numel() == 0:
            return z.clone(), z.new_zeros(get_batch_shape(z, self.event_shape))
        if not kwargs and (torch.is_grad_enabled() or _live_forced()):
            trained = self._run_trainable(z, context, INVERSE)
            if trained is not None:
                return trained
        if not kwargs and native.eligible(z, context) and _params_ok(self):
            fused_out = self._run_fused(z, INVERSE, context)
            if fused_out is not None:
                return fused_out
            plan = self._native_plan(order, "inverse")
            if plan is not None:
                return self._run_native(plan, z, context)
        log_det = torch.zeros(get_batch_shape(z, self.event_shape), dtype=z.dtype, device=z.device)
        for layer in order:
            z, ld = layer.inverse(z, context=context)
            log_det = log_det + ld
        return z, log_det

    def regularization(self, *aux):
        """Sum of the layers' regularisation terms (reference :234-243).  The L2 terms of the
        coupling layers -- ``l2_coef * sum_p ||p||^2`` per layer, layers_base.py:38-48 -- are
        evaluated together per coefficient (one concatenation and one reduction instead of two
        tiny launches per parameter tensor, forward and backward)."""
        total = torch.tensor(0.0)
        by_coef = {}
        for layer in self.layers:
            terms = getattr(layer, "_l2_terms", None)
            got = terms() if terms is not None else None
            if got is None:
                total = total + layer.regularization()
            else:
                coef, params = got
                by_coef.setdefault(coef, []).extend(params)
        for coef, params in by_coef.items():
            if params:
                flat = torch.cat([p.reshape(-1) for p in params])
                total = total + torch.dot(flat, flat) * coef
        return total

    # a composition nested in a composition is itself a native step -- if every layer inside has one (both
    # directions: the outer plan may run either); otherwise the OUTER composition drops to the ATen loop instead
    # of raising from inside _native_step
    def _native_supported(self) -> bool:
        return (self._native_plan(self.layers, "forward") is not None
                and self._native_plan(list(self.layers)[::-1], "inverse") is not None)

    def _native_step(self, state: RowState, context, d: int) -> None:
        order = self.layers if d == FORWARD else list(self.layers)[::-1]
        plan = self._native_plan(order, "forward" if d == FORWARD else "inverse")
        if plan is None:
            raise native.NativeError("nested composition holds a layer without a native step")
        for layer, dd in plan:
            layer._native_step(state, context, dd)


def _params_ok(module: nn.Module) -> bool:
    """The kernels read parameters as fp32 on the same HIP device, without autograd."""
    grad = torch.is_grad_enabled()
    if not grad:                        # nothing can ask for a gradient: a cached static check
        from torchflows_amd import fused
        return fused.static_ok(module)
    for p in module.parameters():
        if p.device.type != "cuda" or p.dtype != torch.float32 or (grad and p.requires_grad):
            return False
    return True
