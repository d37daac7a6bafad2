"""Layer skeletons: a conditioner predicts parameters, a transformer applies them.

Constructor signatures and attribute names follow the reference's ``layers_base.py``
(``AutoregressiveBijection`` :14-48, ``CouplingBijection`` :51-163, ``ElementwiseBijection``
:237-318) so user code and state dicts carry over.  Two execution paths:

* **HIP** (fp32, HIP device, no autograd): ``_native_step`` enqueues one libtfk kernel per
  layer on the shared row buffer of the enclosing composition -- no ``clone``, no
  boolean-mask gather/scatter, no ``repeat`` of batch-constant parameters, log-det reduced
  in-kernel and accumulated into the running sum.  The conditioner MLP stays on
  PyTorch-ROCm GEMMs; for the HalfSplit mask its input is a strided view of the rows
  (no gather copy).
* **ATen composite** (autograd / fp64 / host tensors): index-based gather and scatter
  with pre-registered index buffers instead of the reference's per-call ``nonzero``.
"""
from __future__ import annotations

from typing import Any, Optional, Sequence, Tuple, Type, Union

import torch
import torch.nn as nn

from torchflows_amd import native
from torchflows_amd.bijections.base import (Bijection, FORWARD, INVERSE, RowState, _params_ok,
                                            forward_method, inverse_method)
from torchflows_amd.bijections.finite.autoregressive.conditioning.coupling_masks import (
    PartialCoupling, make_coupling)
from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import (
    ConditionerTransform, FeedForward, Linear, MADE)
from torchflows_amd.bijections.finite.autoregressive.transformers.base import (
    ScalarTransformer, TensorTransformer)
from torchflows_amd.utils import as_rows, get_batch_shape


class AutoregressiveBijection(Bijection):
    """conditioner_transform -> h, transformer(x, h) (reference :14-48)."""

    def __init__(self, event_shape, transformer: Union[TensorTransformer, ScalarTransformer],
                 conditioner_transform: Optional[ConditionerTransform],
                 l2_regularization: bool = False, l2_coef: float = 0.01, **kwargs):
        super().__init__(event_shape=event_shape, **kwargs)
        self.conditioner_transform = conditioner_transform
        self.transformer = transformer
        self.l2_regularization = l2_regularization
        self.l2_coef = l2_coef

    def regularization(self, *aux: Any) -> torch.Tensor:
        if self.l2_regularization and self.l2_coef > 0:
            return self.sq_norm_param() * self.l2_coef
        return torch.tensor(0.0)

    def _l2_terms(self):
        """(coefficient, parameters) of this layer's L2 term when ``regularization`` is the plain
        rule above, so that a composition can evaluate all of them in one reduction; None if a
        subclass overrides ``regularization``."""
        if type(self).regularization is not AutoregressiveBijection.regularization:
            return None
        if self.l2_regularization and self.l2_coef > 0:
            # (module, name) slots of the parameters, walked once (named_modules over 27 layers cost 0.4 ms of a 3 ms
            # training step); looked up by name every step, so a replaced Parameter or a flipped requires_grad is seen.
            # Dropped by fused.invalidate (train() / eval() / load_state_dict / invalidate_native_caches).
            slots = self.__dict__.get("_tfk_l2_slots")
            if slots is None:
                slots = [(m, n) for m in self.modules() for n in m._parameters]
                self.__dict__["_tfk_l2_slots"] = slots
            params = [m._parameters[n] for m, n in slots]
            return float(self.l2_coef), [p for p in params if p is not None and p.requires_grad]
        return 0.0, []

    # run a single layer natively when it is called on its own (outside a composition)
    def _native_standalone(self, x: torch.Tensor, context, d: int):
        rows, batch = as_rows(x, self.event_shape)
        state = RowState(rows, batch)
        self._native_step(state, context, d)
        state.zero_logdet_if_unwritten()
        out = state.rows if state.owned else state.rows.clone()
        return out.view(x.shape), state.logdet.view(batch)


class CouplingBijection(AutoregressiveBijection):
    """``x = (x_A, x_B)``: ``x_A`` passes through and conditions the transform of ``x_B``
    (reference :51-163).  The inverse needs one conditioner pass too, because the
    conditioner only ever sees the untouched part."""

    def __init__(self,
                 event_shape: Sequence[int],
                 transformer_class: Type[TensorTransformer],
                 context_shape: Optional[Sequence[int]] = None,
                 coupling: PartialCoupling = None,
                 conditioner_transform_class: Type[ConditionerTransform] = FeedForward,
                 coupling_kwargs: dict = None,
                 conditioner_kwargs: dict = None,
                 transformer_kwargs: dict = None,
                 l2_regularization: bool = True,
                 **kwargs):
        coupling = coupling if coupling is not None else make_coupling(event_shape, **(coupling_kwargs or {}))
        transformer = transformer_class(event_shape=coupling.target_shape, **(transformer_kwargs or {}))
        conditioner_transform = conditioner_transform_class(
            input_event_shape=coupling.constant_shape,
            context_shape=context_shape,
            parameter_shape=transformer.parameter_shape,
            **(conditioner_kwargs or {}))
        super().__init__(event_shape=event_shape, transformer=transformer,
                         conditioner_transform=conditioner_transform, context_shape=context_shape,
                         l2_regularization=l2_regularization, **kwargs)
        self.coupling = coupling
        # gather lists move with .to(device) once; not part of the state dict (the reference
        # keeps its masks as plain attributes, coupling_masks.py:22-24)
        self.register_buffer("_source_index", coupling.source_index.long(), persistent=False)
        self.register_buffer("_target_index", coupling.target_index.long(), persistent=False)
        self.register_buffer("_target_index32", coupling.target_index.clone(), persistent=False)
        self._target_is_tail = coupling.target_is_tail
        self._source_is_head = coupling.source_is_head

    # -- ATen composite path ---------------------------------------------------
    def get_constant_part(self, x: torch.Tensor) -> torch.Tensor:
        batch = get_batch_shape(x, self.event_shape)
        part = x.reshape(*batch, -1).index_select(-1, self._source_index)
        return part.view(*batch, *self.coupling.constant_shape)

    def get_transformed_part(self, x: torch.Tensor) -> torch.Tensor:
        batch = get_batch_shape(x, self.event_shape)
        part = x.reshape(*batch, -1).index_select(-1, self._target_index)
        return part.view(*batch, *self.coupling.target_shape)

    def partition_and_predict_parameters(self, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        batch = get_batch_shape(x, self.event_shape)
        h = self.conditioner_transform(self.get_constant_part(x), context=context)
        return h.view(*batch, *self.transformer.parameter_shape)

    def _aten_apply(self, x: torch.Tensor, context, transform) -> Tuple[torch.Tensor, torch.Tensor]:
        batch = get_batch_shape(x, self.event_shape)
        h = self.partition_and_predict_parameters(x, context)
        moved, log_det = transform(self.get_transformed_part(x), h)
        flat = x.reshape(*batch, -1)
        out = flat.index_copy(-1, self._target_index, moved.reshape(*batch, -1))
        return out.view(x.shape), log_det

    # -- HIP path ------------------------------------------------------------------
    def _native_supported(self) -> bool:
        """Is there a libtfk kernel for this layer's transformer?"""
        kind = self.transformer.native_kind
        if kind == "conv1x1":
            return self.transformer.n_channels <= 16     # channels are kept in registers
        if kind == "rqs":
            return 2 <= self.transformer.n_bins <= 32
        if kind == "lrs":
            return self.transformer.n_bins in (4, 8)
        return kind in ("affine", "inverse_affine", "shift")

    def _native_ok(self, x, context) -> bool:
        return self._native_supported() and native.eligible(x, context) and _params_ok(self)

    def _native_step(self, state: RowState, context, d: int) -> None:
        rows = state.rows
        N, D = rows.shape
        S, T = self.coupling.source_event_size, self.coupling.target_event_size
        # conditioner input: a strided view for a leading-block source (HalfSplit, channel-wise
        # split), one gather otherwise; reshaped to what the conditioner expects (an image for
        # the convolutional couplings)
        x_a = rows[:, :S] if self._source_is_head else rows.index_select(1, self._source_index)
        x_a = x_a.reshape(N, *self.coupling.constant_shape)
        ctx = None if context is None else context.reshape(N, *self.context_shape)
        h = self.conditioner_transform(x_a, context=ctx).reshape(N, -1).contiguous()
        out = rows if state.owned else state.out_buffer()
        tgt = None if self._target_is_tail else self._target_index32
        acc = state.started
        kind = self.transformer.native_kind
        if kind in ("affine", "inverse_affine"):
            inv = (d == INVERSE) != (kind == "inverse_affine")
            native.affine_coupling(rows, h, out, state.logdet, tgt, T, accumulate=acc, inverse=inv)
        elif kind == "rqs":
            tr = self.transformer
            native.rqs_coupling(rows, h, out, state.logdet, tgt, T, tr.n_bins, tr.boundary,
                                accumulate=acc, inverse=(d == INVERSE))
        elif kind == "lrs":
            tr = self.transformer
            native.lrs_coupling(rows, h, out, state.logdet, tgt, T, tr.n_bins, tr.boundary,
                                accumulate=acc, inverse=(d == INVERSE))
        elif kind == "shift":
            native.shift_coupling(rows, h, out, state.logdet, tgt, T, accumulate=acc,
                                  inverse=(d == INVERSE))
        elif kind == "conv1x1":
            native.conv1x1_coupling(rows, h, out, state.logdet, tgt, T, self.transformer.n_channels,
                                    accumulate=acc, inverse=(d == INVERSE))
        else:
            raise native.NativeError(f"no kernel for transformer kind {kind!r}")
        state.started = True
        state.commit(out)

    # -- public maps ---------------------------------------------------------------
    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(x, context):
            return self._native_standalone(x, context, FORWARD)
        return self._aten_apply(x, context, self.transformer.forward)

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(z, context):
            return self._native_standalone(z, context, INVERSE)
        return self._aten_apply(z, context, self.transformer.inverse)


class MaskedAutoregressiveBijection(AutoregressiveBijection):
    """MADE conditioner + scalar transformer on every element (reference :166-225).
    ``forward`` is one parallel pass.  ``inverse`` walks the D elements in order, re-running
    the conditioner on the partially inverted row each time; as in the reference, the returned
    log-det is the one of the LAST pass (every element evaluated at its current -- already
    inverted -- value), which equals the exact log-det for affine transformers.
    On the HIP path each pass is the masked GEMMs on PyTorch-ROCm + one libtfk coupling kernel
    with all D positions as targets."""

    _sequential_when = INVERSE          # which of the two maps is the sequential one

    def __init__(self, event_shape: Sequence[int], transformer_class: Type[ScalarTransformer],
                 context_shape: Optional[Sequence[int]] = None, transformer_kwargs: dict = None,
                 conditioner_kwargs: dict = None, l2_regularization: bool = True, **kwargs):
        transformer = transformer_class(event_shape=event_shape, **(transformer_kwargs or {}))
        conditioner_transform = MADE(input_event_shape=event_shape, transformed_event_shape=event_shape,
                                     parameter_shape_per_element=transformer.parameter_shape_per_element,
                                     context_shape=context_shape, **(conditioner_kwargs or {}))
        super().__init__(transformer.event_shape, transformer, conditioner_transform,
                         context_shape=context_shape, l2_regularization=l2_regularization, **kwargs)

    # -- ATen composite path ---------------------------------------------------------
    def apply_conditioner_transformer(self, inputs, context, forward: bool = True):
        h = self.conditioner_transform(inputs, context)
        return self.transformer.forward(inputs, h) if forward else self.transformer.inverse(inputs, h)

    def _parallel(self, x, context):
        return self.apply_conditioner_transformer(x, context, True)

    def _sequential(self, z, context):
        batch = get_batch_shape(z, self.event_shape)
        flat = z.reshape(*batch, -1).clone()
        log_det = torch.zeros(batch, device=z.device)
        for i in range(flat.shape[-1]):
            tmp, log_det = self.apply_conditioner_transformer(flat.view(z.shape), context, False)
            flat = flat.clone()
            flat[..., i] = tmp.reshape(*batch, -1)[..., i]
        return flat.view(z.shape), log_det

    # -- HIP path ----------------------------------------------------------------------
    def _native_supported(self) -> bool:
        kind = self.transformer.native_kind
        if kind == "rqs":
            return 2 <= self.transformer.n_bins <= 32
        if kind == "lrs":
            return self.transformer.n_bins in (4, 8)
        return kind in ("affine", "inverse_affine")

    def _native_ok(self, x, context) -> bool:
        return self._native_supported() and native.eligible(x, context) and _params_ok(self)

    def _native_pass(self, rows, out, logdet, context, transformer_inverse: bool, accumulate: bool):
        N, D = rows.shape
        ctx = None if context is None else context.reshape(N, *self.context_shape)
        h = self.conditioner_transform(rows.view(N, *self.event_shape), ctx).reshape(N, -1).contiguous()
        kind, tr = self.transformer.native_kind, self.transformer
        if kind in ("affine", "inverse_affine"):
            native.affine_coupling(rows, h, out, logdet, None, D, accumulate=accumulate,
                                   inverse=transformer_inverse != (kind == "inverse_affine"))
        elif kind == "rqs":
            native.rqs_coupling(rows, h, out, logdet, None, D, tr.n_bins, tr.boundary,
                                accumulate=accumulate, inverse=transformer_inverse)
        else:
            native.lrs_coupling(rows, h, out, logdet, None, D, tr.n_bins, tr.boundary,
                                accumulate=accumulate, inverse=transformer_inverse)

    def _made_pack(self):
        """(W1t, b1, W2, b2) for tfk_made_affine_sequential / tfk_made_rqs_sequential -- masked,
        transposed, zero-padded to 8 / 16 / 32 / 64 hidden units -- when the layer qualifies (affine or
        8-bin RQ-spline transformer, MADE with two masked linear layers and no global parameters); cached
        until a weight changes."""
        import torch.nn as nn
        from torchflows_amd import fused
        from torchflows_amd.utils import debug_switch
        if debug_switch("made_fused", "1") == "0":      # (comparison runs)
            return None
        ct = self.conditioner_transform
        kind = self.transformer.native_kind
        if kind in ("rqs", "lrs") and self.transformer.n_bins != 8:
            return None
        if kind not in ("affine", "inverse_affine", "rqs", "lrs") or ct.n_global_parameters != 0:
            return None
        P = {"rqs": 23, "lrs": 32}.get(kind, 2)
        mods = list(ct.sequential)
        if not (len(mods) == 3 and isinstance(mods[0], MADE.MaskedLinear) and isinstance(mods[1], nn.Tanh)
                and isinstance(mods[2], MADE.MaskedLinear)):
            return None
        lo, hi = ct.output_lower_bound, ct.output_upper_bound
        H, D = mods[0].out_features, self.n_dim
        if lo != float("-inf") or hi != float("inf") or H > 64 or mods[0].in_features != D:
            return None
        version = fused._params_version(ct)
        hit = self.__dict__.get("_tfk_made_pack")
        if hit is not None and hit[0] == version:
            return hit[1]
        HP = 8 if H <= 8 else (16 if H <= 16 else (32 if H <= 32 else 64))
        if kind in ("rqs", "lrs"):
            need = (native.lib().tfk_made_rqs_sequential_lds_bytes if kind == "rqs"
                    else native.lib().tfk_made_lrs_sequential_lds_bytes)
            if HP > 16 or need(D, HP, 8) > 160 * 1024:
                return None
        elif 4 * (3 * D * HP + HP + 2 * D) + 4 * 64 * (D + 1) > 160 * 1024:
            return None                                  # weights + 64 staged rows do not fit the LDS
        with torch.no_grad():
            w1 = mods[0].weight * mods[0].mask                       # (H, D)
            w2 = mods[2].weight * mods[2].mask                       # (2 D, H)
            W1t = w1.new_zeros(D, HP)
            W1t[:, :H] = w1.t()
            b1 = w1.new_zeros(HP)
            b1[:H] = mods[0].bias
            W2 = w1.new_zeros(D, P, HP)
            W2[:, :, :H] = w2.view(D, P, H)
            b2 = mods[2].bias.detach().view(D, P).contiguous()
        packed = (W1t.contiguous(), b1, W2.contiguous(), b2)
        self.__dict__["_tfk_made_pack"] = (version, packed)
        return packed

    def _native_step(self, state: RowState, context, d: int) -> None:
        rows = state.rows
        N, D = rows.shape
        if d != self._sequential_when:                       # one parallel pass
            out = rows if state.owned else state.out_buffer()
            self._native_pass(rows, out, state.logdet, context, False, state.started)
            state.started = True
            state.commit(out)
            return
        packed = self._made_pack() if context is None else None
        if packed is not None:                               # the D passes in ONE launch (tfk_made.hip)
            out = rows if state.owned else state.out_buffer()
            if self.transformer.native_kind in ("rqs", "lrs"):
                native.made_rqs_sequential(rows, out, state.logdet, *packed, self.transformer.n_bins,
                                           self.transformer.boundary, accumulate=state.started,
                                           lrs=self.transformer.native_kind == "lrs")
            else:
                divide = self.transformer.native_kind == "affine"      # Affine.inverse divides
                native.made_affine_sequential(rows, out, state.logdet, *packed, divide, accumulate=state.started)
            state.started = True
            state.commit(out)
            return
        cur = rows.clone()
        tmp = torch.empty_like(cur)
        ld = torch.empty(N, dtype=torch.float32, device=rows.device)
        for i in range(D):                                   # layers_base.py:213-221
            self._native_pass(cur, tmp, ld, context, True, False)
            cur[:, i] = tmp[:, i]
        if state.started:
            state.logdet.add_(ld)
        else:
            state.logdet.copy_(ld)
            state.started = True
        state.commit(cur)

    # -- public maps ---------------------------------------------------------------------
    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(x, context):
            return self._native_standalone(x, context, FORWARD)
        return self._parallel(x, context)

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(z, context):
            return self._native_standalone(z, context, INVERSE)
        return self._sequential(z, context)


class InverseMaskedAutoregressiveBijection(MaskedAutoregressiveBijection):
    """The two maps exchanged (reference :227-234): ``forward`` is the sequential one (IAF)."""

    _sequential_when = FORWARD

    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(x, context):
            return self._native_standalone(x, context, FORWARD)
        return self._sequential(x, context)

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(z, context):
            return self._native_standalone(z, context, INVERSE)
        return self._parallel(z, context)


class ElementwiseBijection(AutoregressiveBijection):
    """One scalar transformer per event element.  Without a context the parameters are a
    learned tensor ``value`` of ``transformer.parameter_shape``; with a context they are
    predicted from it by a (default: linear) conditioner (reference :237-318)."""

    def __init__(self,
                 event_shape: Sequence[int],
                 transformer_class: Type[ScalarTransformer],
                 context_shape: Optional[Sequence[int]] = None,
                 transformer_kwargs: dict = None,
                 fill_value: Union[float, torch.Tensor] = None,
                 conditioner_transform_class: Type[ConditionerTransform] = Linear,
                 conditioner_kwargs: dict = None,
                 **kwargs):
        transformer = transformer_class(event_shape=event_shape, **(transformer_kwargs or {}))
        if context_shape is None:
            if fill_value is None:
                init = torch.randn(*transformer.parameter_shape)
            elif isinstance(fill_value, torch.Tensor):
                if fill_value.shape != transformer.parameter_shape:
                    raise ValueError("Shape of fill_value must match the transformer parameter shape")
                init = fill_value
            else:
                init = torch.full(size=tuple(transformer.parameter_shape), fill_value=fill_value)
            super().__init__(event_shape=event_shape, context_shape=None, transformer=transformer,
                             conditioner_transform=None, **kwargs)
            self.register_parameter("value", nn.Parameter(init))
            self.use_global_parameters = True
        else:
            conditioner_transform = conditioner_transform_class(
                input_event_shape=None, context_shape=context_shape,
                parameter_shape=transformer.parameter_shape, **(conditioner_kwargs or {}))
            super().__init__(event_shape=event_shape, context_shape=context_shape,
                             transformer=transformer, conditioner_transform=conditioner_transform,
                             **kwargs)
            self.register_buffer("value", torch.empty(size=()))
            self.use_global_parameters = False

    def prepare_h(self, context: torch.Tensor, batch_shape) -> torch.Tensor:
        if self.use_global_parameters:
            # broadcast view -- the reference repeats value to (N, D, 2) (layers_base.py:303)
            return self.value.expand(*batch_shape, *self.value.shape)
        if context is None:
            raise RuntimeError("Context must be provided")
        return self.conditioner_transform(x=None, context=context)

    # -- HIP path --------------------------------------------------------------------
    def _native_supported(self) -> bool:
        return self.transformer.native_kind in ("affine", "inverse_affine")

    def _native_ok(self, x, context) -> bool:
        return self._native_supported() and native.eligible(x, context) and _params_ok(self)

    def _native_step(self, state: RowState, context, d: int) -> None:
        rows = state.rows
        N, D = rows.shape
        kind = self.transformer.native_kind
        if kind not in ("affine", "inverse_affine"):
            raise native.NativeError(f"no elementwise kernel for transformer kind {kind!r}")
        out = rows if state.owned else state.out_buffer()
        inverse_affine = (kind == "inverse_affine")
        if self.use_global_parameters:
            native.elementwise_affine(rows, self.value.detach().reshape(D, 2).contiguous(), out,
                                      state.logdet, inverse_affine, accumulate=state.started,
                                      inverse=(d == INVERSE))
        else:
            # context-conditioned parameters: h (N, D, 2) from the conditioner; this is the
            # affine-coupling kernel with every position a target
            if context is None:
                raise RuntimeError("Context must be provided")
            h = self.conditioner_transform(x=None, context=context.reshape(N, *self.context_shape))
            h = h.reshape(N, -1).contiguous()
            inv = (d == INVERSE) != inverse_affine
            native.affine_coupling(rows, h, out, state.logdet, None, D, accumulate=state.started,
                                   inverse=inv)
        state.started = True
        state.commit(out)

    # -- public maps -------------------------------------------------------------------
    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(x, context):
            return self._native_standalone(x, context, FORWARD)
        h = self.prepare_h(context, get_batch_shape(x, self.event_shape))
        return self.transformer.forward(x, h)

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_ok(z, context):
            return self._native_standalone(z, context, INVERSE)
        h = self.prepare_h(context, get_batch_shape(z, self.event_shape))
        return self.transformer.inverse(z, h)
