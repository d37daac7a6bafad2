"""AdamW over ONE buffer: the trainable parameters of a flow re-homed as consecutive views of a flat tensor.

``Flow.fit`` (reference flows.py:268) builds ``torch.optim.AdamW(self.parameters(), lr)``.  On the device an eager
RealNVP-64 step spends a third of a millisecond of host time in that optimiser -- eleven ``_foreach`` calls over 42 small
tensors, 42 CPU step counters, 84 ``.item()`` calls -- and as much again in the 42 gradient hand-overs in front of it.
Here the parameters keep their identity (``nn.Parameter`` objects, shapes, names, ``state_dict``) but alias slices of one
buffer ``P``; the moments are flat as well, and when the backward pass hands the gradients over as the matching slices of
one buffer ``G`` (torchflows_amd/autograd.py does) the update is the SAME ``torch._foreach_*`` calls as
``torch.optim.adam._multi_tensor_adam`` runs (decoupled weight decay, lerp, addcmul, sqrt / div / add, addcdiv) on
one-element lists: bit-identical parameter trajectories, a tenth of the host time.  Gradients that arrive any other way
(a user's own graph, missing gradients) take the same calls over lists of views, with per-parameter step counts, which is
torch's algorithm itself."""
import weakref
from typing import Dict, Iterable, List, Optional

import torch

ALIGN = 4          # floats: every parameter starts on a 16-byte boundary (libtfk reads some of them as float4)

_set_versions = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
_BY_PARAM: Dict[int, "weakref.ReferenceType[FlatParams]"] = {}


class FlatParams:
    """``params`` (fp32, one device, requires_grad) as views of ``self.P``: slot k lives at
    ``P[offset[k] : offset[k] + numel[k]]``, zero padding between the slots and four zeros at the tail."""

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params: List[torch.Tensor] = list(params)
        assert self.params, "no parameters"
        dev = self.params[0].device
        assert all(p.dtype == torch.float32 and p.device == dev for p in self.params)
        self.numel = [p.numel() for p in self.params]
        self.offset, off = [], 0
        for n in self.numel:
            self.offset.append(off)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.zero_slot = off                              # index of a zero that stays zero (padding is never updated
        self.n = off + ALIGN                              # away from zero: its gradient and moments are zero)
        self.P = torch.zeros(self.n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offset, self.numel):
                self.P[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.P[o:o + n].view(p.shape)
        # split sizes that cut P (or a gradient buffer laid out like it) into [slot, pad, slot, pad, ...]
        self.split_sizes, self.piece_of_slot = [], []
        for k, (o, n) in enumerate(zip(self.offset, self.numel)):
            self.piece_of_slot.append(len(self.split_sizes))
            self.split_sizes.append(n)
            end = self.offset[k + 1] if k + 1 < len(self.offset) else self.n
            if end - o - n:
                self.split_sizes.append(end - o - n)
        self.slot_of = {id(p): k for k, p in enumerate(self.params)}
        self.last_grad: Optional[torch.Tensor] = None     # the buffer the last flat backward pass filled
        ref = weakref.ref(self)
        for p in self.params:
            _BY_PARAM[id(p)] = ref

    def intact(self) -> bool:
        """Every parameter still aliases its slot (``.to()`` / ``.cuda()`` / ``p.data = ...`` re-home them)."""
        base = self.P.data_ptr()
        return all(n == 0 or p.data_ptr() == base + 4 * o for p, o, n in zip(self.params, self.offset, self.numel))

    def grads_are_flat(self) -> Optional[torch.Tensor]:
        """The buffer ``G`` when every ``p.grad`` is slot k of it, else None."""
        G = self.last_grad
        if G is None:
            return None
        base = G.data_ptr()
        for p, o, n in zip(self.params, self.offset, self.numel):
            if n == 0:                                    # (FeedForward.global_theta_flat of a plain conditioner)
                continue
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * o or not g.is_contiguous():
                return None
        return G


def lookup(params: List[torch.Tensor]) -> Optional[FlatParams]:
    """The FlatParams that holds ALL of ``params`` (those that require a gradient), still intact; else None."""
    fb = None
    for p in params:
        if not p.requires_grad:
            continue
        ref = _BY_PARAM.get(id(p))
        got = ref() if ref is not None else None
        if got is None or (fb is not None and got is not fb) or got.params[got.slot_of.get(id(p), 0)] is not p:
            return None
        fb = got
    if fb is None or not fb.intact():
        return None
    return fb


class FlatAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` (default hyper-parameters: betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2) over a
    FlatParams buffer.  ``state_dict`` / ``load_state_dict`` are not carried over from torch's layout: ``Flow.fit``
    builds its optimiser per call and never stores it."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._trainable = [p for p in params if p.requires_grad]
        self._flat: Optional[FlatParams] = None
        self._steps: List[int] = []
        self.fast_steps = self.general_steps = 0          # updates over the one buffer / over lists of views
        self._rehome()

    def _rehome(self):
        old = self._flat
        self._flat = fb = FlatParams(self._trainable)
        M = torch.zeros_like(fb.P)
        V = torch.zeros_like(fb.P)
        if old is not None and hasattr(self, "_M"):      # the parameters were moved: the moments follow them
            with torch.no_grad():
                M.copy_(self._M.to(M.device))
                V.copy_(self._V.to(V.device))
        else:
            self._steps = [0] * len(self._trainable)
        self._M, self._V = M, V
        cut = lambda t: [t[o:o + n].view(p.shape) for p, o, n in zip(fb.params, fb.offset, fb.numel)]
        self._Mv, self._Vv = cut(M), cut(V)

    @property
    def flat(self) -> FlatParams:
        if self._flat is None:
            self._rehome()
        return self._flat

    # copy.deepcopy / pickle go through torch.optim.Optimizer.__getstate__, which keeps defaults, state and param_groups
    # only: the copy re-homes ITS parameters at first use and starts from zero moments (as documented above: the buffers
    # are not part of the optimiser's portable state)
    def __setstate__(self, state):
        super().__setstate__(state)
        self._trainable = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        self._flat = None
        self._steps = []
        self.fast_steps = self.general_steps = 0

    def zero_grad(self, set_to_none: bool = True):
        if self._flat is not None:
            self._flat.last_grad = None
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._trainable and self._trainable[0].is_cuda and torch.cuda.is_current_stream_capturing():
            # the step count and the bias corrections are host numbers: a captured step would replay ONE step's
            # corrections for ever (torch's own _cuda_graph_capture_health_check, for the same reason)
            raise RuntimeError("FlatAdamW.step() cannot be captured into a hipGraph: build the optimiser with "
                               "make_adamw(..., capturable=True)")
        fb = self._flat
        if fb is None or not fb.intact():
            self._rehome()
            fb = self._flat
        group = self.param_groups[0]
        lr, (beta1, beta2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
        G = fb.grads_are_flat()
        if G is not None and len(set(self._steps)) == 1:
            t = self._steps[0] + 1
            self._steps = [t] * len(self._steps)
            self.fast_steps += 1
            self._update([fb.P], [G], [self._M], [self._V], [t], lr, beta1, beta2, eps, wd)
            # the parameters changed through the buffer they alias: their own version counters -- what the compiled
            # inference programs are keyed on (fused._params_version) -- have to move as well
            if _set_versions is not None:
                _set_versions(fb.params, [p._version + 1 for p in fb.params])
            else:
                from torchflows_amd import fused
                fused._EPOCH[0] += 1
            return loss
        active = [k for k, p in enumerate(fb.params) if p.grad is not None]
        if not active:
            return loss
        self.general_steps += 1
        for k in active:
            self._steps[k] += 1
        self._update([fb.params[k] for k in active], [fb.params[k].grad for k in active],
                     [self._Mv[k] for k in active], [self._Vv[k] for k in active],
                     [self._steps[k] for k in active], lr, beta1, beta2, eps, wd)
        return loss

    @staticmethod
    def _update(params, grads, exp_avgs, exp_avg_sqs, steps, lr, beta1, beta2, eps, wd):
        """torch/optim/adam.py:_multi_tensor_adam, decoupled weight decay, not capturable, no amsgrad / maximize --
        call for call."""
        if wd != 0:
            torch._foreach_mul_(params, 1 - lr * wd)
        torch._foreach_lerp_(exp_avgs, grads, 1 - beta1)
        torch._foreach_mul_(exp_avg_sqs, beta2)
        torch._foreach_addcmul_(exp_avg_sqs, grads, grads, 1 - beta2)
        bias_correction1 = [1 - beta1 ** t for t in steps]
        bias_correction2 = [1 - beta2 ** t for t in steps]
        step_size = [(lr / bc) * -1 for bc in bias_correction1]
        bias_correction2_sqrt = [bc ** 0.5 for bc in bias_correction2]
        exp_avg_sq_sqrt = torch._foreach_sqrt(exp_avg_sqs)
        torch._foreach_div_(exp_avg_sq_sqrt, bias_correction2_sqrt)
        torch._foreach_add_(exp_avg_sq_sqrt, eps)
        torch._foreach_addcdiv_(params, exp_avgs, exp_avg_sq_sqrt, step_size)
