"""``Flow``: a bijection plus a base density = a distribution with ``log_prob`` and ``sample``.

The callers of the hot path, with the reference's signatures and conventions
(``torchflows/flows.py``: ``BaseFlow`` :18-67, ``Flow`` :606-713):

* ``log_prob(x) = log p_base(f(x)) + log|det df/dx|``;
* ``sample(n, return_log_prob=True)`` returns ``log p_base(z) + log|det dx/dz|`` --
  the reference's convention (:710-712), kept as is;
* inputs are moved to the module's device, context batch shapes are checked (:637-645).

On an MI355X the whole of ``log_prob`` is a chain of libtfk launches on one stream with a
single running log-det buffer; the base log-density kernel also performs the final add.
``fit`` (reference :226-455) is here too: same arguments, loss and bookkeeping, but the data set
lives on the device and the gradient runs through the reverse-mode kernels (autograd.py).
"""
from __future__ import annotations

import contextlib
import os
import time
import warnings
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from torchflows_amd.base_distributions.gaussian import DiagonalGaussian
from torchflows_amd.bijections.base import Bijection
from torchflows_amd.utils import make_adamw, event_size, flatten_event, get_batch_shape, unflatten_event

# Flow.fit, TORCHFLOWS_AMD_GRAPH=auto: full-size steps a call must run before its step is captured into a hipGraph
GRAPH_AUTO_MIN_STEPS = 32
# ... and how many replayed steps may run before their losses are read (one transfer for all of them)
GRAPH_LOSS_LAG = 16


def _optimizer_capturable(opt) -> bool:
    """True for an optimiser whose step may be captured into a hipGraph: torch's own with ``capturable=True`` in every
    parameter group (its step counters then live on the device).  FlatAdamW keeps its step count on the host."""
    from torchflows_amd.flat_optim import FlatAdamW
    if isinstance(opt, FlatAdamW):
        return False
    return all(bool(g.get("capturable", False)) for g in opt.param_groups)


def _drop_native_caches(module, incompatible_keys=None) -> None:
    module.invalidate_native_caches()


class BaseFlow(nn.Module):
    def __init__(self, event_shape,
                 base_distribution: Union[torch.distributions.Distribution, str] = "standard_normal"):
        super().__init__()
        self.event_shape = event_shape
        self.event_size = event_size(event_shape)
        if isinstance(base_distribution, str):
            if base_distribution != "standard_normal":
                raise ValueError(f"Invalid base distribution: {base_distribution}")
            self.base = DiagonalGaussian(loc=torch.zeros(self.event_size),
                                         scale=torch.ones(self.event_size))
        elif isinstance(base_distribution, torch.distributions.Distribution):
            self.base = base_distribution
        else:
            raise ValueError(f"Invalid base distribution: {base_distribution}")
        self.register_buffer("device_buffer", torch.empty(size=()))
        self._optimizer = None
        self.register_load_state_dict_post_hook(_drop_native_caches)

    def __getstate__(self):
        # copy.deepcopy / pickle: caches of the HIP path stay behind (see Bijection.__getstate__)
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_tfk_")}

    def invalidate_native_caches(self) -> None:
        """Drop every packed copy of the parameters kept for the HIP kernels (see ``Bijection.invalidate_native_caches``):
        needed only after edits the version counters do not see (``p.data.mul_(...)``, manual weight averaging)."""
        from torchflows_amd import fused
        fused.invalidate(self)

    def train(self, mode: bool = True):
        self.invalidate_native_caches()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)       # (see Bijection._apply)
        from torchflows_amd import fused
        fused.tensors_moved(self)
        return out

    def get_device(self) -> torch.device:
        return self.device_buffer.device

    def base_log_prob(self, z: torch.Tensor) -> torch.Tensor:
        return self.base.log_prob(flatten_event(z, self.event_shape))

    def base_sample(self, sample_shape) -> torch.Tensor:
        return unflatten_event(self.base.sample(sample_shape), self.event_shape)

    def regularization(self, *args, **kwargs) -> torch.Tensor:
        return torch.tensor(0.0)

    # -- forward-KL fit against a known target density (reference flows.py:79-197) ------------------------------------
    def _loss_kl_p_to_q(self, data: torch.Tensor, log_prob_target_data: torch.Tensor,
                        use_regularization: bool = True) -> torch.Tensor:
        """``mean(log p(x) - log q(x)) [+ regularization]`` on one batch (reference :79-94)."""
        dev = self.get_device()
        loss = torch.mean(log_prob_target_data.to(dev) - self.log_prob(data.to(dev)))
        if use_regularization:
            loss = loss + self.regularization()
        return loss

    def fit_kl_p_to_q(self, x_train: torch.Tensor, x_val: torch.Tensor, neg_log_prob_target, n_epochs: int = 500,
                      lr: float = 0.05, batch_size: int = 1024, show_progress: bool = False,
                      keep_best_weights: bool = True, early_stopping: bool = False, early_stopping_threshold: int = 50,
                      time_limit_seconds: float = None, reset_optimizer: bool = True):
        """Fit by minimising KL(p || q) on samples of p whose (negative) log-density is known; arguments, loss and
        bookkeeping as the reference's (:96-197): batches in data order (its DataLoader does not shuffle), the
        validation loss is the SUM of the batch means without the regularisation term, best weights by that sum,
        early stopping on it, ``eval()`` on return.  As in ``fit`` the two sets and their target log-densities live on
        the flow's device for the whole call; ``log_prob`` with gradients runs on the reverse-mode kernels there."""
        if len(list(self.parameters())) == 0:
            return
        dev = self.get_device()
        with torch.no_grad():
            train = (x_train.to(dev), (-neg_log_prob_target(x_train)).detach().to(dev))
            val = (x_val.to(dev), (-neg_log_prob_target(x_val)).detach().to(dev))
        self.train()
        t0 = time.time()
        if self._optimizer is None or reset_optimizer:
            self._optimizer = make_adamw(self.parameters(), lr)

        def snapshot():
            return {k: v.detach().clone() for k, v in self.state_dict().items()}

        def in_batches(pair):
            for lo in range(0, len(pair[0]), batch_size):
                yield pair[0][lo:lo + batch_size], pair[1][lo:lo + batch_size]

        val_loss, best_val, best_epoch, best_weights = None, float("inf"), 0, snapshot()
        epochs, pbar = range(n_epochs), None
        if show_progress:
            from tqdm import tqdm
            epochs = pbar = tqdm(epochs, desc="Fitting NF")
        for epoch in epochs:
            if time_limit_seconds is not None and time.time() - t0 >= time_limit_seconds:
                print("Training time limit exceeded")
                break
            for xb, tb in in_batches(train):
                self._optimizer.zero_grad()
                train_loss = self._loss_kl_p_to_q(xb, tb)
                train_loss.backward()
                self._optimizer.step()
                if pbar is not None:
                    tail = "" if val_loss is None else (f", Validation loss: {val_loss:.4f} "
                                                        f"[best: {best_val:.4f} @ {best_epoch}]")
                    pbar.set_postfix_str(f"Training loss (batch): {float(train_loss):.4f}" + tail)
            with torch.no_grad():
                total = sum(self._loss_kl_p_to_q(xb, tb, use_regularization=False) for xb, tb in in_batches(val))
            val_loss = float(total)                    # (one transfer per epoch)
            if val_loss < best_val:
                best_val, best_epoch = val_loss, epoch
                if keep_best_weights:
                    best_weights = snapshot()
            if early_stopping and epoch - best_epoch > early_stopping_threshold:
                break
        if keep_best_weights:
            self.load_state_dict(best_weights)
        self.eval()


class Flow(BaseFlow):
    def __init__(self, bijection: Bijection, **kwargs):
        super().__init__(event_shape=bijection.event_shape, **kwargs)
        self.register_module("bijection", bijection)

    @property
    def context_shape(self):
        return self.bijection.context_shape

    def _checked_context(self, x: torch.Tensor, context):
        if context is None:
            return None
        if self.context_shape is None:
            raise ValueError("Context shape must be set.")
        if self.event_shape is None:
            raise ValueError("Event shape must be set.")
        if get_batch_shape(x, self.event_shape) != get_batch_shape(context, self.context_shape):
            raise AssertionError("x and context must share their batch shape")
        return context.to(self.get_device())

    def _fused_log_prob(self, x: torch.Tensor, want_z: bool, context: torch.Tensor = None, want_sum: bool = False):
        """log_prob (and z) as flow programs ending in the base log-density: 4*D + 4 bytes of
        HBM traffic per evaluation.  None when the chain is not compilable.  ``want_sum``: a third value, the fp64
        sum of the log-probabilities from the same launch (1-element tensor) or None if the chain cannot carry it."""
        from torchflows_amd import fused, native
        from torchflows_amd.bijections.base import (BijectiveComposition, _params_ok,
                                                    method_direction)
        from torchflows_amd.utils import as_rows
        b = self.bijection
        if not (isinstance(b, BijectiveComposition) and isinstance(self.base, DiagonalGaussian)
                and native.eligible(x, self.base.loc, self.base.log_scale) and _params_ok(self)):
            return None
        if context is not None and not native.eligible(context):
            return None
        from torchflows_amd import autograd as hip_autograd
        if hip_autograd.live_forced():
            return None             # (a captured validation pass: launches that read the live parameters, autograd.live_route)
        d = method_direction(b.forward)
        chain = None if d is None else fused.get_compiled(b, d, x.device, context=context is not None)
        if chain is None:
            return None
        rows, batch = as_rows(x, self.event_shape)
        crows = None if context is None else context.reshape(rows.shape[0], -1).contiguous()
        total = None
        if want_sum and crows is None and fused.sum_ready(chain) and rows.shape[0] > 0:
            total = torch.empty(1, dtype=torch.float64, device=rows.device)
        z, _, lp = fused.run_chain(chain, rows, want_rows=want_z,
                                   base=(self.base.loc.detach(), self.base.log_scale.detach()), context=crows,
                                   sum_out=total)
        if want_sum:
            return (z.view(x.shape) if want_z else None), lp.view(batch), total
        return (z.view(x.shape) if want_z else None), lp.view(batch)

    def _fused_sample(self, z: torch.Tensor):
        """``(x, base_log_prob(z) + log_det)`` of ``sample(return_log_prob=True)`` (flows.py:699-707) as ONE
        flow-program launch when the inverse chain is a single matrix-core program; else None."""
        from torchflows_amd import fused, native
        from torchflows_amd.bijections.base import (BijectiveComposition, _params_ok,
                                                    method_direction)
        from torchflows_amd.utils import as_rows
        b = self.bijection
        if not (isinstance(b, BijectiveComposition) and isinstance(self.base, DiagonalGaussian)
                and native.eligible(z, self.base.loc, self.base.log_scale) and _params_ok(self)):
            return None
        d = method_direction(b.inverse)
        chain = None if d is None else fused.get_compiled(b, d, z.device)
        if not fused.sample_ready(chain):
            return None
        rows, batch = as_rows(z, self.event_shape)
        with torch.no_grad():
            x, _, lp = fused.run_chain(chain, rows, want_rows=True, base_of_input=True,
                                       base=(self.base.loc.detach(), self.base.log_scale.detach()))
        return x.view(z.shape), lp.view(batch)

    def log_prob_and_sum(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """``(log_prob(x), its sum in fp64 as a 1-element tensor)``: on the HIP path the sum comes out of the same
        launch as the log-probabilities (tfk_flow_run_mfma_sum) -- the per-rank term of the sharded log-likelihood
        (torchflows_amd.distributed); otherwise ``log_prob`` (flows.py:650-658) followed by a reduction."""
        from torchflows_amd.distributed import local_sum_f64
        if not torch.is_grad_enabled() or not x.requires_grad:
            ctx = self._checked_context(x, context)
            got = self._fused_log_prob(x.to(self.get_device()), want_z=False, context=ctx, want_sum=True)
            if got is not None and got[2] is not None:
                return got[1], got[2]
            if got is not None:
                return got[1], local_sum_f64(got[1])
        lp = self.log_prob(x, context=context)
        return lp, local_sum_f64(lp)

    def forward_with_log_prob(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        context = self._checked_context(x, context)
        fused_out = self._fused_log_prob(x.to(self.get_device()), want_z=True, context=context)
        if fused_out is not None:
            return fused_out
        z, log_det = self.bijection.forward(x.to(self.get_device()), context=context)[:2]
        zf = flatten_event(z, self.event_shape)
        if isinstance(self.base, DiagonalGaussian):
            return z, self.base.log_prob_plus(zf, log_det)      # fused base density + add
        return z, self.base.log_prob(zf) + log_det

    def regularization(self, *args, **kwargs):
        return self.bijection.regularization(*args, **kwargs)

    # -- maximum-likelihood training (reference flows.py:199-224, :226-455) ----------------
    def _base_batch_loss(self, batch, reduction=torch.mean, use_regularization: bool = True) -> torch.Tensor:
        """``-reduction(log_prob(x) * w) / event_size [+ regularization]`` (reference :199-224)."""
        x, weights = batch[:2]
        context = batch[2] if len(batch) == 3 else None
        dev = self.get_device()
        # on the HIP path the L2 penalty of the coupling layers is evaluated inside the chain's autograd node when the
        # parameters live in one buffer (autograd.py, flat_optim.py); otherwise as the reference does, below
        b = self.bijection
        folded = (use_regularization and dev.type == "cuda" and torch.is_grad_enabled()
                  and hasattr(b, "_request_l2") and b._request_l2())
        lp = self.log_prob(x.to(dev), context=context)
        reg = None
        if folded:
            b.__dict__.pop("_tfk_l2_request", None)
            reg = b.__dict__.pop("_tfk_l2_out", None)
        loss = -reduction(lp * weights.to(dev)) / self.event_size
        if use_regularization:
            loss = loss + (reg if reg is not None else self.regularization())
        return loss

    def _graph_safe(self) -> bool:
        from torchflows_amd import autograd as hip_autograd
        from torchflows_amd.bijections.base import BijectiveComposition, method_direction
        b = self.bijection
        if not (hip_autograd.enabled() and isinstance(self.base, DiagonalGaussian)):
            return False
        if self.base.loc.requires_grad or self.base.log_scale.requires_grad:
            return False
        if any(p.dtype != torch.float32 or p.device.type != "cuda" for p in self.parameters()):
            return False
        from torchflows_amd.bijections.finite.multiscale.base import MultiscaleBijection
        if isinstance(b, MultiscaleBijection):
            return self._image_graph_safe(b)
        if not isinstance(b, BijectiveComposition) or len(self.base.event_shape) != 1:
            return False
        d = method_direction(b.forward)
        plan = None if d is None else hip_autograd.training_plan(b, d)
        return plan is not None and hip_autograd.fully_fused(plan, b.n_dim)

    def _image_graph_safe(self, b) -> bool:
        """An image flow whose training step is libtfk launches and elementwise / index ATen ops only: every coupling on
        the reverse-mode chain (autograd.training_plan) and every ConvNet conditioner on csrc/tfk_convtrain.hip -- no
        MIOpen, no GEMM-library call, no host synchronisation."""
        from torchflows_amd import autograd as hip_autograd, convnet_train
        from torchflows_amd.bijections.base import BijectiveComposition, method_direction
        from torchflows_amd.bijections.finite.multiscale.base import MultiscaleBijection, Squeeze
        from torchflows_amd.bijections.finite.multiscale.conditioning.classic import ConvNet
        dev = self.get_device()
        for m in b.modules():
            if isinstance(m, ConvNet) and not convnet_train.static_usable(m, dev):
                return False
        todo = [b]
        while todo:
            m = todo.pop()
            layers = list(m.checkerboard_layers) + (list(m.channel_wise_layers) if m.n_blocks > 1 else [])
            for layer in layers:
                d = method_direction(layer.forward)
                if not isinstance(layer, BijectiveComposition) or d is None \
                        or hip_autograd.training_plan(layer, d) is None:
                    return False
            if m.n_blocks > 1:
                if not isinstance(m.small_bijection, MultiscaleBijection) or not isinstance(m.squeeze, Squeeze):
                    return False
                todo.append(m.small_bijection)
        return True

    def fit(self,
            x_train: torch.Tensor,
            n_epochs: int = 500,
            lr: float = 0.05,
            batch_size: Union[int, str, None] = 1024,
            shuffle: bool = True,
            show_progress: bool = False,
            w_train: torch.Tensor = None,
            context_train: torch.Tensor = None,
            x_val: torch.Tensor = None,
            w_val: torch.Tensor = None,
            context_val: torch.Tensor = None,
            keep_best_weights: bool = True,
            early_stopping: bool = False,
            early_stopping_threshold: int = 50,
            max_batch_size_mb: int = None,
            time_limit_seconds: Union[float, int] = None,
            reset_optimizer: bool = True):
        """Maximum-likelihood fit with AdamW; arguments and bookkeeping as the reference's
        ``Flow.fit`` (:226-455): per-batch loss ``-mean(log_prob * w) / event_size +
        regularization``, best weights by validation (else training) loss per epoch, early
        stopping, roll-back on a non-finite loss, ``batch_size="adaptive"`` doubling every
        10 epochs.  What differs is where the data lives: the training and validation sets
        are moved to the flow's device ONCE and batches are index views of them (a device
        ``randperm`` per epoch) -- no host DataLoader, no per-batch host-to-device copies --
        and on an MI355X the gradient runs through the reverse-mode kernels (autograd.py)."""
        t0 = time.time()
        self.train()
        if not any(p.requires_grad for p in self.parameters()):
            self.eval()                      # nothing to fit (also: no parameters at all)
            return
        dev = self.get_device()

        def resident(x, w, c, label):
            n = len(x)
            if w is None:
                w = torch.ones(size=tuple(get_batch_shape(x, self.event_shape)))
            if len(w) != n:
                raise ValueError(f"Expected same number of {label} data and {label} weights, "
                                 f"but found {n} and {len(w)}")
            if c is not None and len(c) != n:
                raise ValueError(f"Expected same number of {label} data and {label} contexts, "
                                 f"but found {n} and {len(c)}")
            return x.to(dev), w.to(dev), (None if c is None else c.to(dev))

        def batches(data, size, permute):
            x, w, c = data
            n = len(x)
            order = torch.randperm(n, device=dev) if (permute and n > 1) else None
            for lo in range(0, n, size):
                idx = slice(lo, lo + size) if order is None else order[lo:lo + size]
                yield (x[idx], w[idx]) if c is None else (x[idx], w[idx], c[idx])

        train = resident(x_train, w_train, context_train, "training")
        val = None if x_val is None else resident(x_val, w_val, context_val, "validation")
        n_train = len(x_train)

        adaptive, max_batch_size = False, None
        if isinstance(batch_size, int) and batch_size > n_train > 0:
            batch_size = n_train         # (the same batches; lets a data set smaller than one batch -- the reference's
                                         # notebooks: 1 000 rows, batch size 1 024 -- count as full-size steps below)
        if batch_size is None:
            batch_size = n_train
        elif isinstance(batch_size, str):
            if batch_size != "adaptive":
                raise ValueError(f"Unknown batch size rule: {batch_size}")
            adaptive = True
            max_batch_size = min(4096, n_train // 10)
            if max_batch_size_mb is not None:
                max_batch_size = max(1, min(max_batch_size, int(max_batch_size_mb / (self.event_size / 2 ** 20))))
            batch_size = max(32, min(1024, n_train // 100))

        # On the device a small-batch step is bound by the host (~1.4 ms of Python / launch work per step whatever the
        # batch size): after two eager steps (lazy initialisation, ActNorm statistics) a full-size batch step is captured
        # ONCE into a hipGraph -- forward, backward and the AdamW update -- and replayed on static buffers.
        # Only the route that is known to be capture-safe qualifies (``_graph_safe``): a composition whose couplings all
        # run as the fused launches (libtfk kernels + elementwise ATen ops on fixed shapes, no GEMM-library calls, no
        # host synchronisation) and a fixed diagonal Gaussian base.  Steps with hipBLASLt GEMMs invalidated the capture
        # on this stack, and an invalidated capture does not raise here, it crashes the process -- so no speculative
        # tries: TORCHFLOWS_AMD_GRAPH=1 captures whenever that route applies, "auto" (the default) additionally asks for
        # a fixed batch size and enough full-size steps to pay for the capture, 0 never captures.  Every replay first
        # checks that the parameters still live where the capture saw them (a replay over freed or moved tensors is the
        # other way to lose the process) and drops to eager steps otherwise.  The whole loop then runs on a side stream:
        # autograd state created on the legacy default stream would invalidate the capture.
        graph_mode = os.environ.get("TORCHFLOWS_AMD_GRAPH", "auto")
        use_graph = (dev.type == "cuda" and graph_mode != "0" and context_train is None and self._graph_safe())
        if use_graph and graph_mode != "1":
            use_graph = (not adaptive) and n_epochs * (n_train // max(int(batch_size), 1)) >= GRAPH_AUTO_MIN_STEPS
        if self._optimizer is None or reset_optimizer:
            from torchflows_amd.bijections.finite.multiscale.base import MultiscaleBijection
            self._optimizer = make_adamw(self.parameters(), lr, capturable=use_graph,
                                         fused=use_graph and isinstance(self.bijection, MultiscaleBijection))
        elif use_graph and not _optimizer_capturable(self._optimizer):
            # fit(reset_optimizer=False) behind variational_fit or a short eager fit keeps THAT optimiser: a FlatAdamW (or a
            # non-capturable AdamW) counts its steps on the host, and a captured step would replay one step's bias
            # corrections for ever -- a silently wrong trajectory.  Such a fit runs eager steps.
            use_graph = False
        graphed = None               # (batch size, graph, static x, static w, static loss, tensor addresses)
        from torchflows_amd.utils import debug_switch as _debug_switch
        val_graph = None if _debug_switch("val_graph", "1") != "0" else False     # None: not tried yet; False: not used
        stats = {"eager_steps": 0, "graph_replays": 0, "graph_captures": 0}
        self._fit_stats = stats

        side = main_stream = None
        if use_graph:
            main_stream = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(main_stream)

        def where():                     # what a replay reads and writes besides its static inputs
            return tuple(t.data_ptr() for t in list(self.parameters()) + list(self.buffers()))

        def capture(xb, wb):
            xs, ws = xb.clone(), wb.clone()
            graph = torch.cuda.CUDAGraph()
            self._optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                static_loss = self._base_batch_loss((xs, ws), reduction=torch.mean, use_regularization=True)
                static_loss.backward()
                self._optimizer.step()
            stats["graph_captures"] += 1
            # (only the VALUE is read back: a tensor that kept the captured step's autograd graph -- and with it the
            # AccumulateGrad nodes made on the capture stream -- alive made later eager steps warn about a stream mismatch)
            return len(xb), graph, xs, ws, static_loss.detach(), where()

        def snapshot(into=None):
            """The state dict's values, copied -- into the previous snapshot's tensors when there is one (only the latest
            kept weights are ever read back): a few multi-tensor copies instead of one launch per entry (an image flow
            has ~350 of them; the copy per epoch was 10 % of its training step)."""
            state = self.state_dict()
            if into is None or into.keys() != state.keys():
                return {k: v.detach().clone() for k, v in state.items()}
            groups = {}
            for k, v in state.items():
                dst = into[k]
                if dst.shape != v.shape or dst.dtype != v.dtype or dst.device != v.device:
                    return {k: v.detach().clone() for k, v in state.items()}
                groups.setdefault((v.dtype, v.device), ([], []))
                groups[(v.dtype, v.device)][0].append(dst)
                groups[(v.dtype, v.device)][1].append(v.detach())
            with torch.no_grad():
                for dsts, srcs in groups.values():
                    torch._foreach_copy_(dsts, srcs)
            return into

        best_weights = snapshot()
        best_val, best_train = float("inf"), float("inf")
        best_val_epoch = best_train_epoch = 0
        val_loss: Optional[float] = None
        diverged = False
        epochs = range(n_epochs)
        pbar = None
        if show_progress:
            from tqdm import tqdm
            epochs = pbar = tqdm(epochs, desc="Fitting NF")
        with (torch.cuda.stream(side) if use_graph else contextlib.nullcontext()):
            for epoch in epochs:
                if time_limit_seconds is not None and time.time() - t0 >= time_limit_seconds:
                    print("Training time limit exceeded")
                    break
                if adaptive and epoch % 10 == 9 and batch_size < max_batch_size:
                    batch_size = min(2 * batch_size, max_batch_size)
                total, count = 0.0, 0
                # replayed steps leave their loss on the device; the values are read GRAPH_LOSS_LAG steps late, in one
                # transfer (a read per step would idle the GPU while the host prepares the next replay).  A non-finite loss
                # is therefore noticed a few updates late -- the roll-back restores the kept weights either way.
                pending = []

                def drain():
                    nonlocal total, count
                    if not pending:
                        return True
                    values = torch.stack(pending).tolist()
                    pending.clear()
                    for v in values:
                        if v != v or v in (float("inf"), float("-inf")):
                            return False
                        total += v
                        count += 1
                    return True

                for batch in batches(train, batch_size, shuffle):
                    replay = use_graph and stats["eager_steps"] >= 2 and len(batch[0]) == batch_size
                    if replay and (graphed is None or graphed[0] != batch_size):
                        loss = None                   # no autograd graph of an eager step may be alive
                        try:
                            graphed = capture(batch[0], batch[1])
                        except Exception as exc:      # capture is an optimisation: fall back, say so once
                            warnings.warn(f"hipGraph capture of the training step failed ({exc}); running eagerly")
                            use_graph, replay, graphed = False, False, None
                            torch.cuda.synchronize()
                            self._optimizer.zero_grad(set_to_none=True)
                    if replay and graphed[5] != where():
                        warnings.warn("a parameter or buffer moved since the training step was captured; running eagerly")
                        use_graph, replay, graphed = False, False, None
                        stats["graph_dropped"] = stats.get("graph_dropped", 0) + 1
                    if replay:
                        # (a non-finite loss is noticed after the captured update has run; the
                        # roll-back below restores the kept weights either way)
                        graphed[2].copy_(batch[0])
                        graphed[3].copy_(batch[1])
                        graphed[1].replay()
                        stats["graph_replays"] += 1
                        if pbar is None:
                            pending.append(graphed[4].detach().clone())
                            if len(pending) < GRAPH_LOSS_LAG or drain():
                                continue
                            value = float("nan")       # (a non-finite loss among the drained steps)
                        else:
                            value = float(graphed[4].detach())
                    else:
                        ok = drain()
                        self._optimizer.zero_grad()
                        loss = self._base_batch_loss(batch, reduction=torch.mean, use_regularization=True)
                        value = float(loss.detach()) if ok else float("nan")
                    if value != value or value in (float("inf"), float("-inf")):
                        self.load_state_dict(best_weights)     # the last kept (else the initial) weights
                        diverged = True
                        warnings.warn("Flow training diverged. Reverting to previous weights.")
                        break
                    total += value
                    count += 1
                    if not replay:
                        loss.backward()
                        self._optimizer.step()
                        stats["eager_steps"] += 1
                    if pbar is not None:
                        text = f"Training loss (batch): {value:.4f} [{best_train:.4f} @ {best_train_epoch}]"
                        if val_loss is not None:
                            text += f" , Validation loss (batch): {val_loss:.4f} [{best_val:.4f} @ {best_val_epoch}]"
                        pbar.set_postfix_str(text)
                if graphed is not None:       # replays move the weights, not their version counters
                    from torchflows_amd import fused
                    fused.invalidate(self, compiled_only=True)
                if not diverged and not drain():
                    self.load_state_dict(best_weights)
                    diverged = True
                    warnings.warn("Flow training diverged. Reverting to previous weights.")
                if diverged:
                    break
                average = total / count
                if average < best_train:
                    best_train, best_train_epoch = average, epoch
                if val is not None:
                    acc = None
                    if val_graph is not False and graphed is not None and len(x_val) <= batch_size \
                            and val[2] is None and graphed[5] == where() and stats.get("val_eager_passes", 0) >= 1:
                        # the validation pass (one batch, resident tensors) on the training route's launches, which read
                        # the LIVE parameters (autograd.live_route): captured once as well and replayed per epoch
                        # (2.2 -> 0.7 ms of a 7.9 ms epoch of the notebook's multiscale fit).  The packed flow programs
                        # that a no-grad evaluation normally takes are copies of the weights: a replay would never
                        # refresh them.
                        if val_graph is None:
                            try:
                                from torchflows_amd import autograd as hip_autograd, convnet_train
                                # (lazy state of this route, set up outside the capture; BatchNorm's running statistics
                                # are left alone: this epoch's validation batch is counted by the replay below)
                                with torch.no_grad(), hip_autograd.live_route(), convnet_train.recomputing():
                                    self._base_batch_loss((val[0], val[1]), reduction=torch.sum, use_regularization=False)
                                vg = torch.cuda.CUDAGraph()
                                with torch.no_grad(), hip_autograd.live_route(), torch.cuda.graph(vg):
                                    static_val = self._base_batch_loss((val[0], val[1]), reduction=torch.sum,
                                                                       use_regularization=False)
                                val_graph = (vg, static_val)
                                stats["val_graph_captures"] = 1
                            except Exception as exc:
                                warnings.warn(f"hipGraph capture of the validation pass failed ({exc}); running it eagerly")
                                val_graph = False
                                torch.cuda.synchronize()
                        if val_graph:
                            val_graph[0].replay()
                            acc = float(val_graph[1])
                            stats["val_graph_replays"] = stats.get("val_graph_replays", 0) + 1
                    if acc is None:
                        acc = 0.0
                        stats["val_eager_passes"] = stats.get("val_eager_passes", 0) + 1      # (lazy state is set up)
                        with torch.no_grad():
                            for batch in batches(val, batch_size, False):
                                acc += float(self._base_batch_loss(batch, reduction=torch.sum, use_regularization=False))
                    val_loss = acc / len(x_val)
                    stats["val_loss"] = val_loss
                    if val_loss < best_val:
                        best_val, best_val_epoch = val_loss, epoch
                mark = best_val_epoch if val is not None else best_train_epoch
                if keep_best_weights and mark == epoch:
                    best_weights = snapshot(best_weights)
                if early_stopping and epoch - mark > early_stopping_threshold:
                    break
        if side is not None:
            main_stream.wait_stream(side)
        graphed = val_graph = None
        if keep_best_weights:
            self.load_state_dict(best_weights)
        self.eval()

    # -- stochastic variational inference (reference flows.py:457-603) ------------------------
    def _variational_loss(self, target_log_prob, n_samples: int, use_regularization: bool = True,
                          check_for_divergences: bool = False):
        """``-mean(target_log_prob(x) + flow_log_prob)`` on ``n_samples`` draws of the flow, with the
        reference's ``sample(return_log_prob=True)`` convention (reference :457-497).  Returns
        (loss, flow log-prob, target log-prob, diverged)."""
        flow_x, flow_lp = self.sample(n_samples, return_log_prob=True)
        target_lp = target_log_prob(flow_x)
        loss = -torch.mean(target_lp + flow_lp)
        if use_regularization:
            loss = loss + self.regularization()
        diverged = False
        if check_for_divergences:
            with torch.no_grad():
                diverged = bool((~torch.isfinite(loss)) | (flow_x.abs().max() > 1e8) | (flow_lp.abs().max() > 1e6)
                                | (~torch.isfinite(flow_x)).any() | (~torch.isfinite(flow_lp)).any())
        return loss, flow_lp, target_lp, diverged

    def variational_fit(self, target_log_prob, n_epochs: int = 500, lr: float = 0.05, n_samples: int = 1,
                        early_stopping: bool = False, early_stopping_threshold: int = 50,
                        keep_best_weights: bool = True, show_progress: bool = False,
                        check_for_divergences: bool = False, time_limit_seconds: Union[float, int] = None,
                        reset_optimizer: bool = True):
        """Fit to an unnormalised target log-density by stochastic variational inference (reference
        :499-603; Rezende & Mohamed 2015).  One AdamW step per epoch on ``n_samples`` fresh draws; the
        draws go through ``bijection.inverse`` with gradients -- on an MI355X the inverse-direction
        reverse-mode kernels (autograd.py)."""
        t0 = time.time()
        if len(list(self.parameters())) == 0:
            return
        self.train()
        if self._optimizer is None or reset_optimizer:
            self._optimizer = make_adamw(self.parameters(), lr)

        def snapshot():
            return {k: v.detach().clone() for k, v in self.state_dict().items()}

        initial, best = snapshot(), snapshot()
        best_loss, best_epoch, n_div, gave_up = float("inf"), 0, 0, False
        stats = {"eager_steps": 0}
        self._fit_stats = stats
        epochs = range(n_epochs)
        pbar = None
        if show_progress:
            from tqdm import tqdm
            epochs = pbar = tqdm(epochs, desc="Fitting with SVI")
        for epoch in epochs:
            if time_limit_seconds is not None and time.time() - t0 >= time_limit_seconds:
                print("Training time limit exceeded")
                break
            if check_for_divergences and not all(bool(torch.isfinite(p).all()) for p in self.parameters()):
                gave_up = True
                print("Flow training diverged")
                print("Reverting to initial weights")
                break
            self._optimizer.zero_grad()
            value = mean_flow = mean_target = float("nan")
            try:
                loss, flow_lp, target_lp, diverged = self._variational_loss(
                    target_log_prob, n_samples, use_regularization=True, check_for_divergences=True)
                if not diverged:
                    loss.backward()
                    self._optimizer.step()
                    stats["eager_steps"] += 1
                    value = float(loss.detach())
                    if value < best_loss:
                        best_loss, best_epoch = value, epoch
                        if keep_best_weights:
                            best = snapshot()
                    mean_flow, mean_target = float(flow_lp.detach().mean()), float(target_lp.detach().mean())
            except ValueError:
                diverged = True
            n_div += int(diverged)
            if pbar is not None:
                pbar.set_postfix_str(f"Loss: {value:.4f} [best: {best_loss:.4f} @ {best_epoch}], divergences: {n_div}, "
                                     f"flow log_prob: {mean_flow:.2f}, target log_prob: {mean_target:.2f}")
            if early_stopping and epoch - best_epoch > early_stopping_threshold:
                break
        if gave_up:
            self.load_state_dict(initial)
        elif keep_best_weights:
            self.load_state_dict(best)
        self.eval()

    def log_prob(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        ctx = None if context is None else self._checked_context(x, context)
        fused_out = self._fused_log_prob(x.to(self.get_device()), want_z=False, context=ctx)
        if fused_out is not None:
            return fused_out[1]
        got = self._image_log_prob(x, ctx)
        if got is not None:
            return got
        return self.forward_with_log_prob(x, context)[1]

    def _image_log_prob(self, x: torch.Tensor, context):
        """``log_prob`` of an image flow whose bijection compiles to an image program (image_program.py): the couplings, then
        the deferred ActNorm maps and the base density in ONE read of the rows (tfk_rows_fma_gauss_logprob) -- z is never
        written.  None when that route does not apply."""
        from torchflows_amd import image_program, native
        from torchflows_amd.bijections.base import _params_ok, method_direction
        from torchflows_amd.bijections.finite.multiscale.base import MultiscaleBijection
        b = self.bijection
        if (context is not None or not isinstance(b, MultiscaleBijection) or not isinstance(self.base, DiagonalGaussian)
                or b.training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in b.parameters())))):
            return None
        x = x.to(self.get_device())
        if x.numel() == 0 or not native.eligible(x, self.base.loc, self.base.log_scale) or not _params_ok(self):
            return None
        d = method_direction(b.forward)
        prog = None if d is None else image_program.get_program(b, d, x.device)
        if prog is None:
            return None
        return image_program.log_prob(prog, x, self.event_shape, self.base.loc.detach(), self.base.log_scale.detach())

    def sample(self, sample_shape: Union[int, torch.Size, Tuple[int, ...]],
               context: torch.Tensor = None, no_grad: bool = False,
               return_log_prob: bool = False):
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        sample_shape = tuple(sample_shape)
        if context is not None:
            context = context.to(self.get_device())
            if tuple(get_batch_shape(context, self.context_shape)) != sample_shape:
                # one context row per conditioning case: draw sample_shape events for each.
                # (The reference's version of this branch, flows.py:687-692, draws z before
                # widening the shape and only runs when len(context) == event size.)
                n_ctx = len(context)
                context = context.expand(*sample_shape, *context.shape).contiguous()
                sample_shape = (*sample_shape, n_ctx)
        z = self.base_sample(sample_shape=sample_shape)
        z_in = z.view(*sample_shape, *self.bijection.event_shape)
        if return_log_prob and context is None and (no_grad or not torch.is_grad_enabled()):
            fused_out = self._fused_sample(z_in)
            if fused_out is not None:
                return fused_out
        if no_grad:
            with torch.no_grad():
                x, log_det = self.bijection.inverse(z_in.detach(), context=context)[:2]
        else:
            x, log_det = self.bijection.inverse(z_in, context=context)[:2]
        x = x.to(self.get_device())
        if return_log_prob:
            return x, self.base_log_prob(z) + log_det
        return x


class FlowMixture(BaseFlow):
    """Mixture of flows with categorical weights (reference flows.py:716-829).  OUTSIDE the hot-path scope of this
    build (SURVEY.md 2 lists it out of scope, 8(f) does not name it): plain host code over the components'
    ``Flow.log_prob`` / ``Flow.sample`` kept so that code written against the reference's ``torchflows.flows`` imports;
    it has no kernel, no entry point and no bench line of its own.  ``log_prob`` is a log-sum-exp over the components'
    ``Flow.log_prob``; ``sample`` draws from every component and keeps one per row (the reference's scheme, including
    its ``return_log_prob`` convention: the mixture of the components' sample log-probs)."""

    def __init__(self, flows, weights=None, trainable_weights: bool = False, constrain_weights: bool = False):
        super().__init__(event_shape=flows[0].event_shape)
        if weights is None:
            weights = [1.0 / len(flows)] * len(flows)
        if len(weights) != len(flows) or not all(w > 0.0 for w in weights) or abs(sum(weights) - 1.0) > 1e-8:
            raise AssertionError("weights must be positive, one per flow, and sum to 1")
        self.constrain_weights = constrain_weights
        self.flows = nn.ModuleList(flows)
        logits = torch.log(torch.tensor(weights))
        if trainable_weights:
            self.logit_weights = nn.Parameter(logits)
        else:
            self.logit_weights = logits

    @property
    def n_components(self) -> int:
        return len(self.flows)

    @property
    def weights(self) -> torch.Tensor:
        u = self.logit_weights
        if self.constrain_weights:
            u = torch.sigmoid(u) * 10.0 - 5.0              # squashed into [-5, 5]
        return torch.softmax(u, dim=0)

    @property
    def log_weights(self) -> torch.Tensor:
        return self.weights.log()

    def log_prob(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        per_flow = torch.stack([flow.log_prob(x, context=context) for flow in self.flows])
        lw = self.log_weights.to(per_flow.device).view(-1, *([1] * (per_flow.dim() - 1)))
        return torch.logsumexp(lw + per_flow, dim=0)

    def sample(self, sample_shape, context: torch.Tensor = None, no_grad: bool = False,
               return_log_prob: bool = False):
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        draws = [flow.sample(sample_shape, context=context, no_grad=no_grad, return_log_prob=True)
                 for flow in self.flows]
        xs = torch.stack([d[0] for d in draws])                       # (n_flows, *sample, *event)
        which = torch.distributions.Categorical(probs=self.weights).sample(sample_shape=sample_shape)
        pick = torch.nn.functional.one_hot(which, num_classes=len(draws)).movedim(-1, 0).to(xs.device)
        pick = pick.view(*pick.shape, *([1] * len(self.event_shape)))
        samples = torch.sum(pick * xs, dim=0)
        if not return_log_prob:
            return samples
        lps = torch.stack([d[1] for d in draws])
        lw = self.log_weights.to(lps.device).view(-1, *([1] * (lps.dim() - 1)))
        return samples, torch.logsumexp(lw + lps, dim=0)

    def regularization(self):
        return sum(flow.regularization() for flow in self.flows)
