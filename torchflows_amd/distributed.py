"""Multi-GPU evaluation: batch-sharded replicas, one all-reduce.

Rows of a batch are independent on this path (every op is per sample; the only
batch-coupled op, ActNorm's train-mode initialisation, does not run in eval mode), and the
weights are a few hundred KB, so every rank holds a full copy of the flow and evaluates its
own contiguous slice of the rows.  Nothing crosses GPUs except the summed log-likelihood:
one fp64 scalar per rank, all-reduced over RCCL/xGMI (``torch.distributed`` backend
``nccl`` on ROCm) -- 8 bytes, latency-bound.  Per-row ``log_prob`` stays sharded on device.
The reference has no multi-device code at all; this is new.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from torchflows_amd import native
from torchflows_amd.utils import make_adamw


def shard_bounds(n_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of ``n_rows`` rows owned by ``rank``."""
    base, extra = divmod(n_rows, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def local_sum_f64(log_prob: torch.Tensor) -> torch.Tensor:
    """fp64 sum of an fp32 vector as a 1-element tensor on its device (tfk_sum_f32 on HIP)."""
    if log_prob.device.type == "cuda" and log_prob.dtype == torch.float32:
        return native.sum_f32(log_prob.reshape(-1).contiguous())
    return log_prob.double().sum().reshape(1)


def _local_log_likelihood(flow, x_local, context, chunk_rows):
    n = x_local.shape[0]
    step = chunk_rows or n
    if step >= n:
        if hasattr(flow, "log_prob_and_sum"):        # the sum rides in the log_prob launch where it can
            return flow.log_prob_and_sum(x_local, context=context)
        lp = flow.log_prob(x_local, context=context)
    else:
        lp = torch.empty(n, dtype=x_local.dtype, device=x_local.device)
        for lo in range(0, n, step):
            c = None if context is None else context[lo:lo + step]
            lp[lo:lo + step] = flow.log_prob(x_local[lo:lo + step], context=c)
    return lp, local_sum_f64(lp)


def sharded_log_likelihood(flow, x_local: torch.Tensor, context: Optional[torch.Tensor] = None,
                           chunk_rows: Optional[int] = None, group=None
                           ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(log_prob of the local rows, sum over ALL ranks' rows)``.

    ``x_local`` is this rank's shard, already on its device.  ``chunk_rows`` bounds the rows
    per pass (the RQ-spline conditioner output is 2.9 KB per row per layer).  With no process
    group initialised this is the single-GPU evaluation plus its local sum."""
    lp, total = _local_log_likelihood(flow, x_local, context, chunk_rows)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return lp, total


class _Done:
    def wait(self):
        return True


def sharded_log_likelihood_async(flow, x_local: torch.Tensor, context: Optional[torch.Tensor] = None,
                                 chunk_rows: Optional[int] = None, group=None):
    """As :func:`sharded_log_likelihood`, but the all-reduce is left in flight on the collective's own
    stream: returns ``(log_prob, total, work)`` and ``total`` is only valid after ``work.wait()``.
    A caller that evaluates batch after batch waits for the sums at the end, so the 8-byte exchange
    (latency-bound on xGMI) and the rank-to-rank skew it would expose overlap with the next batch's
    kernels instead of stalling the compute stream every 0.5 ms."""
    lp, total = _local_log_likelihood(flow, x_local, context, chunk_rows)
    work = _Done()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return lp, total, work


# ---------------------------------------------------------------------------------------------
# data-parallel maximum-likelihood training (SURVEY.md 8(e) + 8(f)-2)
# ---------------------------------------------------------------------------------------------
def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class _GlobalActNormInit:
    """While active, ActNorm's data-dependent initialisation (layers.py:51-69) uses the statistics
    of the GLOBAL first batch: one all-reduce of (count, sum x, sum x^2) per ActNorm layer, once."""

    def __init__(self, flow, group):
        from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm
        self.layers = [m for m in flow.modules() if isinstance(m, ActNorm)]
        self.group = group

    def __enter__(self):
        group = self.group

        def make(layer):
            @torch.no_grad()
            def init(x):
                n_batch_dims = x.dim() - len(layer.event_shape)
                flat = x.reshape(-1, *x.shape[n_batch_dims:]).double()
                stats = torch.cat([flat.new_tensor([flat.shape[0]]), flat.sum(0).reshape(-1),
                                   flat.square().sum(0).reshape(-1)])
                dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
                n = stats[0]
                d = flat[0].numel()
                mean = stats[1:1 + d] / n
                var = (stats[1 + d:] - n * mean * mean) / torch.clamp(n - 1, min=1.0)   # unbiased, as torch.std
                scale = torch.sqrt(torch.clamp(var, min=0.0)) if float(n) > 1 else torch.ones_like(mean)
                shape = tuple(layer.event_shape)
                shift = mean.reshape(*shape, 1).to(layer.value)
                scale = scale.reshape(*shape, 1).to(layer.value)
                layer.value.copy_(torch.cat([layer.transformer.unconstrain_scale(scale), shift], dim=-1))
                layer.first_training_batch_pass = False
            return init

        for layer in self.layers:
            layer.__dict__["_data_dependent_init"] = make(layer)
        return self

    def __exit__(self, *exc):
        for layer in self.layers:
            layer.__dict__.pop("_data_dependent_init", None)
        return False


def sharded_fit(flow, x_local: torch.Tensor, n_epochs: int = 500, lr: float = 0.05,
                batch_size: int = 1024, shuffle: bool = True, w_local: Optional[torch.Tensor] = None,
                seed: int = 0, group=None, optimizer=None):
    """Data-parallel ``Flow.fit`` (flows.py:226-455 semantics for the loss: ``-mean(log_prob * w) /
    event_size + regularization`` over the GLOBAL batch): every rank holds a replica and its own
    shard of the training rows (already on its device), takes ``batch_size // world_size`` of them
    per step, and the ranks exchange exactly one all-reduce(SUM) per step -- the flat gradient with the
    loss riding along (a few tens of KB: latency-bound on xGMI) -- before the identical AdamW update:
    replicas stay bit-identical.  Shards may differ in size: the ranks exchange their row counts once,
    every rank runs the longest shard's number of steps, and a rank without rows left contributes an
    empty batch.
    ActNorm takes its initial statistics from the global first batch (one more all-reduce, once).
    ``optimizer``: ``callable(parameters, lr)`` (default AdamW, as ``Flow.fit``).
    Returns the list of global per-step training losses."""
    rank, world = _world(group)
    distributed = world > 1
    dev = flow.get_device()
    x_local = x_local.to(dev)
    n_local = x_local.shape[0]
    w_local = torch.ones(n_local, device=dev) if w_local is None else w_local.to(dev)
    local_bs = max(1, batch_size // world)
    params = [p for p in flow.parameters() if p.requires_grad]
    if not params:
        return []
    # Shard sizes may differ (shard_bounds hands out shards one row apart), so the ranks agree ONCE on
    # every rank's row count: the number of steps per epoch (the longest shard's) and the global row
    # count of every step follow from it on the host, and a rank that has run out of rows contributes
    # an empty batch (zero data gradient) instead of leaving the others waiting in the all-reduce.
    if distributed:
        sizes_t = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(sizes_t, torch.tensor([n_local], dtype=torch.int64, device=dev), group=group)
        sizes = [int(t.item()) for t in sizes_t]
    else:
        sizes = [n_local]
    if distributed and min(sizes) == 0:
        # such a rank would never enter flow.log_prob on the first step and so never join the per-ActNorm all-reduce of
        # the data-dependent initialisation the other ranks issue from inside it: the job would hang
        raise ValueError(f"sharded_fit: every rank needs at least one training row, got shard sizes {sizes}")
    steps_per_epoch = max(-(-n // local_bs) for n in sizes)
    opt = (optimizer or make_adamw)(flow.parameters(), lr)
    gen = torch.Generator(device="cpu").manual_seed(seed * 1000003 + rank)
    flow.train()
    # one slot per step, written on the device (no host sync per step, and no view that would keep each step's whole
    # flat gradient buffer alive until the end of training)
    losses = torch.empty(n_epochs * steps_per_epoch, dtype=params[0].dtype, device=dev)
    n_steps = 0
    stats = flow._fit_stats = {"exchanged_in_place": 0, "exchanged_after_a_copy": 0}     # (steps, by how the buffer came about)
    ctx = _GlobalActNormInit(flow, group) if distributed else None
    if ctx is not None:
        ctx.__enter__()
    try:
        for _ in range(n_epochs):
            order = torch.randperm(n_local, generator=gen).to(dev) if shuffle and n_local > 1 else None
            for step in range(steps_per_epoch):
                lo = step * local_bs
                count = sum(min(local_bs, max(0, n - lo)) for n in sizes)        # global batch rows (host, no sync)
                idx = slice(lo, lo + local_bs) if order is None else order[lo:lo + local_bs]
                xb, wb = x_local[idx], w_local[idx]
                opt.zero_grad(set_to_none=True)
                # (on the HIP path the L2 penalty is evaluated inside the chain's autograd node when the parameters live in
                # one buffer -- Flow._base_batch_loss does the same --, so that the gradients leave as slices of ONE buffer)
                b = flow.bijection
                folded = (xb.shape[0] > 0 and dev.type == "cuda" and hasattr(b, "_request_l2") and b._request_l2())
                lp = flow.log_prob(xb) if xb.shape[0] > 0 else None
                reg = None
                if folded:
                    b.__dict__.pop("_tfk_l2_request", None)
                    reg = b.__dict__.pop("_tfk_l2_out", None)
                loss = (reg if reg is not None else flow.regularization()) / world
                if lp is not None:
                    loss = loss - (lp * wb).sum() / (float(count) * flow.event_size)
                if isinstance(loss, torch.Tensor) and loss.requires_grad:
                    loss.backward()
                loss_t = loss.detach().reshape(1).to(dev) if isinstance(loss, torch.Tensor) \
                    else torch.tensor([float(loss)], device=dev)
                fb = getattr(opt, "flat", None)
                if (fb is not None and fb.intact() and len(params) == len(fb.params)
                        and all(p is q for p, q in zip(params, fb.params))):
                    # FlatAdamW: the exchange buffer IS the optimiser's gradient buffer (flat_optim.py): filled by the
                    # backward pass itself when the whole step ran on the fused launches (no copy at all), else from the
                    # per-tensor gradients; the loss rides in its tail padding.  Every rank takes this branch together.
                    G = fb.grads_are_flat()
                    stats["exchanged_in_place" if G is not None else "exchanged_after_a_copy"] += 1
                    if G is None:
                        G = torch.zeros(fb.n, dtype=params[0].dtype, device=dev)
                        for p, o, n in zip(fb.params, fb.offset, fb.numel):
                            if p.grad is not None and n:
                                G[o:o + n].copy_(p.grad.reshape(-1))
                        for p, o, n in zip(fb.params, fb.offset, fb.numel):
                            p.grad = G[o:o + n].view(p.shape)
                        fb.last_grad = G
                    slot = fb.zero_slot + 1
                    G[slot:slot + 1].copy_(loss_t.to(G.dtype))
                    if distributed:
                        dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)    # THE exchange step (the only one)
                    losses[n_steps].copy_(G[slot])
                    G[slot:slot + 1].zero_()                  # (the buffer's tail stays zero: the optimiser updates all of it)
                else:
                    stats["exchanged_after_a_copy"] += 1
                    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                      for p in params] + [loss_t.to(params[0].dtype)])
                    if distributed:
                        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)     # THE exchange step (the only one)
                    lo_f = 0
                    for p in params:
                        n = p.numel()
                        p.grad = flat[lo_f:lo_f + n].view_as(p)
                        lo_f += n
                    losses[n_steps].copy_(flat[-1])
                n_steps += 1
                opt.step()
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    flow.eval()
    return [float(v) for v in losses[:n_steps].cpu()]
