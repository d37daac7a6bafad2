"""2-rank gloo worker for tests/test_host_cpu.py: data-parallel sharded_fit vs one process."""
import copy
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torchflows_amd as tfa  # noqa: E402
from torchflows_amd.distributed import shard_bounds, sharded_fit  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)                       # same initial weights on every rank
    flow = tfa.Flow(tfa.RealNVP(6, n_layers=2))
    g = torch.Generator().manual_seed(7)
    x = torch.randn(512, 6, generator=g) * torch.tensor([0.5, 1.0, 2.0, 1.0, 3.0, 0.2]) + 1.0
    lo, hi = shard_bounds(x.shape[0], rank, world)
    single = copy.deepcopy(flow)
    # plain SGD for the comparison: Adam turns last-bit differences of near-zero gradient entries
    # into +-lr steps, SGD is linear in the gradient
    sgd = lambda ps, lr: torch.optim.SGD(ps, lr=lr)
    losses = sharded_fit(flow, x[lo:hi], n_epochs=3, lr=0.01, batch_size=128, shuffle=False, optimizer=sgd)

    # one-process emulation of the same global batches: [rank 0's slice b ; rank 1's slice b]
    per = 128 // world
    bounds = [shard_bounds(x.shape[0], r, world) for r in range(world)]
    opt = torch.optim.SGD(single.parameters(), lr=0.01)
    single.train()
    ref_losses = []
    for _ in range(3):
        for b in range(0, bounds[0][1] - bounds[0][0], per):
            xb = torch.cat([x[l + b:min(l + b + per, h)] for l, h in bounds])
            opt.zero_grad()
            loss = -single.log_prob(xb).mean() / single.event_size + single.regularization()
            loss.backward()
            opt.step()
            ref_losses.append(float(loss.detach()))
    single.eval()
    assert len(losses) == len(ref_losses)
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-5, (losses[:3], ref_losses[:3])
    for (k, a), (_, b) in zip(flow.state_dict().items(), single.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), k
    # replicas stay identical
    flat = torch.cat([p.detach().reshape(-1) for p in flow.parameters()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(o, other[0]) for o in other)
    # the default optimizer (AdamW): replicas identical, loss goes down
    adam = tfa.Flow(tfa.RealNVP(6, n_layers=2))
    adam.load_state_dict(single.state_dict())
    adam_losses = sharded_fit(adam, x[lo:hi], n_epochs=3, lr=0.01, batch_size=128, shuffle=True, seed=3)
    flat = torch.cat([p.detach().reshape(-1) for p in adam.parameters()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(o, other[0]) for o in other)
    assert adam_losses[-1] < adam_losses[0]
    # shards whose step counts differ (513 rows: 257 + 256 at 64 rows per rank and step = 5 and 4 steps):
    # every rank must run the same number of collectives (this used to hang) and stay identical
    g2 = torch.Generator().manual_seed(11)
    x2 = torch.randn(513, 6, generator=g2)
    lo2, hi2 = shard_bounds(513, rank, world)
    torch.manual_seed(0)
    ragged = tfa.Flow(tfa.RealNVP(6, n_layers=2))
    single2 = copy.deepcopy(ragged)
    r_losses = sharded_fit(ragged, x2[lo2:hi2], n_epochs=2, lr=0.01, batch_size=128, shuffle=False, optimizer=sgd)
    assert len(r_losses) == 2 * 5, len(r_losses)
    bounds2 = [shard_bounds(513, r, world) for r in range(world)]
    opt2 = torch.optim.SGD(single2.parameters(), lr=0.01)
    single2.train()
    ref2 = []
    for _ in range(2):
        for b in range(0, 5 * per, per):
            xb = torch.cat([x2[min(l + b, h):min(l + b + per, h)] for l, h in bounds2])
            opt2.zero_grad()
            loss = -single2.log_prob(xb).mean() / single2.event_size + single2.regularization()
            loss.backward()
            opt2.step()
            ref2.append(float(loss.detach()))
    assert max(abs(a - b) for a, b in zip(r_losses, ref2)) < 2e-5, (r_losses, ref2)
    flat = torch.cat([p.detach().reshape(-1) for p in ragged.parameters()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(o, other[0]) for o in other)
    dist.barrier()
    if rank == 0:
        print("DIST_FIT_OK", losses[0], losses[-1])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
