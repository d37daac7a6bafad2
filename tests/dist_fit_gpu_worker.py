"""2-rank gloo worker for tests/test_gpu_train.py: sharded_fit with both replicas on ONE card (RCCL refuses two ranks per
device; the exchange step is the same all-reduce)."""
import copy
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torchflows_amd as tfa  # noqa: E402
from torchflows_amd.distributed import shard_bounds, sharded_fit  # noqa: E402
from torchflows_amd.utils import make_adamw  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    for arch, D in (("RealNVP", 6), ("RealNVP", 64), ("CouplingRQNSF", 64)):
        torch.manual_seed(0)                       # same initial weights on every rank
        flow = getattr(tfa, arch)(D, n_layers=2)
        flow = tfa.Flow(flow)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(1024, D, generator=g) * 0.7 + 0.5
        flow.train()
        with torch.no_grad():
            flow.log_prob(x[:256])                 # ActNorm statistics, the same on both ranks and in the reference run
        single = copy.deepcopy(flow).to(dev)
        flow = flow.to(dev)
        lo, hi = shard_bounds(x.shape[0], rank, world)
        losses = sharded_fit(flow, x[lo:hi].to(dev), n_epochs=3, lr=0.01, batch_size=256, shuffle=False)
        stats = dict(flow._fit_stats)
        # every step's gradients left the backward pass as ONE buffer and were exchanged in place
        want = {"exchanged_in_place": 12, "exchanged_after_a_copy": 0}
        assert stats == want, (arch, D, stats)
        flat = torch.cat([p.detach().reshape(-1) for p in flow.parameters()])
        other = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert all(torch.equal(o, other[0]) for o in other), (arch, D)          # replicas stay bit-identical
        assert losses[-1] < losses[0], (arch, D, losses)
        # one-process run over the same global batches [rank 0's slice b ; rank 1's slice b]
        per = 256 // world
        bounds = [shard_bounds(x.shape[0], r, world) for r in range(world)]
        opt = make_adamw(single.parameters(), 0.01)
        single.train()
        ref = []
        xs = x.to(dev)
        for _ in range(3):
            for b in range(0, bounds[0][1] - bounds[0][0], per):
                xb = torch.cat([xs[l + b:min(l + b + per, h)] for l, h in bounds])
                opt.zero_grad()
                loss = single._base_batch_loss((xb, torch.ones(len(xb), device=dev)))
                loss.backward()
                opt.step()
                ref.append(float(loss.detach()))
        # (Adam turns last-bit differences of near-zero gradient entries into +-lr steps: the first steps agree closely,
        # the later ones to what two different summation orders allow)
        assert max(abs(a - b) for a, b in zip(losses[:3], ref[:3])) < 5e-5, (arch, D, losses[:3], ref[:3])
        assert abs(losses[-1] - ref[-1]) < 5e-3 * max(1.0, abs(ref[-1])), (arch, D, losses[-1], ref[-1])
    dist.barrier()
    if rank == 0:
        print("DIST_FIT_GPU_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
