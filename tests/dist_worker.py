"""2-rank gloo worker for tests/test_host_cpu.py: batch-sharded Flow.log_prob + one all-reduce."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torchflows_amd as tfa  # noqa: E402
from torchflows_amd.distributed import (shard_bounds, sharded_log_likelihood,  # noqa: E402
                                       sharded_log_likelihood_async)


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)                       # same weights on every rank
    flow = tfa.Flow(tfa.RealNVP(6, n_layers=3)).eval()
    x = torch.randn(1001, 6, generator=torch.Generator().manual_seed(5))   # same global batch
    lo, hi = shard_bounds(x.shape[0], rank, world)
    with torch.no_grad():
        lp_local, total = sharded_log_likelihood(flow, x[lo:hi], chunk_rows=300)
        full = flow.log_prob(x)
    assert lp_local.shape == (hi - lo,)
    assert torch.allclose(lp_local, full[lo:hi], atol=1e-5)
    expect = full.double().sum()
    assert abs(float(total) - float(expect)) < 1e-6 * abs(float(expect)), (float(total), float(expect))
    # the in-flight variant (bench.py): three evaluations queued, sums valid after wait()
    with torch.no_grad():
        queued = [sharded_log_likelihood_async(flow, x[lo:hi], chunk_rows=300) for _ in range(3)]
    for lp_a, total_a, work in queued:
        work.wait()
        assert torch.equal(lp_a, lp_local) and float(total_a) == float(total)
    gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, total)
    assert all(float(g) == float(total) for g in gathered)      # every rank holds the same sum
    covered = sorted(shard_bounds(x.shape[0], r, world) for r in range(world))
    assert covered[0][0] == 0 and covered[-1][1] == x.shape[0]
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    dist.barrier()
    if rank == 0:
        print("DIST_OK", float(total))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
