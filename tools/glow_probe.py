#!/usr/bin/env python3
"""AffineGlow (3,32,32): log_prob throughput with different MIOpen / chunking settings."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from torchflows_amd.distributed import sharded_log_likelihood
flow = bench.make_flow("AffineGlow", (3, 32, 32), 3).cuda()
x = torch.randn(1 << 15, 3, 32, 32, device="cuda")
def run(chunk, n=2):
    with torch.no_grad():
        sharded_log_likelihood(flow, x, chunk_rows=chunk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): sharded_log_likelihood(flow, x, chunk_rows=chunk)
        torch.cuda.synchronize()
    return x.shape[0] * n / (time.perf_counter() - t0)
for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    for chunk in (1 << 11, 1 << 13, 1 << 15):
        print(f"benchmark={bm} chunk={chunk}: {run(chunk):.3e} evals/s", flush=True)
with torch.no_grad():
    flow_cl = flow.to(memory_format=torch.channels_last)
    print(f"channels_last weights, chunk 8192: {run(1 << 13):.3e} evals/s")
