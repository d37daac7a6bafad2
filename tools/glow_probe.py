#!/usr/bin/env python3
"""AffineGlow (3,32,32): log_prob throughput (chunks of 8192 rows); run with MIOpen solver switches in
the environment to compare, e.g. MIOPEN_DEBUG_CONV_GEMM=0."""
import os, sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from torchflows_amd.distributed import sharded_log_likelihood
flow = bench.make_flow("AffineGlow", (3, 32, 32), 3).cuda()
x = torch.randn(1 << 15, 3, 32, 32, device="cuda")
with torch.no_grad():
    sharded_log_likelihood(flow, x, chunk_rows=1 << 13)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): lp, _ = sharded_log_likelihood(flow, x, chunk_rows=1 << 13)
    torch.cuda.synchronize()
print(f"MIOPEN_DEBUG_CONV_GEMM={os.environ.get('MIOPEN_DEBUG_CONV_GEMM')}: {x.shape[0] * 2 / (time.perf_counter() - t0):.3e} evals/s, checksum {float(lp.double().sum()):.6f}")
