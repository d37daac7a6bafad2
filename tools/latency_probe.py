#!/usr/bin/env python3
"""Small-batch latency of Flow.log_prob / Flow.sample on the HIP path (host overhead included)."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

for arch, D in (("RealNVP", 64), ("CouplingRQNSF", 64)):
    flow = bench.make_flow(arch, D, 8).cuda()
    for n in (1, 1024, 65536):
        x = torch.randn(n, D, device="cuda")
        with torch.no_grad():
            for _ in range(20):
                flow.log_prob(x)
                flow.sample((n,))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                flow.log_prob(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                flow.sample((n,))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"{arch}({D}) N={n}: log_prob {1e6 * (t1 - t0) / 200:.1f} us/call, sample {1e6 * (t2 - t1) / 50:.1f} us/call")
