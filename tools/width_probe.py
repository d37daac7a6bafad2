#!/usr/bin/env python3
"""log_prob rate of RealNVP(D, 8 layers) at N = 2^20 for a few event sizes (per-element rate relative to D = 64)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
from torchflows_amd.distributed import sharded_log_likelihood as sll
ref = None
for D in [int(v) for v in sys.argv[1:]] or [64, 128, 100, 256]:
    torch.manual_seed(0)
    flow = tfa.Flow(tfa.RealNVP(D, n_layers=8))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(4096, D))
    flow = flow.eval().cuda()
    n = (1 << 20) if D <= 128 else (1 << 19)
    x = torch.randn(n, D, device="cuda")
    with torch.no_grad():
        for _ in range(10): sll(flow, x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): sll(flow, x)
        torch.cuda.synchronize()
    rate = n * 20 / (time.perf_counter() - t0)
    ref = ref or rate * 64
    print(f"RealNVP({D}): {rate:.3e} evals/s, per-element rate {rate * D / ref:.2f} x D=64's", flush=True)
