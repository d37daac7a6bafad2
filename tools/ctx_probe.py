#!/usr/bin/env python3
"""Conditional flows (log_prob(x, context=c)) at N = 2^20 rows: evals/s per preset (tools/ctx_probe.py [C])."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 1 << 20
for arch, D in (("RealNVP", 64), ("CouplingRQNSF", 64), ("RealNVP", 22), ("CouplingRQNSF", 22)):
    torch.manual_seed(0)
    flow = tfa.Flow(getattr(tfa, arch)(D, context_shape=(C,), n_layers=8))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(4096, D), context=torch.randn(4096, C))
    flow = flow.eval().cuda()
    x, c = torch.randn(N, D, device="cuda"), torch.randn(N, C, device="cuda")
    from torchflows_amd import native
    with torch.no_grad():
        before = native.calls
        lp = flow.log_prob(x, context=c)
        launches = native.calls - before
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): lp = flow.log_prob(x, context=c)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{arch}({D}) context {C}: {N / dt:.3e} evals/s, {dt * 1e3:.3f} ms per call, {launches} libtfk launches", flush=True)
