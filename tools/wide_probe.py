#!/usr/bin/env python3
"""RealNVP(64, 8 layers) with wider conditioners: flow program vs layer-by-layer, 2^20 rows."""
import os, sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
x = torch.randn(1 << 20, 64, device="cuda")
for H in (9, 16, 32, 64):
    torch.manual_seed(0)
    flow = tfa.Flow(tfa.RealNVP(64, n_layers=8, conditioner_kwargs=dict(n_hidden=H)))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(4096, 64))
    flow = flow.eval().cuda()
    res = []
    for fused in ("1", "0"):
        os.environ["TORCHFLOWS_AMD_FUSED"] = fused
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        with torch.no_grad():
            for _ in range(3):
                flow.log_prob(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                flow.log_prob(x)
            torch.cuda.synchronize()
        res.append((1 << 20) * 10 / (time.perf_counter() - t0))
    os.environ["TORCHFLOWS_AMD_FUSED"] = "1"
    print(f"hidden {H}: flow program {res[0]:.3e} evals/s, layer by layer {res[1]:.3e} evals/s")
