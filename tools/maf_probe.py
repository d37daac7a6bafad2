#!/usr/bin/env python3
"""MAF(64, 8 layers) sampling = the sequential map: one launch per layer vs the reference's D passes.
Usage: maf_probe.py [MAF|MaskedAutoregressiveRQNSF]"""
import os, sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
torch.manual_seed(0)
arch = sys.argv[1] if len(sys.argv) > 1 else "MAF"
flow = tfa.Flow(getattr(tfa, arch)(64, n_layers=8))
flow.train()
with torch.no_grad():
    flow.log_prob(torch.randn(4096, 64))
flow = flow.eval().cuda()
for n in (1024, 1 << 16, 1 << 20):
    for mode in ("1", "0"):
        if mode == "0" and n > (1 << 16):
            continue
        os.environ["TORCHFLOWS_AMD_DEBUG"] = "made_fused=" + str(mode)
        with torch.no_grad():
            flow.sample((n,))
            torch.cuda.synchronize()
            reps = 5 if mode == "1" else 1
            t0 = time.perf_counter()
            for _ in range(reps):
                flow.sample((n,))
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"N={n}: {'one launch per layer' if mode == '1' else 'D passes per layer'}: {1e3 * dt:.2f} ms per call, {n / dt:.3e} samples/s")
os.environ["TORCHFLOWS_AMD_DEBUG"] = "made_fused=" + str("1")
x = torch.randn(1 << 20, 64, device="cuda")
with torch.no_grad():
    flow.log_prob(x); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): flow.log_prob(x)
    torch.cuda.synchronize()
print(f"log_prob (parallel map) N=2^20: {(1 << 20) * 5 / (time.perf_counter() - t0):.3e} evals/s")
