#!/usr/bin/env python3
"""RealNVP with an even event size that is not 64 / 128 / 256 (N = 2^20): padded flow programs vs layer by layer."""
import os, sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
arch = sys.argv[1] if len(sys.argv) > 1 else "RealNVP"
sizes = [int(v) for v in sys.argv[2:]] or [8, 22, 62, 100, 3, 21, 43, 63]
for D in sizes:
    torch.manual_seed(0)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=8))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(4096, D))
    flow = flow.eval().cuda()
    x = torch.randn(1 << 20, D, device="cuda")
    for mode in ("1", "0"):
        os.environ["TORCHFLOWS_AMD_FUSED_PAD"] = mode
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        with torch.no_grad():
            from torchflows_amd.distributed import sharded_log_likelihood as sll
            sll(flow, x, chunk_rows=1 << 18); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): lp = sll(flow, x, chunk_rows=1 << 18)
            torch.cuda.synchronize()
        print(f"{arch}({D}) {'padded flow program' if mode == '1' else 'layer by layer'}: {(1 << 20) * 5 / (time.perf_counter() - t0):.3e} evals/s")
