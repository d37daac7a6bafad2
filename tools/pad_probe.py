#!/usr/bin/env python3
"""Event sizes that are not 64 / 128 / 256 (N = 2^20 rows): padded flow programs with the rows read as they are
(tfk_flow_run_mfma_in), with the host-side padding pass (TORCHFLOWS_AMD_DEBUG=narrow_in=0), and layer by layer; the
per-element rate is compared with D = 64's (within 1.3x)."""
import os, sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
arch = sys.argv[1] if len(sys.argv) > 1 else "RealNVP"
sizes = [int(v) for v in sys.argv[2:]] or [64, 8, 22, 62, 100, 3, 21, 43, 63]
ref_rate = None
for D in sizes:
    torch.manual_seed(0)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=8))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(4096, D))
    flow = flow.eval().cuda()
    x = torch.randn(1 << 20, D, device="cuda")
    for mode, env in (("rows as they are", dict(TORCHFLOWS_AMD_DEBUG="fused_pad=1,narrow_in=1", TORCHFLOWS_AMD_FUSED="1")),
                      ("host padding pass", dict(TORCHFLOWS_AMD_DEBUG="fused_pad=1,narrow_in=0", TORCHFLOWS_AMD_FUSED="1")),
                      ("layer by layer", dict(TORCHFLOWS_AMD_DEBUG="fused_pad=0", TORCHFLOWS_AMD_FUSED="0" if D == 64 else "1"))):
        os.environ.update(env)
        flow.invalidate_native_caches()
        import warnings
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from torchflows_amd.distributed import sharded_log_likelihood as sll
            sll(flow, x, chunk_rows=1 << 20); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): lp = sll(flow, x, chunk_rows=1 << 20)
            torch.cuda.synchronize()
        rate = (1 << 20) * 20 / (time.perf_counter() - t0)
        if D == 64 and ref_rate is None:
            ref_rate = rate * 64
        extra = f", per-element rate {rate * D / ref_rate:.2f} x D=64's" if ref_rate else ""
        print(f"{arch}({D}) {mode}: {rate:.3e} evals/s{extra}", flush=True)
    os.environ["TORCHFLOWS_AMD_FUSED"] = "1"
