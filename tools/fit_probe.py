#!/usr/bin/env python3
"""Wall time of Flow.fit at the reference's default batch size (1024) on the device: eager steps against the captured step
(TORCHFLOWS_AMD_GRAPH=0 / auto).  argv[2] or RealNVP (D = argv[1] or 64, 8 layers), 65 536 training rows, 3 epochs = 192 steps."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

torch.manual_seed(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ARCH = sys.argv[2] if len(sys.argv) > 2 else "RealNVP"
base = bench.make_flow(ARCH, D, 8).cuda()
x = torch.randn(1 << 16, D, device="cuda") * 0.5 + 0.3
for mode in ("0", "auto", "0", "auto"):
    os.environ["TORCHFLOWS_AMD_GRAPH"] = mode
    flow = copy.deepcopy(base)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flow.fit(x, n_epochs=3, lr=1e-3, shuffle=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with torch.no_grad():
        lp = float(flow.log_prob(x).mean())
    print(f"{ARCH}({D}) TORCHFLOWS_AMD_GRAPH={mode:5s} fit: {dt:.3f} s for 192 steps = {1e3 * dt / 192:.3f} ms per step, stats {flow._fit_stats}, "
          f"mean log-likelihood {lp:.4f}", flush=True)
