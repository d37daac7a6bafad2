#!/usr/bin/env python3
"""The same fit as tools/fit_probe.py on the host's ATen path (what a reference user gets on the CPU): ms per step."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

torch.manual_seed(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ARCH = sys.argv[2] if len(sys.argv) > 2 else "RealNVP"
flow = bench.make_flow(ARCH, D, 8)
x = torch.randn(1 << 14, D) * 0.5 + 0.3
t0 = time.perf_counter()
flow.fit(x, n_epochs=2, lr=1e-3, shuffle=False)
dt = time.perf_counter() - t0
print(f"{ARCH}({D}) host fit ({torch.get_num_threads()} threads): {1e3 * dt / 32:.3f} ms per step of 1024 rows")
