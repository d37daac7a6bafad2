#!/usr/bin/env python3
"""Does Flow.fit capture the training step of an image flow into a hipGraph, and what does a replayed step cost?
   python tools/image_graph_probe.py [epochs] [glow]     (TORCHFLOWS_AMD_GRAPH=1 forces the capture, 0 forbids it)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])

from torchflows.flows import Flow  # noqa: E402
from torchflows.architectures import MultiscaleRealNVP  # noqa: E402

dev = torch.device("cuda:0")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
glow = len(sys.argv) > 2 and sys.argv[2] == "glow"          # AffineGlow((3, 32, 32)), config 5's model
g = torch.Generator().manual_seed(1)
img = torch.randn(1024, *((3, 32, 32) if glow else (1, 28, 28)), generator=g)
img = (img - img.mean()) / img.std()
torch.manual_seed(0)
if glow:
    from torchflows.bijections.finite.multiscale.architectures import AffineGlow
    flow = Flow(AffineGlow((3, 32, 32))).to(dev)
else:
    flow = Flow(MultiscaleRealNVP((1, 28, 28))).to(dev)
print("graph safe:", flow._graph_safe(), flush=True)
flow.fit(img, n_epochs=4)
torch.cuda.synchronize()
print("warm-up fit stats:", flow._fit_stats, flush=True)
t0 = time.perf_counter()
flow.fit(img, n_epochs=epochs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stats = flow._fit_stats
n = stats.get("eager_steps", 0) + stats.get("graph_replays", 0)
ok = all(bool(torch.isfinite(p).all()) for p in flow.parameters())
print(f"IMAGE_GRAPH epochs={n} wall_ms_per_epoch={1e3 * dt / max(n, 1):.3f} stats={stats} finite={ok}", flush=True)
