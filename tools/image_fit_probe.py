#!/usr/bin/env python3
"""Where an image flow's training step spends its time: wall time per step of ``Flow.fit`` on the reference's
image workload (image_modeling.ipynb:91: MultiscaleRealNVP((1, 28, 28)), 1 000 images = one AdamW step per epoch)
against the GPU's busy time for the same steps (HIP events around each step = wall; run under
``rocprofv3 --kernel-trace --stats`` for the kernel sum and the launch count).

    python tools/image_fit_probe.py [epochs] [val]      val=1: the notebook's validation split + early stopping
"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])

from torchflows.flows import Flow  # noqa: E402
from torchflows.architectures import MultiscaleRealNVP  # noqa: E402

dev = torch.device("cuda:0")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
with_val = len(sys.argv) > 2 and sys.argv[2] == "1"

g = torch.Generator().manual_seed(1)
img = torch.randn(1200, 1, 28, 28, generator=g)
img = (img - img.mean()) / img.std()
xt, xv = img[:1000], img[1000:]
torch.manual_seed(0)
flow = Flow(MultiscaleRealNVP((1, 28, 28))).to(dev)
kw = dict(x_val=xv, early_stopping=True) if with_val else {}
flow.fit(xt, n_epochs=3, **kw)                 # lazy state
torch.cuda.synchronize()
t0 = time.perf_counter()
flow.fit(xt, n_epochs=epochs, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stats = getattr(flow, "_fit_stats", {})
n = stats.get("eager_steps", 0) + stats.get("graph_replays", 0)
print(f"IMAGE_FIT epochs={n} wall_ms_per_epoch={1e3 * dt / max(n, 1):.3f} stats={stats}", flush=True)

# one bare training step, timed on the host and on the device
flow.train()
x = xt.to(dev)
w = torch.ones(len(x), device=dev)
opt = flow._optimizer
for _ in range(2):
    opt.zero_grad()
    flow._base_batch_loss((x, w), reduction=torch.mean, use_regularization=True).backward()
    opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 10
t0 = time.perf_counter()
e0.record()
for _ in range(K):
    opt.zero_grad()
    loss = flow._base_batch_loss((x, w), reduction=torch.mean, use_regularization=True)
    loss.backward()
    opt.step()
e1.record()
host_ms = 1e3 * (time.perf_counter() - t0) / K          # host time to ENQUEUE the steps
torch.cuda.synchronize()
print(f"IMAGE_STEP enqueue_ms={host_ms:.3f} device_span_ms={e0.elapsed_time(e1) / K:.3f}", flush=True)
