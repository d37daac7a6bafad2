#!/usr/bin/env python3
"""log_prob of MultiscaleRealNVP((1, 28, 28)) in eval mode (odd plane sizes below the first level: the image compiler declines,
the ConvModifiers there are 2-wide convolutions): time per call on 4 096 images, libtfk launches, and parity with the host.
   python tools/mnist_infer_probe.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from torchflows.flows import Flow  # noqa: E402
from torchflows.architectures import MultiscaleRealNVP  # noqa: E402
from torchflows_amd import native  # noqa: E402

torch.manual_seed(0)
flow = Flow(MultiscaleRealNVP((1, 28, 28)))
flow.train()
with torch.no_grad():
    flow.log_prob(torch.randn(256, 1, 28, 28))
flow.eval()
x = torch.randn(4096, 1, 28, 28)
with torch.no_grad():
    want = flow.log_prob(x[:64])
flow = flow.cuda()
xd = x.cuda()
with torch.no_grad():
    got = flow.log_prob(xd[:64])
    for _ in range(3):
        flow.log_prob(xd)
    torch.cuda.synchronize()
    before = native.calls
    t0 = time.perf_counter()
    for _ in range(10):
        flow.log_prob(xd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
rel = float(((got.cpu() - want).abs() / want.abs().clamp_min(1.0)).max())
print(f"MNIST_INFER ms_per_call={1e3 * dt:.3f} evals_per_s={4096 / dt:.3e} libtfk_launches={(native.calls - before) / 10:.0f} "
      f"log_prob_rel_vs_host={rel:.2e}", flush=True)
