#!/usr/bin/env python3
"""Profiling probe for the training step (used under rocprofv3, see profiles/README.md):
RealNVP / NSF on 2^18 resident rows, N maximum-likelihood steps, prints ms per step."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "RealNVP"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
flow = bench.make_flow(arch, D, 8).cuda()
x = torch.randn(1 << 18, D, device="cuda")
flow.train()
from torchflows_amd.utils import make_adamw
opt = make_adamw(flow.parameters(), 1e-4)          # what Flow.fit builds (on the device: flat_optim.FlatAdamW)
w = torch.ones(x.shape[0], device="cuda")


def step():                                        # the step Flow.fit runs
    opt.zero_grad(set_to_none=True)
    loss = flow._base_batch_loss((x, w), reduction=torch.mean, use_regularization=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{arch}({D}): {1e3 * (time.perf_counter() - t0) / steps:.3f} ms per step")
