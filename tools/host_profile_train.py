#!/usr/bin/env python3
"""Host-side profile of the eager maximum-likelihood step (RealNVP-64, 2^18 rows): cProfile over 30 steps."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

flow = bench.make_flow("RealNVP", 64, 8).cuda()
x = torch.randn(1 << 18, 64, device="cuda")
w = torch.ones(1 << 18, device="cuda")
flow.train()
from torchflows_amd.utils import make_adamw
opt = make_adamw(flow.parameters(), 1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss = flow._base_batch_loss((x, w), reduction=torch.mean, use_regularization=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("ms per step:", 1e3 * (time.perf_counter() - t0) / 20)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
