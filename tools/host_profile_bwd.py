#!/usr/bin/env python3
"""Host-side profile of the BACKWARD of the eager maximum-likelihood step (RealNVP-64, 2^18 rows): the autograd engine runs
torchflows_amd.autograd's backward on its own thread, which cProfile around loss.backward() does not see -- the profiler is
switched on inside the backward itself."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from torchflows_amd import autograd as tfa_autograd  # noqa: E402

flow = bench.make_flow("RealNVP", 64, 8).cuda()
x = torch.randn(1 << 18, 64, device="cuda")
flow.train()
opt = torch.optim.AdamW(flow.parameters(), lr=1e-4)
pr = cProfile.Profile()
fn_cls = [c for c in vars(tfa_autograd).values() if isinstance(c, type) and issubclass(c, torch.autograd.Function)
          and c is not torch.autograd.Function]
orig = {c: c.backward for c in fn_cls}
active = [False]


def wrap(c):
    f = orig[c]

    def backward(ctx, *grads):
        if not active[0]:
            return f(ctx, *grads)
        pr.enable()
        try:
            return f(ctx, *grads)
        finally:
            pr.disable()
    c.backward = staticmethod(backward)


for c in fn_cls:
    wrap(c)


def step():
    opt.zero_grad(set_to_none=True)
    loss = -flow.log_prob(x).mean() / flow.event_size + flow.regularization()
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
active[0] = True
for _ in range(30):
    step()
torch.cuda.synchronize()
print("autograd.Function classes:", [c.__name__ for c in fn_cls])
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
