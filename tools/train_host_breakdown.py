#!/usr/bin/env python3
"""Where the host time of one eager maximum-likelihood step goes (RealNVP-64, 2^18 rows): enqueue time of the forward,
the backward and the optimizer (perf_counter without synchronising, the queue drained before each phase), and the
ATen / libtfk call counts of one step from torch.profiler."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from torchflows_amd import native  # noqa: E402
from torchflows_amd.utils import make_adamw  # noqa: E402

flow = bench.make_flow("RealNVP", 64, 8).cuda()
x = torch.randn(1 << 18, 64, device="cuda")
w = torch.ones(1 << 18, device="cuda")
flow.train()
opt = make_adamw(flow.parameters(), 1e-4)
sync = torch.cuda.synchronize


def phases(n=30):
    acc = [0.0] * 5
    for _ in range(n):
        sync()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        t1 = time.perf_counter()
        loss = flow._base_batch_loss((x, w), reduction=torch.mean, use_regularization=True)
        t2 = time.perf_counter()
        sync()
        t3 = time.perf_counter()
        loss.backward()
        t4 = time.perf_counter()
        sync()
        t5 = time.perf_counter()
        opt.step()
        t6 = time.perf_counter()
        sync()
        t7 = time.perf_counter()
        for i, v in enumerate((t1 - t0, t2 - t1, t4 - t3, t6 - t5, (t3 - t2) + (t5 - t4) + (t7 - t6))):
            acc[i] += v
    return [1e3 * a / n for a in acc]


def step():
    opt.zero_grad(set_to_none=True)
    loss = flow._base_batch_loss((x, w), reduction=torch.mean, use_regularization=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
sync()
t0 = time.perf_counter()
for _ in range(50):
    step()
sync()
print(f"eager step: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms")
z, f, b, o, drain = phases()
print(f"host enqueue per phase (ms): zero_grad {z:.3f}  forward+loss {f:.3f}  backward {b:.3f}  optimizer {o:.3f}  "
      f"(sum {z + f + b + o:.3f}); GPU drain after the phases {drain:.3f}")
before = native.calls
step()
print("libtfk launches per step:", native.calls - before)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(10):
        step()
    sync()
rows = [(e.key, e.count / 10, e.self_cpu_time_total / 10) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[2])
print("op, calls/step, self host us/step")
for k, c, t in rows[:45]:
    print(f"  {k[:60]:60s} {c:7.1f} {t:9.1f}")
print("total calls/step:", sum(r[1] for r in rows), " self host us/step:", sum(r[2] for r in rows))
