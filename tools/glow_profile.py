#!/usr/bin/env python3
"""Per-kernel breakdown of one steady-state 8192-row AffineGlow(3,32,32) log_prob chunk (torch profiler)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from torch.profiler import profile, ProfilerActivity
flow = bench.make_flow("AffineGlow", (3, 32, 32), 3).cuda()
x = torch.randn(1 << 13, 3, 32, 32, device="cuda")
with torch.no_grad():
    for _ in range(3): flow.log_prob(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        flow.log_prob(x); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name == "CUDA"]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows); cnt = sum(e.count for e in rows)
print(f"total {tot / 1e3:.2f} ms over {cnt} kernels")
for e in rows[:40]:
    print(f"{e.device_time_total / 1e3:8.3f} ms {e.count:5d}  {e.key[:110]}")
