#!/usr/bin/env python3
"""Per-kernel breakdown of one maximum-likelihood training step (torch profiler): train_profile.py [arch] [D]."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from torch.profiler import profile, ProfilerActivity
arch = sys.argv[1] if len(sys.argv) > 1 else "CouplingRQNSF"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
flow = bench.make_flow(arch, D, 8).cuda()
x = torch.randn(1 << 18, D, device="cuda")
flow.train()
opt = torch.optim.AdamW(flow.parameters(), lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss = -flow.log_prob(x).mean() / flow.event_size + flow.regularization()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name == "CUDA"]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print(f"{arch}({D}): {tot / 1e3:.2f} ms of GPU time over {sum(e.count for e in rows)} kernels")
for e in rows[:25]:
    print(f"{e.device_time_total / 1e3:8.3f} ms {e.count:5d}  {e.key[:120]}")
if len(sys.argv) > 3 and sys.argv[3] == "cpu":
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
