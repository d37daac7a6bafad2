import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from torch.profiler import profile, ProfilerActivity
flow = bench.make_flow("AffineGlow", (3, 32, 32), 3).cuda()
x = torch.randn(512, 3, 32, 32, device="cuda")
with torch.no_grad():
    flow.log_prob(x)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        flow.log_prob(x)
        torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if "conv" in e.key.lower() or "Im2" in e.key]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:14]:
    print(f"{e.key[:40]:40s} n={e.count:4d} cuda={e.device_time_total/1e3:9.2f} ms shapes={str(e.input_shapes)[:110]}")
