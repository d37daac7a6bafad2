#!/usr/bin/env python3
"""Experiment: capture one training step in a hipGraph (torch.cuda.CUDAGraph) and replay it."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

flow = bench.make_flow("RealNVP", 64, 8).cuda()
x = torch.randn(1 << 18, 64, device="cuda")
flow.train()
opt = torch.optim.AdamW(flow.parameters(), lr=1e-4, capturable=True)


def step():
    opt.zero_grad(set_to_none=True)
    loss = -flow.log_prob(x).mean() / flow.event_size + flow.regularization()
    loss.backward()
    opt.step()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warm-up done", flush=True)
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
stage = sys.argv[1] if len(sys.argv) > 1 else "full"
with torch.cuda.graph(graph):
    if stage == "fwd":
        with torch.no_grad():
            loss = flow.log_prob(x).mean()
    elif stage == "fwdgrad":
        loss = -flow.log_prob(x).mean() / flow.event_size + flow.regularization()
    elif stage == "bwd":
        loss = -flow.log_prob(x).mean() / flow.event_size + flow.regularization()
        loss.backward()
    else:
        loss = step()
print("captured", stage, flush=True)
graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    graph.replay()
torch.cuda.synchronize()
print(f"{stage}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per replay, loss {float(loss):.5f}")
