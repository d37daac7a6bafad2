#!/usr/bin/env python3
"""Where the ~57 us of host time per small-batch Flow.log_prob call go (cProfile over 3000 calls)."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

flow = bench.make_flow("RealNVP", 64, 8).cuda()
x = torch.randn(1024, 64, device="cuda")
with torch.no_grad():
    for _ in range(50):
        flow.log_prob(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        flow.log_prob(x)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
