#!/usr/bin/env python3
"""Experiment: a level of K copies of ONE step (same weights: scalar-cache hits) against the real level of K different
steps -- how much of the level launch is the scalar cache missing on a new coupling's weights every step?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_glow32
from torchflows_amd import image_program, native

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
flow, fx = load_glow32()
flow = flow.cuda()
dev = torch.device("cuda", 0)
prog = image_program.get_program(flow.bijection, 0, dev)
rows = torch.randn(N, 3072, device="cuda")
logdet = torch.zeros(N, device="cuda")


def time_level(lv, reps=3):
    native.glow_level(rows, rows, logdet, lv.row_idx, lv.blob_host, lv.blob_dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        native.glow_level(rows, rows, logdet, lv.row_idx, lv.blob_host, lv.blob_dev)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / reps


for lv in prog.levels:
    steps = prog.steps[lv.first: lv.first + lv.count]
    print(f"level {lv.count} steps D_level {lv.D_level}: real {time_level(lv):.0f} us", flush=True)
    for pick in sorted({0, lv.count // 2, lv.count - 1}):
        same = image_program.build_levels([steps[pick]] * lv.count, 3072, dev)
        print(f"    {lv.count} x step {lv.first + pick} ({steps[pick].info['kind']} {steps[pick].info['image']}): {time_level(same[0]):.0f} us;"
              f"  1 x: {time_level(image_program.build_levels([steps[pick]], 3072, dev)[0]):.0f} us", flush=True)
