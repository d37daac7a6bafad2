#!/usr/bin/env python3
"""One coupling step of AffineGlow((3,32,32)) repeated (for rocprofv3 / PMC passes and launch-shape experiments):
    python tools/glow_step_bench.py <step index> [rows] [reps] [slots block cg1 cg2]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_glow32          # noqa: E402
from torchflows_amd import image_program, native   # noqa: E402


def main():
    i = int(sys.argv[1])
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    flow, _ = load_glow32()
    flow = flow.cuda()
    prog = image_program.get_program(flow.bijection, 0, torch.device("cuda", 0))
    step = prog.steps[i]
    L = native.GlowLayer()
    C.memmove(C.byref(L), C.byref(step.layer), C.sizeof(L))
    if len(sys.argv) > 7:
        L.slots, L.block, L.cg1, L.cg2 = (int(v) for v in sys.argv[4:8])
    plan = native.glow_plan(L, 3072)
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.randn(N, 3072, device="cuda", generator=g)
    logdet = torch.zeros(N, device="cuda")
    native.glow_coupling(rows, logdet, L, step.inverse)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        native.glow_coupling(rows, logdet, L, step.inverse)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    print(f"step {i} {step.info['kind']} image {step.info['image']} {plan}: {us:.1f} us per launch, {us * 1e3 / N:.2f} ns/row")


if __name__ == "__main__":
    main()
