#!/usr/bin/env python3
"""Sweep the launch shape (slots, block, channel groups) of tfk_glow_coupling for every distinct coupling geometry of
AffineGlow((3,32,32)):   python tools/glow_tune.py [rows] [reps]"""
import copy
import ctypes as C
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_glow32          # noqa: E402
from torchflows_amd import image_program, native   # noqa: E402


def clone(layer, **kw):
    new = native.GlowLayer()
    C.memmove(C.byref(new), C.byref(layer), C.sizeof(layer))
    for k, v in kw.items():
        setattr(new, k, v)
    return new


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    flow, fx = load_glow32()
    flow = flow.cuda()
    prog = image_program.get_program(flow.bijection, 0, torch.device("cuda", 0))
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.randn(N, 3072, device="cuda", generator=g)
    logdet = torch.zeros(N, device="cuda")
    seen = {}
    for step in prog.steps:
        key = (step.info["kind"], step.info["image"])
        if key in seen:
            continue
        seen[key] = True
        results = []
        for block, slots, cg1, cg2 in itertools.product((256, 512, 1024), range(1, 17), (8, 4, 2), (8, 4, 2)):
            L = clone(step.layer, slots=slots, block=block, cg1=cg1, cg2=cg2)
            try:
                plan = native.glow_plan(L, 3072)
            except native.NativeError:
                continue
            if os.environ.get("TUNE_FAST") and (cg1, cg2) not in ((8, 8), (8, 4), (4, 4), (4, 2)):
                continue
            native.glow_coupling(rows, logdet, L, step.inverse)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(reps):
                native.glow_coupling(rows, logdet, L, step.inverse)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
            results.append((us, block, slots, cg1, cg2, plan["lds_bytes"], plan["tile_rows"]))
            rows.normal_(generator=g)
        results.sort()
        print(key, "default", {k: step.info[k] for k in ("slots", "block", "cg1", "cg2")}, flush=True)
        for r in results[:12]:
            print("   %8.1f us  block %4d slots %2d cg %d/%d lds %6d tile_rows %2d" % r, flush=True)
        print("   ... worst %8.1f us" % results[-1][0], flush=True)


if __name__ == "__main__":
    main()
