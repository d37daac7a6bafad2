#!/usr/bin/env python3
"""AffineGlow((3,32,32)): the level launches (tfk_glow_level: rows in the LDS, one launch per level) against the
one-launch-per-coupling route and the reference's fixture; per-launch HIP-event times.
    python tools/glow_level_probe.py [rows] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_glow32          # noqa: E402
from torchflows_amd import image_program, native   # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1, np.abs(b))))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    flow, fx = load_glow32()
    flow = flow.cuda()
    dev = torch.device("cuda", 0)
    x, z_in = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["z_in"]).cuda()
    for d, name in ((0, "forward"), (1, "inverse")):
        prog = image_program.get_program(flow.bijection, d, dev)
        inp = x if d == 0 else z_in
        with torch.no_grad():
            out_l, ld_l = image_program.run(prog, inp, (3, 32, 32))
            levels, prog.levels = prog.levels, None
            out_s, ld_s = image_program.run(prog, inp, (3, 32, 32))
            prog.levels = levels
        want, want_ld = (fx["z"], fx["log_det"]) if d == 0 else (fx["x_inv"], fx["log_det_inv"])
        print(f"{name}: levels vs reference: out {rel(out_l.cpu().numpy(), want):.3g} log_det {rel(ld_l.cpu().numpy(), want_ld):.3g};"
              f"  steps vs reference: out {rel(out_s.cpu().numpy(), want):.3g} log_det {rel(ld_s.cpu().numpy(), want_ld):.3g};"
              f"  levels vs steps: out {rel(out_l.cpu().numpy(), out_s.cpu().numpy()):.3g} "
              f"log_det {rel(ld_l.cpu().numpy(), ld_s.cpu().numpy()):.3g}", flush=True)
    with torch.no_grad():
        lp = flow.log_prob(x)
    print("log_prob vs reference:", rel(lp.cpu().numpy(), fx["log_prob"]))
    # odd batch sizes (tail tiles)
    with torch.no_grad():
        for n in (1, 3, 5, 63):
            a = flow.log_prob(x[:n])
            assert os.environ.get("TFK_GLOW_SKIP", "0") != "0" or rel(a.cpu().numpy(), fx["log_prob"][:n]) < 1e-5, n
    print("tail tiles ok")
    g = torch.Generator(device="cuda").manual_seed(5)
    xs = torch.randn(N, 3, 32, 32, device="cuda", generator=g)
    prog = image_program.get_program(flow.bijection, 0, dev)
    for mode in ("levels", "steps"):
        saved = prog.levels
        if mode == "steps":
            prog.levels = None
        with torch.no_grad():
            for _ in range(2):
                flow.log_prob(xs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                flow.log_prob(xs)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
        prog.levels = saved
        print(f"log_prob [{mode}]: {N} rows in {dt * 1e3:.2f} ms = {N / dt:.3e} evals/s", flush=True)
    rows = xs.reshape(N, -1).clone()
    logdet = torch.zeros(N, device="cuda")
    total = 0.0
    for lv in prog.levels:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        native.glow_level(rows, rows, logdet, lv.row_idx, lv.blob_host, lv.blob_dev)
        ev[0].record()
        for _ in range(reps):
            native.glow_level(rows, rows, logdet, lv.row_idx, lv.blob_host, lv.blob_dev)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
        total += us
        print(f"  level of {lv.count} steps on {lv.D_level} elements {lv.info}: {us:9.1f} us ({us * 1e3 / N:.1f} ns/row)")
    print(f"  sum of level launches: {total / 1e3:.2f} ms = {N / total * 1e6:.3e} evals/s")
    total = 0.0
    for i, step in enumerate(prog.steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        native.glow_coupling(rows, logdet, step.layer, step.inverse)
        ev[0].record()
        for _ in range(reps):
            native.glow_coupling(rows, logdet, step.layer, step.inverse)
        ev[1].record()
        torch.cuda.synchronize()
        total += ev[0].elapsed_time(ev[1]) * 1e3 / reps
    print(f"  sum of the 19 coupling launches: {total / 1e3:.2f} ms = {N / total * 1e6:.3e} evals/s")


if __name__ == "__main__":
    main()
