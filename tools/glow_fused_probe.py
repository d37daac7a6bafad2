#!/usr/bin/env python3
"""AffineGlow((3,32,32)) on the one-launch-per-coupling image program: parity against the reference's fixture and
per-launch times (HIP events).   python tools/glow_fused_probe.py [rows] [reps]"""
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    flow, fx = load_glow32()
    flow = flow.cuda()
    x, z_in = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["z_in"]).cuda()
    with torch.no_grad():
        before = native.calls
        lp = flow.log_prob(x)
        print("launches per log_prob:", native.calls - before)
        z, ld = flow.bijection.forward(x)
        xr, ldr = flow.bijection.inverse(z_in)
    print("parity vs reference: log_prob %.3g  z %.3g  log_det %.3g  x_inv %.3g  log_det_inv %.3g" % (
        rel(lp.cpu().numpy(), fx["log_prob"]), rel(z.cpu().numpy(), fx["z"]), rel(ld.cpu().numpy(), fx["log_det"]),
        rel(xr.cpu().numpy(), fx["x_inv"]), rel(ldr.cpu().numpy(), fx["log_det_inv"])))
    prog = image_program.get_program(flow.bijection, 0, x.device)
    g = torch.Generator(device="cuda").manual_seed(5)
    xs = torch.randn(N, 3, 32, 32, device="cuda", generator=g)
    with torch.no_grad():
        for _ in range(2):
            flow.log_prob(xs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            flow.log_prob(xs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print(f"log_prob: {N} rows in {dt * 1e3:.2f} ms = {N / dt:.3e} evals/s")
    rows = xs.reshape(N, -1).clone()
    logdet = torch.zeros(N, device="cuda")
    total = 0.0
    for i, step in enumerate(prog.steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        native.glow_coupling(rows, logdet, step.layer, step.inverse)
        ev[0].record()
        for _ in range(reps):
            native.glow_coupling(rows, logdet, step.layer, step.inverse)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
        total += us
        info = step.info
        print(f"  step {i:2d} {info['kind']:8s} image {info['image']} slots {info['slots']:2d} block {info['block']:4d} "
              f"cg {info['cg1']}/{info['cg2']} lds {info['lds_bytes']:6d}: {us:9.1f} us  ({us * 1e3 / N:.1f} ns/row)")
    print(f"  sum of coupling launches: {total / 1e3:.2f} ms = {N / total * 1e6:.3e} evals/s")


if __name__ == "__main__":
    main()
