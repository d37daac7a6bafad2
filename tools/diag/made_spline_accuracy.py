#!/usr/bin/env python3
"""Where does the elementwise accuracy of the layer-by-layer MADE spline route go?  For the sibling fixtures
(flow_malrs5 / flow_marqnsf5 / flow_iarqnsf5, "init" weights) the 1e-5 pass rate of z against the reference evaluated in fp64
is printed for: the product path; the transform kernel fed with the conditioner output computed on the HOST (= the
reference's own fp32 h); the kernel fed with the conditioner evaluated in fp64 on the device and rounded once."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, state_dict_of   # noqa: E402
import torchflows_amd as tfa                       # noqa: E402
from torchflows_amd import native                  # noqa: E402


def pass_rate(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.mean(np.abs(a - b) <= 1e-5 * np.maximum(1.0, np.abs(b))))


def chain(flow, x, mode):
    """forward through the layers; MADE layers per `mode`."""
    from torchflows_amd.bijections.finite.autoregressive.layers_base import MaskedAutoregressiveBijection
    import copy
    cur = x
    for layer in flow.bijection.layers:
        if not isinstance(layer, MaskedAutoregressiveBijection) or mode == "product":
            cur, _ = layer.forward(cur)
            continue
        rows = cur.contiguous()
        N, D = rows.shape
        if mode == "host_h":
            host = copy.deepcopy(layer).cpu()
            h = host.conditioner_transform(rows.cpu(), None).reshape(N, -1).contiguous().cuda()
        else:
            dbl = copy.deepcopy(layer.conditioner_transform).double()
            h = dbl(rows.double(), None).float().reshape(N, -1).contiguous()
        out = torch.empty_like(rows)
        ld = torch.zeros(N, device=rows.device)
        tr = layer.transformer
        fn = native.rqs_coupling if tr.native_kind == "rqs" else native.lrs_coupling
        fn(rows, h, out, ld, None, D, tr.n_bins, tr.boundary, accumulate=False, inverse=False)
        cur = out
    return cur


for name, arch in (("flow_malrs5.npz", "MaskedAutoregressiveLRS"), ("flow_marqnsf5.npz", "MaskedAutoregressiveRQNSF")):
    fx = load_golden(name)
    torch.manual_seed(0)
    flow = tfa.Flow(getattr(tfa, arch)(5, n_layers=2))
    flow.load_state_dict({k: torch.from_numpy(v) for k, v in state_dict_of(fx, "init").items()})
    flow = flow.cuda().eval()
    x = torch.from_numpy(fx["x"]).cuda()
    z64, zref = fx["init/z64"], fx["init/z"]
    print(name, "reference fp32 vs fp64:", round(pass_rate(zref, z64), 4))
    with torch.no_grad():
        for mode in ("product", "host_h", "fp64_h"):
            z = chain(flow, x, mode).cpu().numpy()
            print(f"   {mode:8s}: vs fp64 {pass_rate(z, z64):.4f}   vs the reference's fp32 value {pass_rate(z, zref):.4f}")
