#!/usr/bin/env python3
"""Training step of config 5's model, AffineGlow((3, 32, 32)), at Flow.fit's default batch size (1 024 images), three ways:
   the library route (ConvNet conditioner on ATen / MIOpen: TORCHFLOWS_AMD_DEBUG=convnet_train=0), the libtfk route eager,
   and the libtfk route captured into a hipGraph by Flow.fit.  Prints one line ``GLOW_TRAIN_JSON {...}`` (bench.py carries
   it as train.glow32).      python tools/glow_train_probe.py [steps]"""
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])

from torchflows.flows import Flow  # noqa: E402
from torchflows.bijections.finite.multiscale.architectures import AffineGlow  # noqa: E402
from torchflows_amd import native  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 1024
torch.manual_seed(0)
x = torch.randn(B, 3, 32, 32)
base = Flow(AffineGlow((3, 32, 32)))


def fit_ms(env, n):
    for k, v in env.items():
        os.environ[k] = v
    flow = copy.deepcopy(base).to(dev)
    flow.fit(x, n_epochs=3, batch_size=B)               # lazy state (kernels, index maps, the allocator), ActNorm statistics
    torch.cuda.synchronize()
    before = native.calls
    t0 = time.perf_counter()
    flow.fit(x, n_epochs=n, batch_size=B, keep_best_weights=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = dict(flow._fit_stats)
    done = st.get("eager_steps", 0) + st.get("graph_replays", 0)
    ok = all(bool(torch.isfinite(p).all()) for p in flow.parameters())
    for k in env:
        os.environ.pop(k, None)
    return {"ms_per_step": 1e3 * dt / max(done, 1), "steps": done, "fit_stats": st, "finite": ok,
            "libtfk_launches_per_step": (native.calls - before) / max(st.get("eager_steps", 0), 1) if not st.get("graph_replays") else None}


out = {"model": "AffineGlow((3, 32, 32))", "batch": B,
       "library_route": fit_ms({"TORCHFLOWS_AMD_DEBUG": "convnet_train=0", "TORCHFLOWS_AMD_GRAPH": "0"}, max(steps // 4, 3)),
       "eager": fit_ms({"TORCHFLOWS_AMD_GRAPH": "0"}, steps),
       "hipgraph": fit_ms({"TORCHFLOWS_AMD_GRAPH": "1"}, 4 * steps)}
print("GLOW_TRAIN_JSON " + json.dumps(out), flush=True)
