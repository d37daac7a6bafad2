#!/usr/bin/env python3
"""The reference's only published timings are the tqdm rates its notebooks left behind (hardware
not stated).  This runs the SAME calls on this build, on cuda:0, each bounded to a few seconds, and prints one JSON object
``{"notebook_<name>": {...}}`` that bench.py carries under ``train`` next to the reference figure:

  fit_realnvp50        docs/notebooks/training_with_datasets.ipynb:42    Flow(RealNVP((50,))).fit(x[1000, 50])            59.25 epochs/s
  fit_realnvp50_val    ... :78   + x_val[200, 50], early_stopping=True, early_stopping_threshold=50                       43.23 epochs/s
  vi_realnvp11         training_with_variational_inference.ipynb:69      variational_fit(n_samples=1, early stopping 500) 202.55 steps/s
  fit_realnvp_mnist    image_modeling.ipynb:91   Flow(RealNVP((1, 28, 28))).fit(x[1000], x_val[200], early_stopping)       8.30 epochs/s
  fit_msrealnvp_mnist  image_modeling.ipynb:91   Flow(MultiscaleRealNVP((1, 28, 28))).fit(same)                            2.28 s / epoch

The data are synthetic of the notebooks' shapes (no MNIST here: standardised noise images); an epoch is one pass over the
1 000 training rows at the default batch size 1 024 = one AdamW step, as in the notebooks.  Every leg first runs a short
warm-up call on a copy of the model (lazy state: kernels' code objects, index maps, the allocator), then times ONE call with
the notebook's arguments cut to ``n_epochs`` that fit the leg's time bound; epochs/s = epochs run / wall time of that call,
the quantity tqdm prints."""
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])

from torchflows.flows import Flow  # noqa: E402   (the reference's import spelling, through the alias)
from torchflows.architectures import RealNVP, MultiscaleRealNVP  # noqa: E402

dev = torch.device(os.environ.get("NOTEBOOK_PROBE_DEVICE", "cuda:0"))      # ("cpu": dry run of the script itself)
BOUND_S = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def sync():
    if dev.type == "cuda":
        torch.cuda.synchronize()


def leg(make, call, n_epochs, warm_epochs, reference, unit="epochs/s"):
    torch.manual_seed(0)
    flow = make().to(dev)
    warm = copy.deepcopy(flow)
    call(warm, warm_epochs)                              # lazy state
    del warm
    # size the timed call: a short pilot gives the rate, the timed call runs min(n_epochs, what fits the bound)
    pilot = copy.deepcopy(flow)
    sync()
    t0 = time.perf_counter()
    call(pilot, warm_epochs)
    sync()
    per_epoch = (time.perf_counter() - t0) / warm_epochs
    del pilot
    n = int(max(warm_epochs, min(n_epochs, BOUND_S / max(per_epoch, 1e-6))))
    sync()
    t0 = time.perf_counter()
    call(flow, n)
    sync()
    dt = time.perf_counter() - t0
    stats = dict(getattr(flow, "_fit_stats", {}) or {})
    if "eager_steps" in stats:                         # (early stopping may end the call before n epochs; one step per epoch)
        n = int(stats.get("eager_steps", 0)) + int(stats.get("graph_replays", 0))
    with torch.no_grad():
        ok = bool(all(torch.isfinite(p).all() for p in flow.parameters()))
    return {"value": n / dt, "unit": unit, "epochs": n, "seconds": dt, "reference": reference,
            "reference_hardware": "not stated", "finite": ok, **({"fit_stats": stats} if stats else {})}


def main():
    out = {}
    torch.manual_seed(0)
    x50 = torch.randn(1000, 50) * 5 + 7
    v50 = torch.randn(200, 50) * 5 + 7
    out["notebook_fit_realnvp50"] = leg(
        lambda: Flow(RealNVP(event_shape=(50,))), lambda f, n: f.fit(x50, n_epochs=n), 500, 40,
        {"value": 59.25, "unit": "epochs/s", "source": "docs/notebooks/training_with_datasets.ipynb:42"})
    out["notebook_fit_realnvp50_val"] = leg(
        lambda: Flow(RealNVP(event_shape=(50,))),
        lambda f, n: f.fit(x50, x_val=v50, early_stopping=True, early_stopping_threshold=50, n_epochs=n), 3250, 40,
        {"value": 43.23, "unit": "epochs/s", "source": "docs/notebooks/training_with_datasets.ipynb:78"})

    def log_density(x, mean=5, std=2):
        return -0.5 * torch.sum((x - mean) ** 2 / std ** 2, dim=-1)

    out["notebook_vi_realnvp11"] = leg(
        lambda: Flow(RealNVP(event_shape=(11,))),
        lambda f, n: f.variational_fit(target_log_prob=log_density, n_epochs=n, n_samples=1, early_stopping=True,
                                       early_stopping_threshold=500), 1383, 40,
        {"value": 202.55, "unit": "steps/s", "source": "docs/notebooks/training_with_variational_inference.ipynb:69"},
        unit="steps/s")
    g = torch.Generator().manual_seed(1)
    img = torch.randn(1200, 1, 28, 28, generator=g)
    img = (img - img.mean()) / img.std()
    xt, xv = img[:1000], img[1000:]
    out["notebook_fit_realnvp_mnist"] = leg(
        lambda: Flow(RealNVP((1, 28, 28))), lambda f, n: f.fit(xt, x_val=xv, early_stopping=True, n_epochs=n), 500, 10,
        {"value": 8.30, "unit": "epochs/s", "source": "docs/notebooks/image_modeling.ipynb:91 (first model)"})
    out["notebook_fit_msrealnvp_mnist"] = leg(
        lambda: Flow(MultiscaleRealNVP((1, 28, 28))),
        lambda f, n: f.fit(xt, x_val=xv, early_stopping=True, n_epochs=n), 152, 3,
        {"value": 1.0 / 2.28, "unit": "epochs/s", "source": "docs/notebooks/image_modeling.ipynb:91 (second model: 2.28 s/it)"})
    print("NOTEBOOK_JSON " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
