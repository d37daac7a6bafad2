"""Helpers shared by the tests and bench.py's parity leg: rebuild a model whose seed-reproducible weights are
pinned by a hash in a golden fixture (tests/golden/make_golden.py: gen_glow32)."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def state_hash(tensors) -> str:
    h = hashlib.sha256()
    for k, v in sorted(tensors, key=lambda kv: kv[0]):
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def load_glow32():
    """(flow, fixture): AffineGlow((3, 32, 32)) of config 5 in the state the reference was in when it produced
    tests/golden/flow_glow_3x32x32.npz -- constructed from seed 0 (weights must hash to the reference's), the
    data-dependent tensors (ActNorm values, BatchNorm statistics) loaded from the fixture, eval mode, on the host."""
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    fx = np.load(os.path.join(GOLDEN, "flow_glow_3x32x32.npz"), allow_pickle=False)
    torch.manual_seed(0)
    flow = tfa.Flow(AffineGlow((3, 32, 32)))
    assert sum(p.numel() for p in flow.parameters()) == int(fx["n_params"])
    stored = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")}
    sd = flow.state_dict()
    fixed = [(k, v) for k, v in sd.items() if k not in stored and k.split(".")[-1] != "device_buffer"]
    assert sum(v.numel() for _, v in fixed) == int(fx["seed_state_entries"])
    assert state_hash(fixed) == str(fx["seed_state_sha256"]), "seed-0 weights differ from the reference's"
    missing = [k for k in stored if k not in sd]
    assert not missing, missing
    sd.update(stored)
    flow.load_state_dict(sd)
    return flow.eval(), fx
