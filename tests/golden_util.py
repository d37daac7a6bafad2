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


def glow32_inputs(seed: int = 5, n: int = 64):
    """The inputs of tests/golden/flow_glow_3x32x32.npz (make_golden.py: glow32_inputs, same code): numpy's frozen legacy
    generator, so they are regenerated here instead of stored -- 32 standard-normal rows, 16 scaled x 4, 16 scaled x 0.01;
    latents 48 standard, 8 x 2, 8 x 0.01."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((n, 3, 32, 32)).astype(np.float32)
    x[n // 2: 3 * n // 4] *= 4.0
    x[3 * n // 4:] *= 0.01
    z_in = rs.standard_normal((n, 3, 32, 32)).astype(np.float32)
    z_in[3 * n // 4: 7 * n // 8] *= 2.0
    z_in[7 * n // 8:] *= 0.01
    return x, z_in


def load_glow32():
    """(flow, fixture): AffineGlow((3, 32, 32)) of config 5 in the state the reference was in when it produced
    tests/golden/flow_glow_3x32x32.npz -- constructed from seed 0 (weights must hash to the reference's), the
    data-dependent tensors (ActNorm values, BatchNorm statistics) loaded from the fixture, eval mode, on the host."""
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    with np.load(os.path.join(GOLDEN, "flow_glow_3x32x32.npz"), allow_pickle=False) as npz:
        fx = {k: npz[k] for k in npz.files}
    fx["x"], fx["z_in"] = glow32_inputs(int(fx["input_seed"]), int(fx["input_rows"]))
    assert hashlib.sha256(fx["x"].tobytes()).hexdigest() == str(fx["x_sha256"]), "regenerated inputs differ"
    assert hashlib.sha256(fx["z_in"].tobytes()).hexdigest() == str(fx["z_in_sha256"]), "regenerated latents differ"
    torch.manual_seed(0)
    flow = tfa.Flow(AffineGlow((3, 32, 32)))
    assert sum(p.numel() for p in flow.parameters()) == int(fx["n_params"])
    stored = {k[3:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("sd/")}
    sd = flow.state_dict()
    fixed = [(k, v) for k, v in sd.items() if k not in stored and k.split(".")[-1] != "device_buffer"]
    assert sum(v.numel() for _, v in fixed) == int(fx["seed_state_entries"])
    assert state_hash(fixed) == str(fx["seed_state_sha256"]), "seed-0 weights differ from the reference's"
    missing = [k for k in stored if k not in sd]
    assert not missing, missing
    sd.update(stored)
    flow.load_state_dict(sd)
    return flow.eval(), fx
