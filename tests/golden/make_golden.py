"""Generate the golden fixtures in tests/golden/ by RUNNING the real reference.

Run in the build container only (the reference is mounted read-only at
/root/reference and cannot travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_golden.py

Every array written is data (inputs, parameters, outputs of the reference);
no reference source is stored.  The fixtures pin oracle/ (tests/test_oracle_golden.py)
and, through it and directly, the HIP path (tests/test_gpu_*.py).
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")

from torchflows.flows import Flow  # noqa: E402  (reference)
from torchflows.bijections.finite.autoregressive.architectures import (  # noqa: E402
    RealNVP, CouplingRQNSF, NICE)
from torchflows.bijections.finite.autoregressive.conditioning.coupling_masks import HalfSplit  # noqa: E402
from torchflows.bijections.finite.autoregressive.layers import (  # noqa: E402
    ActNorm, ElementwiseAffine, AffineCoupling, RQSCoupling)
from torchflows.bijections.finite.autoregressive.transformers.linear.affine import Affine  # noqa: E402
from torchflows.bijections.finite.autoregressive.transformers.spline.rational_quadratic import (  # noqa: E402
    RationalQuadratic)
from torchflows.bijections.finite.matrix.permutation import ReversePermutationMatrix  # noqa: E402
from torchflows.base_distributions.gaussian import DiagonalGaussian  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def np32(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ---------------------------------------------------------------- F1 affine
def gen_affine():
    out = {}
    torch.manual_seed(0)
    for T in (2, 32, 128):
        tr = Affine((T,))
        x = torch.randn(64, T) * 2
        h = torch.randn(64, T, 2)
        # stress rows: large |u| (alpha tiny / huge), zero params (identity)
        h[0] = 0.0
        h[1, :, 0] = 30.0
        h[2, :, 0] = -30.0
        z, ld = tr.forward(x, h)
        xi, ldi = tr.inverse(x, h)
        out.update({f"T{T}_x": np32(x), f"T{T}_h": np32(h), f"T{T}_z": np32(z),
                    f"T{T}_ld": np32(ld), f"T{T}_xinv": np32(xi), f"T{T}_ldinv": np32(ldi)})
    save("affine.npz", **out)


# ---------------------------------------------------------------- F2 spline
def rqs_case(T, boundary, n_bins, x, h):
    tr = RationalQuadratic((T,), boundary=boundary, n_bins=n_bins)
    z, ld = tr.forward(x, h)
    xi, ldi = tr.inverse(x, h)
    # per-element log-dets and bin indices through a 1-element event
    tr1 = RationalQuadratic((1,), boundary=boundary, n_bins=n_bins)
    P = 3 * n_bins - 1
    _, ld_el = tr1.forward(x.reshape(-1, 1), h.reshape(-1, 1, P))
    _, ldi_el = tr1.inverse(x.reshape(-1, 1), h.reshape(-1, 1, P))
    hf = h.reshape(-1, P)
    bin_x, _ = tr.compute_bins(hf[:, :n_bins], -boundary, boundary)
    bin_y, _ = tr.compute_bins(hf[:, :n_bins] + hf[:, n_bins:2 * n_bins] / 1000, -boundary, boundary)
    xf = x.reshape(-1, 1).contiguous()
    inside = ((xf > -boundary) & (xf < boundary)).reshape(-1)
    kf = torch.searchsorted(bin_x, xf).reshape(-1) - 1
    ki = torch.searchsorted(bin_y, xf).reshape(-1) - 1
    kf = torch.where(inside, kf, torch.full_like(kf, -1))
    ki = torch.where(inside, ki, torch.full_like(ki, -1))
    return dict(x=np32(x), h=np32(h), z=np32(z), ld=np32(ld), xinv=np32(xi), ldinv=np32(ldi),
                ld_el=np32(ld_el).reshape(x.shape), ldinv_el=np32(ldi_el).reshape(x.shape),
                k=np32(kf).astype(np.int32).reshape(x.shape),
                kinv=np32(ki).astype(np.int32).reshape(x.shape),
                bin_x=np32(bin_x).reshape(*x.shape, n_bins + 1),
                bin_y=np32(bin_y).reshape(*x.shape, n_bins + 1))


def gen_rqs():
    out = {}
    torch.manual_seed(0)
    T = 8
    cases = []
    for boundary in (50.0, 5.0):
        for n_bins in (8, 4):
            P = 3 * n_bins - 1
            rows_x, rows_h = [], []
            for xs in (0.01, 1.0, 3.0, 30.0, 100.0):
                for hs in (0.3, 1.0, 5.0):
                    rows_x.append(torch.randn(6, T) * xs)
                    rows_h.append(torch.randn(6, T, P) * hs)
            x = torch.cat(rows_x)
            h = torch.cat(rows_h)
            # probes: exact interior knots (-> left bin), exact boundaries (-> identity),
            # just inside / outside, zeros and parameters = 0 (near-identity spline)
            tr = RationalQuadratic((T,), boundary=boundary, n_bins=n_bins)
            hp = torch.randn(4, T, P)
            bx, _ = tr.compute_bins(hp[..., :n_bins], -boundary, boundary)
            by, _ = tr.compute_bins(hp[..., :n_bins] + hp[..., n_bins:2 * n_bins] / 1000,
                                    -boundary, boundary)
            xp = torch.zeros(4, T)
            for t in range(T):
                xp[0, t] = bx[0, t, 1 + t % (n_bins - 1)]      # interior x-knot
                xp[1, t] = by[1, t, 1 + t % (n_bins - 1)]      # interior y-knot
            xp[2] = torch.tensor([boundary, -boundary, boundary * 1.2, -boundary * 1.2,
                                  np.nextafter(np.float32(boundary), np.float32(0)),
                                  np.nextafter(np.float32(-boundary), np.float32(0)),
                                  0.0, 1e-30])[:T]
            xp[3] = torch.linspace(-boundary, boundary, T)
            hz = torch.zeros(2, T, P)
            xz = torch.randn(2, T) * boundary / 3
            x = torch.cat([x, xp, xz])
            h = torch.cat([h, hp, hz])
            tag = f"B{int(boundary)}_K{n_bins}"
            for k, v in rqs_case(T, boundary, n_bins, x, h).items():
                out[f"{tag}_{k}"] = v
            cases.append(tag)
    out["cases"] = np.array(cases)
    save("rqs.npz", **out)


# ---------------------------------------------------------------- F3 masks
def gen_masks():
    out = {}
    shapes = [(2,), (3,), (64,), (256,), (7, 11), (3, 5, 2), (1, 1), (5,)]
    for es in shapes:
        c = HalfSplit(es)
        tag = "x".join(map(str, es))
        out[f"src_{tag}"] = np32(c.source_mask).astype(np.uint8)
        out[f"tgt_{tag}"] = np32(c.target_mask).astype(np.uint8)
        out[f"S_{tag}"] = np.int64(c.source_event_size)
        out[f"T_{tag}"] = np.int64(c.target_event_size)
        p = ReversePermutationMatrix(es)
        out[f"pfwd_{tag}"] = np32(p.forward_permutation).astype(np.int64)
        out[f"pinv_{tag}"] = np32(p.inverse_permutation).astype(np.int64)
    out["shapes"] = np.array(["x".join(map(str, s)) for s in shapes])
    save("masks.npz", **out)


# ---------------------------------------------------------------- F4/F7 flows
def layer_trace(flow, x, context=None):
    """Per-layer (z_i, ld_i) exactly as BijectiveComposition.forward iterates."""
    zs, lds = [], []
    cur = x
    for layer in flow.bijection.layers:
        cur, ld = layer.forward(cur, context=context)
        zs.append(np32(cur).reshape(x.shape[0], -1))
        lds.append(np32(ld))
    return np.stack(zs), np.stack(lds)


def flow_fixture(name, ctor, event_shape, n_rows, ctor_kwargs=None, context_shape=None,
                 trace_rows=8, seed=0):
    ctor_kwargs = dict(ctor_kwargs or {})
    if context_shape is not None:
        ctor_kwargs["context_shape"] = context_shape
    torch.manual_seed(seed)
    flow = Flow(ctor(event_shape, **ctor_kwargs))
    es = flow.bijection.event_shape
    D = int(np.prod(es))
    out = {"event_shape": np.array(es, dtype=np.int64), "n_bijection_layers": np.int64(len(flow.bijection.layers)),
           "layer_types": np.array([type(l).__name__ for l in flow.bijection.layers])}
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(n_rows, *es, generator=g)
    x[: n_rows // 4] *= 3.0
    z_in = torch.randn(n_rows, *es, generator=g)
    ctx = ctx_init = None
    if context_shape is not None:
        ctx = torch.randn(n_rows, *context_shape, generator=g)
        ctx_init = torch.randn(4096, *context_shape, generator=g)
        out["context"] = np32(ctx)
    out["x"] = np32(x)
    out["z_in"] = np32(z_in)
    x_init = torch.randn(4096, *es, generator=g)
    for variant in ("fresh", "init"):
        if variant == "init":
            flow.train()
            with torch.no_grad():
                flow.log_prob(x_init, context=ctx_init)   # ActNorm data-dependent init
        flow.eval()
        sd = flow.state_dict()
        for k, v in sd.items():
            out[f"sd_{variant}/{k}"] = np32(v)
        with torch.no_grad():
            z, ld = flow.bijection.forward(x, context=ctx)
            lp = flow.log_prob(x, context=ctx)
            xr, ldr = flow.bijection.inverse(z_in, context=ctx)
            base_lp = flow.base_log_prob(z_in)
            tz, tl = layer_trace(flow, x[:trace_rows], None if ctx is None else ctx[:trace_rows])
            # fp64 re-evaluation of the same weights: the fp32 noise floor
            f64 = Flow(ctor(event_shape, **ctor_kwargs)).double()
            f64.load_state_dict({k: v.double() for k, v in sd.items()})
            f64.eval()
            c64 = None if ctx is None else ctx.double()
            z64, ld64 = f64.bijection.forward(x.double(), context=c64)
            lp64 = f64.log_prob(x.double(), context=c64)
            xr64, ldr64 = f64.bijection.inverse(z_in.double(), context=c64)
        out.update({
            f"{variant}/z": np32(z), f"{variant}/log_det": np32(ld), f"{variant}/log_prob": np32(lp),
            f"{variant}/x_inv": np32(xr), f"{variant}/log_det_inv": np32(ldr),
            f"{variant}/sample_log_prob": np32(base_lp + ldr),
            f"{variant}/trace_z": tz, f"{variant}/trace_ld": tl,
            f"{variant}/z64": np32(z64), f"{variant}/log_det64": np32(ld64),
            f"{variant}/log_prob64": np32(lp64), f"{variant}/x_inv64": np32(xr64),
            f"{variant}/log_det_inv64": np32(ldr64),
        })
    save(name, **out)


def gen_flows():
    flow_fixture("flow_realnvp3.npz", RealNVP, 3, 64)                       # config 1 (n_layers=2)
    flow_fixture("flow_realnvp64.npz", RealNVP, 64, 64, dict(n_layers=8))   # config 2
    flow_fixture("flow_nsf64.npz", CouplingRQNSF, 64, 64, dict(n_layers=8)) # config 3
    flow_fixture("flow_realnvp256.npz", RealNVP, 256, 32, dict(n_layers=8), trace_rows=4)  # config 4
    flow_fixture("flow_nice7.npz", NICE, 7, 32)
    flow_fixture("flow_realnvp_7x11.npz", RealNVP, (7, 11), 16, trace_rows=4)   # test_cuda.py:14 shape
    flow_fixture("flow_realnvp5_ctx3.npz", RealNVP, 5, 32, context_shape=(3,))
    flow_fixture("flow_nsf6_ctx2.npz", CouplingRQNSF, 6, 960, context_shape=(2,))     # (round 4: >= 5 625 elements per
                                                                                      #  spline sibling: pass-rate slack <= 0.07)
    flow_fixture("flow_nsf_3x5x2.npz", CouplingRQNSF, (3, 5, 2), 192, trace_rows=4)


# ---------------------------------------------------------------- F5 gaussian
def gen_gauss():
    torch.manual_seed(0)
    out = {}
    for D in (3, 64):
        loc = torch.randn(D)
        scale = torch.rand(D) + 0.5
        d = DiagonalGaussian(loc, scale)
        v = torch.randn(50, D) * 2
        out[f"D{D}_loc"] = np32(loc)
        out[f"D{D}_log_scale"] = np32(d.log_scale)
        out[f"D{D}_value"] = np32(v)
        out[f"D{D}_log_prob"] = np32(d.log_prob(v))
    # closed-form checks of the reference's own test (test/test_base_distributions.py:8-33)
    d = DiagonalGaussian(torch.zeros(2), torch.ones(2))
    v = torch.tensor([[0.0, 0.0], [1.0, -2.0]])
    out["std2_value"] = np32(v)
    out["std2_log_prob"] = np32(d.log_prob(v))
    save("gauss.npz", **out)


# ---------------------------------------------------------------- layers
def gen_layers():
    torch.manual_seed(0)
    out = {}
    # ActNorm data-dependent initialisation (layers.py:58-68)
    for tag, n in (("n100", 100), ("n1", 1)):
        a = ActNorm((7,))
        a.train()
        x = torch.randn(n, 7) * 3 + 1
        with torch.no_grad():
            z, ld = a.forward(x)
        out[f"actnorm_{tag}_x"] = np32(x)
        out[f"actnorm_{tag}_value"] = np32(a.value)
        out[f"actnorm_{tag}_z"] = np32(z)
        out[f"actnorm_{tag}_ld"] = np32(ld)
    # single layers with batch shape (5, 2, 3) and event shape (3, 5, 2)
    for cls, tag in ((ElementwiseAffine, "ea"), (AffineCoupling, "ac"), (RQSCoupling, "rc")):
        torch.manual_seed(1)
        layer = cls((3, 5, 2)).eval()
        x = torch.randn(5, 2, 3, 3, 5, 2)
        with torch.no_grad():
            z, ld = layer.forward(x)
            xi, ldi = layer.inverse(x)
        for k, v in layer.state_dict().items():
            out[f"{tag}_sd/{k}"] = np32(v)
        out.update({f"{tag}_x": np32(x), f"{tag}_z": np32(z), f"{tag}_ld": np32(ld),
                    f"{tag}_xinv": np32(xi), f"{tag}_ldinv": np32(ldi)})
    save("layers.npz", **out)


# ---------------------------------------------------------------- F6 image path
def gen_image():
    from torchflows.bijections.finite.multiscale.coupling import Checkerboard, ChannelWiseHalfSplit
    from torchflows.bijections.finite.multiscale.base import (
        Squeeze, CheckerboardCoupling, ChannelWiseCoupling, Invertible1x1ConvolutionalCoupling)
    from torchflows.bijections.finite.multiscale.architectures import AffineGlow
    from torchflows.bijections.finite.autoregressive.transformers.linear.convolution import (
        Invertible1x1ConvolutionTransformer)
    from torchflows.bijections.finite.autoregressive.transformers.linear.matrix import LUTransformer
    out = {}
    shapes = [(3, 32, 32), (12, 16, 16), (6, 16, 16), (24, 8, 8), (12, 8, 8), (1, 4, 4), (3, 8, 8), (2, 2, 6)]
    for es in shapes:
        tag = "x".join(map(str, es))
        for inv in (False, True):
            c = Checkerboard(es, invert=inv)
            out[f"ckb{int(inv)}_src_{tag}"] = np32(c.source_mask).astype(np.uint8)
            out[f"ckb{int(inv)}_shapes_{tag}"] = np.array([*c.constant_shape, *c.target_shape], dtype=np.int64)
            if es[0] > 1:
                w = ChannelWiseHalfSplit(es, invert=inv)
                out[f"chw{int(inv)}_src_{tag}"] = np32(w.source_mask).astype(np.uint8)
                out[f"chw{int(inv)}_shapes_{tag}"] = np.array([*w.constant_shape, *w.target_shape], dtype=np.int64)
        sq = Squeeze(es)
        idx = torch.arange(int(np.prod(es)), dtype=torch.float32).view(1, *es)
        out[f"squeeze_fwd_{tag}"] = np32(sq.forward(idx)[0]).reshape(-1).astype(np.int64)
    out["shapes"] = np.array(["x".join(map(str, s)) for s in shapes])
    save("image_masks.npz", **out)

    out = {}
    torch.manual_seed(0)
    # LU / 1x1 convolution transformers with explicit parameters
    for n in (1, 2, 3, 6, 12):
        lu = LUTransformer((n,))
        x = torch.randn(10, n)
        h = torch.randn(10, *lu.parameter_shape)
        y, ld = lu.forward(x, h)
        xi, ldi = lu.inverse(x, h)
        out.update({f"lu{n}_x": np32(x), f"lu{n}_h": np32(h), f"lu{n}_y": np32(y), f"lu{n}_ld": np32(ld),
                    f"lu{n}_xinv": np32(xi), f"lu{n}_ldinv": np32(ldi)})
    for n, hw in ((3, (4, 4)), (6, (8, 8))):
        tr = Invertible1x1ConvolutionTransformer((n, *hw))
        x = torch.randn(5, n, *hw)
        h = torch.randn(5, *tr.parameter_shape)
        y, ld = tr.forward(x, h)
        xi, ldi = tr.inverse(x, h)
        out.update({f"conv{n}_x": np32(x), f"conv{n}_h": np32(h), f"conv{n}_y": np32(y), f"conv{n}_ld": np32(ld),
                    f"conv{n}_xinv": np32(xi), f"conv{n}_ldinv": np32(ldi)})
    # single convolutional coupling layers, eval mode (BatchNorm running statistics)
    from torchflows.bijections.finite.autoregressive.transformers.linear.affine import Affine as RefAffine
    for tag, ctor, es in (("ckb", lambda: CheckerboardCoupling((3, 8, 8), RefAffine), (3, 8, 8)),
                          ("ckb_alt", lambda: CheckerboardCoupling((3, 8, 8), RefAffine, alternate=True), (3, 8, 8)),
                          ("chw", lambda: ChannelWiseCoupling((4, 4, 4), RefAffine), (4, 4, 4)),
                          ("chw_alt", lambda: ChannelWiseCoupling((4, 4, 4), RefAffine, alternate=True), (4, 4, 4)),
                          ("c1x1", lambda: Invertible1x1ConvolutionalCoupling((4, 4, 4)), (4, 4, 4))):
        torch.manual_seed(3)
        layer = ctor().eval()
        x = torch.randn(6, *es)
        with torch.no_grad():
            z, ld = layer.forward(x)
            xi, ldi = layer.inverse(x)
        for k, v in layer.state_dict().items():
            out[f"{tag}_sd/{k}"] = np32(v)
        out.update({f"{tag}_x": np32(x), f"{tag}_z": np32(z), f"{tag}_ld": np32(ld),
                    f"{tag}_xinv": np32(xi), f"{tag}_ldinv": np32(ldi)})
    save("image_layers.npz", **out)

    # a whole (small) Glow: two blocks on (3, 8, 8)
    torch.manual_seed(0)
    flow = Flow(AffineGlow((3, 8, 8), n_layers=2))
    out = {"n_params": np.int64(sum(p.numel() for p in flow.parameters()))}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 3, 8, 8, generator=g)
    z_in = torch.randn(8, 3, 8, 8, generator=g)
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(64, 3, 8, 8, generator=g))     # ActNorm init + BatchNorm statistics
    flow.eval()
    with torch.no_grad():
        z, ld = flow.bijection.forward(x)
        lp = flow.log_prob(x)
        xr, ldr = flow.bijection.inverse(z_in)
    for k, v in flow.state_dict().items():
        out[f"sd/{k}"] = np32(v)
    out.update({"x": np32(x), "z_in": np32(z_in), "z": np32(z), "log_det": np32(ld), "log_prob": np32(lp),
                "x_inv": np32(xr), "log_det_inv": np32(ldr)})
    save("flow_glow_3x8x8.npz", **out)


def state_hash(tensors):
    """sha256 over the fp32 / int64 bytes of a list of (name, tensor), in name order."""
    import hashlib
    h = hashlib.sha256()
    for k, v in sorted(tensors, key=lambda kv: kv[0]):
        h.update(k.encode())
        h.update(np.ascontiguousarray(np32(v)).tobytes())
    return h.hexdigest()


def glow32_inputs(seed=5, n=64):
    """The fixture's inputs, NOT stored: numpy's frozen legacy generator reproduces them bit for bit anywhere
    (tests/golden_util.py regenerates them the same way and checks their sha256).  64 rows: 32 standard-normal, 16 scaled
    x 4 (activations at the conditioner's sigmoid bound, |z| ~ 1e2) and 16 scaled x 0.01; the latents of the inverse pass
    likewise (48 standard, 8 x 2, 8 x 0.01)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((n, 3, 32, 32)).astype(np.float32)
    x[n // 2: 3 * n // 4] *= 4.0
    x[3 * n // 4:] *= 0.01
    z_in = rs.standard_normal((n, 3, 32, 32)).astype(np.float32)
    z_in[3 * n // 4: 7 * n // 8] *= 2.0
    z_in[7 * n // 8:] *= 0.01
    return x, z_in


def gen_glow32():
    """Config 5 AS CONFIGURED: AffineGlow((3, 32, 32)) (auto n_layers = 3, 3.2 M parameters), seed 0,
    data-initialised on 64 rows.  The seed-reproducible tensors (every weight: 3 143 560 of 3 205 817 entries)
    are NOT stored -- the build constructs the same model from the same seed and the fixture pins their sha256;
    the tensors the train-mode pass changed (ActNorm values, BatchNorm statistics: 62 257 entries) are stored.
    Round 4: 64 evaluation rows (8 before) with stress rows, inputs regenerated from a seed instead of stored."""
    import hashlib
    from torchflows.bijections.finite.multiscale.architectures import AffineGlow
    torch.manual_seed(0)
    flow = Flow(AffineGlow((3, 32, 32)))
    sd0 = {k: v.clone() for k, v in flow.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    torch.randn(8, 3, 32, 32, generator=g), torch.randn(8, 3, 32, 32, generator=g)   # (the draws of the round-1..3 fixture:
    flow.train()                                                                      #  the init batch stays the same one)
    with torch.no_grad():
        flow.log_prob(torch.randn(64, 3, 32, 32, generator=g))
    flow.eval()
    sd1 = flow.state_dict()
    changed = [k for k in sd1 if not torch.equal(sd0[k], sd1[k])]
    fixed = [(k, v) for k, v in sd1.items() if k not in changed and k.split(".")[-1] != "device_buffer"]
    xn, zn = glow32_inputs()
    x, z_in = torch.from_numpy(xn), torch.from_numpy(zn)
    out = {"n_params": np.int64(sum(p.numel() for p in flow.parameters())),
           "seed_state_sha256": np.array(state_hash(fixed)),
           "seed_state_entries": np.int64(sum(v.numel() for _, v in fixed)),
           "input_seed": np.int64(5), "input_rows": np.int64(xn.shape[0]),
           "x_sha256": np.array(hashlib.sha256(xn.tobytes()).hexdigest()),
           "z_in_sha256": np.array(hashlib.sha256(zn.tobytes()).hexdigest())}
    for k in changed:
        out[f"sd/{k}"] = np32(sd1[k])
    with torch.no_grad():
        z, ld = flow.bijection.forward(x)
        lp = flow.log_prob(x)
        xr, ldr = flow.bijection.inverse(z_in)
        f64 = Flow(AffineGlow((3, 32, 32))).double()
        f64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd1.items()})
        f64.eval()
        lp64 = f64.log_prob(x.double())
        z64, ld64 = f64.bijection.forward(x.double())
        xr64, ldr64 = f64.bijection.inverse(z_in.double())
    # fp64 outputs: kept as fp32 for the first 8 rows only (the emulator's pin); for all rows the reference's own
    # fp32-vs-fp64 distances ride as scalars (the floor the tolerances are read against)
    relf = lambda a, b: float((a.double() - b).abs().div(b.abs().clamp(min=1.0)).max())
    out.update({"z": np32(z), "log_det": np32(ld), "log_prob": np32(lp),
                "x_inv": np32(xr), "log_det_inv": np32(ldr), "log_prob64": np32(lp64), "log_det64": np32(ld64),
                "log_det_inv64": np32(ldr64), "z64": np32(z64[:8]).astype(np.float32),
                "floor_z": np.float64(relf(z, z64)), "floor_x_inv": np.float64(relf(xr, xr64))})
    save("flow_glow_3x32x32.npz", **out)


def gen_grads_glow():
    """Reverse mode of the image path from the reference's autograd: (i) the 1x1-convolution transformer alone
    (convolution.py:33-64), both directions, fp32 and fp64; (ii) d sum(log_prob) / d (x, parameters) of the whole
    AffineGlow((3, 8, 8), n_layers=2) of flow_glow_3x8x8.npz (eval mode: BatchNorm on its running statistics)."""
    from torchflows.bijections.finite.multiscale.architectures import AffineGlow
    from torchflows.bijections.finite.autoregressive.transformers.linear.convolution import (
        Invertible1x1ConvolutionTransformer)
    out = {}
    torch.manual_seed(11)
    for n, hw in ((3, (4, 4)), (6, (8, 8)), (12, (2, 6))):
        tr = Invertible1x1ConvolutionTransformer((n, *hw))
        x = torch.randn(5, n, *hw)
        h = torch.randn(5, *tr.parameter_shape)
        gz, gld = torch.randn(5, n, *hw), torch.randn(5)
        out.update({f"conv{n}_x": np32(x), f"conv{n}_h": np32(h), f"conv{n}_gz": np32(gz), f"conv{n}_gld": np32(gld)})
        for inverse in (False, True):
            for dt, tag in ((torch.float32, ""), (torch.float64, "64")):
                gx, gh = _tr_grads(tr, x, h, gz, gld, inverse, dt)
                d = "inv" if inverse else "fwd"
                out[f"conv{n}_{d}_gx{tag}"] = gx
                out[f"conv{n}_{d}_gh{tag}"] = gh
    fx = np.load(os.path.join(OUT, "flow_glow_3x8x8.npz"))
    for dt, tag in ((torch.float32, ""), (torch.float64, "64")):
        torch.manual_seed(0)
        flow = Flow(AffineGlow((3, 8, 8), n_layers=2))
        flow.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")})
        flow = flow.to(dt).eval()
        x = torch.from_numpy(fx["x"]).to(dt).requires_grad_(True)
        named = [(k, p) for k, p in flow.named_parameters() if p.requires_grad and p.numel()]
        grads = torch.autograd.grad(flow.log_prob(x).sum(), [x] + [p for _, p in named], allow_unused=True)
        out[f"glow_gx{tag}"] = np32(grads[0]).astype(np.float32)
        for (k, p), g in zip(named, grads[1:]):
            out[f"glow_g{tag}/{k}"] = (np32(g) if g is not None else np.zeros(tuple(p.shape))).astype(np.float32)
    save("grads_glow_3x8x8.npz", **out)


# ---------------------------------------------------------------- F8 gradients
def _tr_grads(tr, x, h, gz, gld, inverse, dtype):
    x = x.to(dtype).clone().requires_grad_(True)
    h = h.to(dtype).clone().requires_grad_(True)
    out, ld = (tr.inverse if inverse else tr.forward)(x, h)
    loss = (out * gz.to(dtype)).sum() + (ld * gld.to(dtype)).sum()
    gx, gh = torch.autograd.grad(loss, (x, h))
    return np32(gx), np32(gh)


def gen_grads():
    """Reverse-mode gradients of the reference (its autograd graph), transformer level and
    whole-flow level, fp32 and fp64 (the fp64 run is the noise floor)."""
    out = {}
    torch.manual_seed(7)
    for T in (2, 32):
        tr = Affine((T,))
        x = torch.randn(48, T) * 2
        h = torch.randn(48, T, 2)
        h[0] = 0.0
        h[1, :, 0] = 12.0
        h[2, :, 0] = -12.0
        gz, gld = torch.randn(48, T), torch.randn(48)
        out.update({f"affine_T{T}_x": np32(x), f"affine_T{T}_h": np32(h),
                    f"affine_T{T}_gz": np32(gz), f"affine_T{T}_gld": np32(gld)})
        for inverse in (False, True):
            for dt, tag in ((torch.float32, ""), (torch.float64, "64")):
                gx, gh = _tr_grads(tr, x, h, gz, gld, inverse, dt)
                d = "inv" if inverse else "fwd"
                out[f"affine_T{T}_{d}_gx{tag}"] = gx
                out[f"affine_T{T}_{d}_gh{tag}"] = gh
    rq = np.load(os.path.join(OUT, "rqs.npz"))
    cases = []
    for tag, boundary, n_bins in (("B50_K8", 50.0, 8), ("B5_K8", 5.0, 8), ("B5_K4", 5.0, 4)):
        x = torch.from_numpy(rq[f"{tag}_x"])
        h = torch.from_numpy(rq[f"{tag}_h"])
        T = x.shape[1]
        tr = RationalQuadratic((T,), boundary=boundary, n_bins=n_bins)
        gz, gld = torch.randn(*x.shape), torch.randn(x.shape[0])
        out[f"rqs_{tag}_gz"] = np32(gz)
        out[f"rqs_{tag}_gld"] = np32(gld)
        for inverse in (False, True):
            for dt, t2 in ((torch.float32, ""), (torch.float64, "64")):
                gx, gh = _tr_grads(tr, x, h, gz, gld, inverse, dt)
                d = "inv" if inverse else "fwd"
                out[f"rqs_{tag}_{d}_gx{t2}"] = gx
                out[f"rqs_{tag}_{d}_gh{t2}"] = gh
        cases.append(tag)
    out["rqs_cases"] = np.array(cases)
    save("grads.npz", **out)

    # whole flows: weights and x of the existing flow fixtures ("init" variant)
    flows = [("flow_realnvp3.npz", RealNVP, 3, {}),
             ("flow_realnvp64.npz", RealNVP, 64, dict(n_layers=8)),
             ("flow_nsf64.npz", CouplingRQNSF, 64, dict(n_layers=8)),
             ("flow_nice7.npz", NICE, 7, {}),
             ("flow_realnvp_7x11.npz", RealNVP, (7, 11), {})]
    for fname, ctor, es, kw in flows:
        _flow_grads(fname, ctor, es, kw)


def _flow_grads(fname, ctor, es, kw):
    """d(sum_i w_i log_prob(x_i)) / d(x, parameters) of the reference's autograd, fp32 and fp64, on the
    weights and inputs of an existing flow fixture."""
    fx = np.load(os.path.join(OUT, fname))
    sd = {k[len("sd_init/"):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd_init/")}
    x = torch.from_numpy(fx["x"])
    g = torch.Generator().manual_seed(99)
    w = torch.rand(x.shape[0], generator=g) + 0.5          # per-row upstream gradient
    o = {"w": np32(w)}
    for dt, tag in ((torch.float32, ""), (torch.float64, "64")):
        flow = Flow(ctor(es, **kw)).to(dt)
        flow.load_state_dict({k: v.to(dt) for k, v in sd.items()})
        flow.eval()
        xx = x.to(dt).clone().requires_grad_(True)
        lp = flow.log_prob(xx)
        loss = (lp * w.to(dt)).sum()
        names = [n for n, p_ in flow.named_parameters() if p_.requires_grad]
        params = [p_ for n, p_ in flow.named_parameters() if p_.requires_grad]
        grads = torch.autograd.grad(loss, [xx] + params, allow_unused=True)
        o[f"gx{tag}"] = np32(grads[0])
        for n, gr in zip(names, grads[1:]):
            o[f"g{tag}/{n}"] = np32(gr if gr is not None else torch.zeros(()))
    # ActNorm values carry no gradient in the reference (requires_grad False, layers.py:49)
    o["trainable"] = np.array(names)
    save("grads_" + fname, **o)


def gen_grads_maf():
    """MADE-based flows: the parallel (density) direction differentiated by the reference."""
    from torchflows.bijections.finite.autoregressive.architectures import MAF, MaskedAutoregressiveRQNSF
    _flow_grads("flow_maf6.npz", MAF, 6, dict(n_layers=2))
    _flow_grads("flow_marqnsf5.npz", MaskedAutoregressiveRQNSF, 5, dict(n_layers=2))


def gen_grads_lrs():
    """Linear rational spline: the reference's autograd at transformer level (inputs of lrs.npz) and
    through CouplingLRS(16)."""
    from torchflows.bijections.finite.autoregressive.transformers.spline.linear_rational import LinearRational
    from torchflows.bijections.finite.autoregressive.architectures import CouplingLRS
    lr = np.load(os.path.join(OUT, "lrs.npz"))
    out, cases = {}, []
    torch.manual_seed(11)
    for tag, boundary, n_bins in (("B50_K8", 50.0, 8), ("B5_K8", 5.0, 8), ("B5_K4", 5.0, 4)):
        x = torch.from_numpy(lr[f"{tag}_x"])
        h = torch.from_numpy(lr[f"{tag}_h"])
        tr = LinearRational((x.shape[1],), boundary=boundary, n_bins=n_bins)
        gz, gld = torch.randn(*x.shape), torch.randn(x.shape[0])
        out[f"{tag}_gz"], out[f"{tag}_gld"] = np32(gz), np32(gld)
        for inverse in (False, True):
            for dt, t2 in ((torch.float32, ""), (torch.float64, "64")):
                gx, gh = _tr_grads(tr, x, h, gz, gld, inverse, dt)
                d = "inv" if inverse else "fwd"
                out[f"{tag}_{d}_gx{t2}"] = gx
                out[f"{tag}_{d}_gh{t2}"] = gh
        cases.append(tag)
    out["cases"] = np.array(cases)
    save("grads_lrs.npz", **out)
    _flow_grads("flow_lrs16.npz", CouplingLRS, 16, dict(n_layers=3))


# ---------------------------------------------------------------- F9 sibling layers
def gen_siblings():
    from torchflows.bijections.finite.autoregressive.layers import (
        ElementwiseScale, ElementwiseRQSpline, LinearAffineCoupling, LinearRQSCoupling, LinearShiftCoupling)
    from torchflows.bijections.finite.autoregressive.conditioning.transforms import ResidualFeedForward
    out = {}
    cases = [("ElementwiseScale", lambda: ElementwiseScale((6,))),
             ("ElementwiseRQSpline", lambda: ElementwiseRQSpline((6,))),
             ("LinearAffineCoupling", lambda: LinearAffineCoupling((6,))),
             ("LinearRQSCoupling", lambda: LinearRQSCoupling((6,))),
             ("LinearShiftCoupling", lambda: LinearShiftCoupling((6,))),
             ("AffineCoupling_ResidualFeedForward",
              lambda: AffineCoupling((6,), conditioner_transform_class=ResidualFeedForward)),
             ("RQSCoupling_ResidualFeedForward",
              lambda: RQSCoupling((8,), conditioner_transform_class=ResidualFeedForward,
                                  conditioner_kwargs=dict(n_layers=4, block_size=3)))]
    for name, make in cases:
        torch.manual_seed(0)
        layer = make()
        with torch.no_grad():
            for p_ in layer.parameters():
                p_.mul_(2.0)                      # leave the near-identity initialisation
        g = torch.Generator().manual_seed(3)
        D = int(np.prod(layer.event_shape))
        x = torch.randn(40, D, generator=g) * 1.5
        with torch.no_grad():
            z, ld = layer.forward(x)
            xi, ldi = layer.inverse(x)
        for k, v in layer.state_dict().items():
            out[f"{name}/sd/{k}"] = np32(v)
        out.update({f"{name}/x": np32(x), f"{name}/z": np32(z), f"{name}/ld": np32(ld),
                    f"{name}/xinv": np32(xi), f"{name}/ldinv": np32(ldi)})
    out["cases"] = np.array([c[0] for c in cases])
    save("siblings.npz", **out)


# ---------------------------------------------------------------- F10 linear rational spline
def gen_lrs():
    from torchflows.bijections.finite.autoregressive.transformers.spline.linear_rational import LinearRational
    from torchflows.bijections.finite.autoregressive.architectures import CouplingLRS
    out = {}
    torch.manual_seed(0)
    T = 8
    cases = []
    for boundary in (50.0, 5.0):
        for n_bins in (8, 4):
            P = 4 * n_bins
            rows_x, rows_h = [], []
            for xs in (0.01, 1.0, 3.0, 30.0, 100.0):
                for hs in (0.3, 1.0, 5.0):
                    rows_x.append(torch.randn(6, T) * xs)
                    rows_h.append(torch.randn(6, T, P) * hs)
            hz = torch.zeros(2, T, P)
            xz = torch.randn(2, T) * boundary / 3
            xe = torch.tensor([[boundary, -boundary, boundary * 1.2, -boundary * 1.2,
                                np.nextafter(np.float32(boundary), np.float32(0)),
                                np.nextafter(np.float32(-boundary), np.float32(0)), 0.0, 1e-30]])
            he = torch.randn(1, T, P)
            x = torch.cat(rows_x + [xz, xe])
            h = torch.cat(rows_h + [hz, he])
            tr = LinearRational((T,), boundary=boundary, n_bins=n_bins)
            tag = f"B{int(boundary)}_K{n_bins}"
            with torch.no_grad():
                z, ld = tr.forward(x, h)
                xi, ldi = tr.inverse(x, h)
                z64, ld64 = tr.forward(x.double(), h.double())
                xi64, ldi64 = tr.inverse(x.double(), h.double())
            out.update({f"{tag}_x": np32(x), f"{tag}_h": np32(h), f"{tag}_z": np32(z), f"{tag}_ld": np32(ld),
                        f"{tag}_xinv": np32(xi), f"{tag}_ldinv": np32(ldi), f"{tag}_z64": np32(z64),
                        f"{tag}_ld64": np32(ld64), f"{tag}_xinv64": np32(xi64), f"{tag}_ldinv64": np32(ldi64)})
            cases.append(tag)
    out["cases"] = np.array(cases)
    save("lrs.npz", **out)
    flow_fixture("flow_lrs16.npz", CouplingLRS, 16, 360, dict(n_layers=3))


# ---------------------------------------------------------------- F11 MADE-based flows (8f-4)
def gen_maf():
    from torchflows.bijections.finite.autoregressive.architectures import (
        MAF, IAF, MaskedAutoregressiveRQNSF, InverseAutoregressiveRQNSF, MaskedAutoregressiveLRS)
    flow_fixture("flow_maf6.npz", MAF, 6, 32, dict(n_layers=2))
    flow_fixture("flow_iaf6.npz", IAF, 6, 32, dict(n_layers=2))
    flow_fixture("flow_marqnsf5.npz", MaskedAutoregressiveRQNSF, 5, 1200, dict(n_layers=2))
    flow_fixture("flow_iarqnsf5.npz", InverseAutoregressiveRQNSF, 5, 1200, dict(n_layers=2))
    flow_fixture("flow_malrs5.npz", MaskedAutoregressiveLRS, 5, 1200, dict(n_layers=2))


if __name__ == "__main__":
    which = sys.argv[1:] or ["affine", "rqs", "masks", "gauss", "layers", "flows", "image"]
    for w in which:
        globals()[f"gen_{w}"]()
