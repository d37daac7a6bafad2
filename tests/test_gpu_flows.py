"""Flow-level parity on the GPU: ``Flow.log_prob`` / ``bijection.inverse`` of this package,
running on the HIP kernels, against (1) the reference's golden outputs and (2) the CPU
oracle on seeded inputs, then size-independent properties at the BASELINE.json sizes.

Tolerances: log_prob within 1e-5 relative of the reference's fp32 value for the BASELINE.json configurations in the
benchmark's state (data-initialised weights), RQ-spline config 3 included; for the stress variants and sibling flows,
whose reference fp32 values are themselves 4e-6 .. 1e-5 from the reference's fp64 values, within max(1e-5, 2 x that
floor) of the EXACT (fp64) value); RealNVP / NICE z and reconstructed x within 1e-5; RQ-spline z / x -- whose knots are
100 cumsum(softmax) - 50, 1 ulp = 3.8e-6 amplified by 1 / bin width -- within max(4e-5, 3 x floor), with the
elementwise |d| <= 1e-5 max(1, |ref|) pass rate printed beside the reference's own (hard part 1).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, state_dict_of, set_debug

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def pass_rate(a, b, tol=1e-5):
    """share of entries with |a - b| <= tol * max(1, |b|)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.mean(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))) if a.size else 1.0


def normwise(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available()
    import torchflows_amd as tfa
    from torchflows_amd import native
    native.lib()
    return tfa


def build_flow(pkg, arch, event_shape, n_layers, context_shape=None):
    ctor = getattr(pkg, arch)
    kw = dict(n_layers=n_layers)
    if context_shape is not None:
        kw["context_shape"] = context_shape
    return pkg.Flow(ctor(event_shape, **kw))


FLOWS = [
    ("flow_realnvp3.npz", "RealNVP", 2, None, False),
    ("flow_realnvp64.npz", "RealNVP", 8, None, False),
    ("flow_nsf64.npz", "CouplingRQNSF", 8, None, True),
    ("flow_realnvp256.npz", "RealNVP", 8, None, False),
    ("flow_nice7.npz", "NICE", 2, None, False),
    ("flow_realnvp_7x11.npz", "RealNVP", 2, None, False),
    ("flow_realnvp5_ctx3.npz", "RealNVP", 2, (3,), False),
    ("flow_nsf6_ctx2.npz", "CouplingRQNSF", 2, (2,), True),
    ("flow_nsf_3x5x2.npz", "CouplingRQNSF", 2, None, True),
    ("flow_lrs16.npz", "CouplingLRS", 3, None, True),
    ("flow_maf6.npz", "MAF", 2, None, False),
    ("flow_iaf6.npz", "IAF", 2, None, False),
    ("flow_marqnsf5.npz", "MaskedAutoregressiveRQNSF", 2, None, True),
    ("flow_iarqnsf5.npz", "InverseAutoregressiveRQNSF", 2, None, True),
    ("flow_malrs5.npz", "MaskedAutoregressiveLRS", 2, None, True),
]


@pytest.mark.parametrize("variant", ["fresh", "init"])
@pytest.mark.parametrize("name,arch,n_layers,ctx_shape,spline", FLOWS)
def test_flow_golden_on_hip(pkg, name, arch, n_layers, ctx_shape, spline, variant):
    from torchflows_amd import native
    fx = load_golden(name)
    es = tuple(int(v) for v in fx["event_shape"])
    flow = build_flow(pkg, arch, es, n_layers, ctx_shape)
    flow.load_state_dict({k: torch.from_numpy(v) for k, v in state_dict_of(fx, variant).items()})
    flow = flow.cuda().eval()
    ctx = torch.from_numpy(fx["context"]).cuda() if ctx_shape else None
    x = torch.from_numpy(fx["x"]).cuda()
    z_in = torch.from_numpy(fx["z_in"]).cuda()
    g = lambda k: fx[f"{variant}/{k}"]
    before = native.calls
    with torch.no_grad():
        z, lp = flow.forward_with_log_prob(x, context=ctx)
        xr, ldr = flow.bijection.inverse(z_in, context=ctx)
    # fused flow programs: >= 1 launch per call; layer by layer: one per layer
    assert native.calls - before >= 2, "the HIP kernels did not run"
    assert lp.shape == x.shape[:1] and z.shape == x.shape and xr.shape == x.shape

    tol = 4e-5 if spline else 1e-5
    floor_lp = rel(g("log_prob"), g("log_prob64"))
    floor_x = normwise(g("x_inv"), g("x_inv64"))
    floor_ld = rel(g("log_det_inv"), g("log_det_inv64"))
    e_lp = rel(lp.cpu().numpy(), g("log_prob"))
    e_z = normwise(z.cpu().numpy(), g("z"))
    e_x = normwise(xr.cpu().numpy(), g("x_inv"))
    e_ld = rel(ldr.cpu().numpy(), g("log_det_inv"))
    print(f"{name} {variant}: log_prob {e_lp:.2e} (floor {floor_lp:.2e}), z nw {e_z:.2e}, "
          f"x_inv nw {e_x:.2e} (floor {floor_x:.2e}), log_det_inv {e_ld:.2e} (floor {floor_ld:.2e}); "
          f"elementwise 1e-5 pass rate: z {pass_rate(z.cpu().numpy(), g('z')):.4f} "
          f"(reference fp32 vs fp64 {pass_rate(g('z'), g('z64')):.4f}), x_inv {pass_rate(xr.cpu().numpy(), g('x_inv')):.4f} "
          f"(reference {pass_rate(g('x_inv'), g('x_inv64')):.4f}), log_prob {pass_rate(lp.cpu().numpy(), g('log_prob')):.4f} "
          f"(reference {pass_rate(g('log_prob'), g('log_prob64')):.4f})")
    # log_prob.  The BASELINE.json configurations in the benchmark's state (data-initialised weights: configs 1-4 =
    # flow_realnvp3 / realnvp64 / nsf64 / realnvp256 "init"): the stated 1e-5 against the reference's fp32 value,
    # splines included, no floor.  The stress variants ("fresh": randn elementwise layers, |log_prob| ~ 1e4,
    # activations ~ 1e2) and the sibling flows: there the reference's own fp32 value is 4e-6 .. 1e-5 away from its
    # fp64 value on the same rows (floor_lp; the oracle, a bit-faithful restatement, sits at 1.03e-5 on flow_nsf64
    # fresh), so the bar is stated against the reference's EXACT value: no further from it than
    # max(1e-5, 2 x the reference's own fp32 distance), and by the triangle inequality within 1e-5 + 3 x floor of the
    # fp32 value.
    e_lp64 = rel(lp.cpu().numpy(), g("log_prob64"))
    if e_lp >= 1e-5:
        import conftest
        conftest.PARITY_NOTES.append(f"{name[:-4]} {variant} {e_lp:.2e} / {floor_lp:.2e}")
    print(f"    log_prob vs the reference in fp64: {e_lp64:.2e}")
    # Elementwise accuracy where fp32 itself is the limit (spline knots = 100 * cumsum(softmax) - 50: one ulp of a logit
    # moves a knot by up to 1e-5): two independent fp32 evaluations differ from EACH OTHER by the sum of their distances
    # from the exact value, so the kernels are held to the exact value instead -- their elementwise 1e-5 pass rate
    # against the reference evaluated in fp64 must be within 0.05 of the reference's own fp32 pass rate against it
    # (measured on flow_nsf64 init: z 0.80 vs 0.80, x 0.75 vs 0.76; against the reference's fp32 VALUE every fp32
    # implementation sits lower -- 0.68 here, 0.66 for the bit-faithful per-layer kernels behind hipBLASLt GEMMs).
    pr_z, pr_z_ref = pass_rate(z.cpu().numpy(), g("z64")), pass_rate(g("z"), g("z64"))
    pr_x, pr_x_ref = pass_rate(xr.cpu().numpy(), g("x_inv64")), pass_rate(g("x_inv"), g("x_inv64"))
    print(f"    elementwise 1e-5 pass rate vs the reference in fp64: z {pr_z:.4f} (reference's fp32 {pr_z_ref:.4f}), "
          f"x_inv {pr_x:.4f} (reference's fp32 {pr_x_ref:.4f})")
    slack = 0.05 + 1.5 / np.sqrt(z.numel())           # (+ sampling noise; round 4: every spline sibling fixture holds >= 5 625
                                                      #  elements, so the slack is <= 0.07 -- it was 0.19 on 120 elements)
    assert pr_z >= pr_z_ref - slack and pr_x >= pr_x_ref - slack, (pr_z, pr_z_ref, pr_x, pr_x_ref)
    if variant == "init" and name in ("flow_realnvp3.npz", "flow_realnvp64.npz", "flow_nsf64.npz", "flow_realnvp256.npz"):
        assert e_lp < 1e-5, e_lp
    assert e_lp < 1e-5 or e_lp64 < max(1e-5, 2 * floor_lp), (e_lp, e_lp64, floor_lp)
    assert e_lp < 1e-5 + 3 * floor_lp
    assert e_z < max(tol, 3 * normwise(g("z"), g("z64")))
    assert e_x < max(tol, 3 * floor_x)
    # log-dets sum D/2 terms per layer that largely cancel: the bound is per 32 terms (as in
    # test_gpu_kernels), or 3x the reference's own fp32-vs-fp64 distance if that is larger
    D = int(np.prod(es))
    assert e_ld < max(tol * max(1.0, D / 64), 3 * floor_ld)
    # the caller's tensors are untouched
    assert torch.equal(x.cpu(), torch.from_numpy(fx["x"]))
    assert torch.equal(z_in.cpu(), torch.from_numpy(fx["z_in"]))


@pytest.mark.parametrize("arch,D,n_layers,N", [
    ("RealNVP", 64, 8, 4099), ("CouplingRQNSF", 64, 8, 2051), ("RealNVP", 256, 8, 1031),
    ("RealNVP", 3, 2, 1000), ("NICE", 10, 3, 777)])
def test_flow_vs_oracle_seeded(pkg, oracle, arch, D, n_layers, N):
    torch.manual_seed(7)
    flow = build_flow(pkg, arch, D, n_layers)
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(2048, D))          # data-dependent ActNorm init (ATen, host)
    flow.eval()
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, n_layers, sd)
    x = torch.randn(N, D)
    x[: N // 8] *= 4.0
    flow = flow.cuda()
    with torch.no_grad():
        z, lp = flow.forward_with_log_prob(x.cuda())
        xr, ld = flow.bijection.inverse(x.cuda())
    z_ref, lp_ref = ref.log_prob(x.numpy(), return_z=True)
    xr_ref, ld_ref = ref.inverse(x.numpy())
    tol = 4e-5 if arch == "CouplingRQNSF" else 1e-5
    e = dict(lp=rel(lp.cpu().numpy(), lp_ref), z=normwise(z.cpu().numpy(), z_ref),
             x=normwise(xr.cpu().numpy(), xr_ref), ld=rel(ld.cpu().numpy(), ld_ref))
    print(arch, D, e, "elementwise 1e-5 pass rate: z", pass_rate(z.cpu().numpy(), z_ref), "x", pass_rate(xr.cpu().numpy(), xr_ref),
          "log_prob", pass_rate(lp.cpu().numpy(), lp_ref))
    assert e["lp"] < 1e-5, e                         # log_prob: 1e-5 for every flow, splines included
    assert max(e.values()) < tol, e


def test_batch_and_event_shapes(pkg):
    """Any batch rank, any event rank (reference test/constants.py grids)."""
    torch.manual_seed(0)
    for es in ((2,), (3,), (3, 5, 2)):
        for bs in ((1,), (2,), (5,), (5, 2, 3)):
            for ctor in (pkg.RealNVP, pkg.CouplingRQNSF):
                flow = pkg.Flow(ctor(es)).eval()
                x = torch.randn(*bs, *es)
                with torch.no_grad():
                    lp_host = flow.log_prob(x)                          # ATen composite, host
                    lp_dev = flow.cuda().log_prob(x.cuda())             # HIP kernels
                    z, ld = flow.bijection.forward(x.cuda())
                    xr, ldr = flow.bijection.inverse(z)
                assert lp_dev.shape == bs and z.shape == x.shape
                assert rel(lp_dev.cpu().numpy(), lp_host.numpy()) < (1e-5 if ctor is pkg.RealNVP else 4e-5)
                assert torch.allclose(xr.cpu(), x, atol=1e-3)          # reference data_atol
                assert torch.allclose(ld, -ldr, atol=1e-3)             # reference log_det_atol


def test_non_contiguous_input_and_layer_standalone(pkg):
    from torchflows_amd.bijections.finite.autoregressive.layers import AffineCoupling, RQSCoupling, ActNorm
    torch.manual_seed(1)
    for cls in (AffineCoupling, RQSCoupling, ActNorm):
        layer = cls((6,)).eval()
        x = torch.randn(50, 12)[:, ::2]                     # strided view
        with torch.no_grad():
            z_h, ld_h = layer.forward(x)
            z_d, ld_d = layer.cuda().forward(x.cuda())
            x_d, ldi_d = layer.inverse(z_d)
        assert rel(z_d.cpu().numpy(), z_h.numpy()) < 2e-5
        assert rel(ld_d.cpu().numpy(), ld_h.numpy()) < 2e-5
        assert torch.allclose(x_d.cpu(), x, atol=1e-4)


def test_actnorm_train_mode_init_on_device(pkg, oracle):
    """A freshly constructed flow is in training mode: the first forward sets every ActNorm
    from the batch (reference layers.py:58-68).  Same statistics on the HIP path."""
    torch.manual_seed(3)
    flow_h = pkg.Flow(pkg.RealNVP(16, n_layers=2))
    flow_d = pkg.Flow(pkg.RealNVP(16, n_layers=2))
    flow_d.load_state_dict(flow_h.state_dict())
    x = torch.randn(1024, 16) * 2 + 1
    with torch.no_grad():
        lp_h = flow_h.log_prob(x)
        lp_d = flow_d.cuda().log_prob(x.cuda())
    assert rel(lp_d.cpu().numpy(), lp_h.numpy()) < 2e-5
    for (k, a), (_, b) in zip(flow_h.state_dict().items(), flow_d.state_dict().items()):
        assert rel(b.cpu().numpy(), a.numpy()) < 2e-5, k


def test_invert_swaps_direction_on_hip(pkg):
    from torchflows_amd.bijections.base import invert
    torch.manual_seed(0)
    b = pkg.RealNVP(8, n_layers=2).eval().cuda()
    x = torch.randn(33, 8, device="cuda")
    with torch.no_grad():
        z, ld = b.forward(x)
        invert(b)
        x2, ld2 = b.forward(z)          # now the inverse map
    assert torch.allclose(x2, x, atol=1e-4) and torch.allclose(ld2, -ld, atol=1e-4)


# ----------------------------------------------------------------- BASELINE.json sizes
@pytest.mark.parametrize("arch,D,N,chunk", [
    ("RealNVP", 64, 1 << 20, None),          # config 2
    ("CouplingRQNSF", 64, 1 << 20, 1 << 18), # config 3 (h is 2.9 GiB/layer at 2^20: chunked)
    ("RealNVP", 256, 1 << 19, None),         # config 4, one rank's shard
])
def test_full_size_properties(pkg, oracle, arch, D, N, chunk):
    """At full size the oracle is too slow to check every row; check (a) a 4096-row random
    subset against it, (b) round trip x -> z -> x, (c) ld_fwd = -ld_inv, (d) chunking
    invariance (rows are independent), (e) the fp64 sum of log_prob."""
    from torchflows_amd import native
    torch.manual_seed(0)
    flow = build_flow(pkg, arch, D, 8)
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(4096, D))
    flow.eval()
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, 8, sd)
    flow = flow.cuda()
    gen = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(N, D, device="cuda", generator=gen)
    step = chunk or N
    tol = 4e-5 if arch == "CouplingRQNSF" else 1e-5
    lp = torch.empty(N, device="cuda")
    with torch.no_grad():
        for lo in range(0, N, step):
            z, lp[lo:lo + step] = flow.forward_with_log_prob(x[lo:lo + step])
            if lo == 0:
                _, ld = flow.bijection.forward(x[:step])
                xr, ldr = flow.bijection.inverse(z)
                assert float((xr - x[:step]).abs().max()) < 1e-3
                assert float((ld + ldr).abs().max()) < 1e-3
                assert bool(torch.isfinite(z).all())
        # chunking invariance: rows do not interact
        lp_small = flow.log_prob(x[1000:1000 + 777])
    # (not bitwise: the conditioner GEMMs are rocBLAS/hipBLASLt, which picks kernels by M)
    assert torch.allclose(lp_small, lp[1000:1000 + 777], rtol=tol, atol=tol)
    assert bool(torch.isfinite(lp).all())
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(0))[:4096]
    lp_ref = ref.log_prob(x[idx.cuda()].cpu().numpy())
    e = rel(lp[idx.cuda()].cpu().numpy(), lp_ref)
    print(f"{arch} D={D} N={N}: log_prob vs oracle on 4096 rows: {e:.2e}, 1e-5 pass rate "
          f"{pass_rate(lp[idx.cuda()].cpu().numpy(), lp_ref):.4f}")
    assert e < 1e-5                                  # log_prob: 1e-5 for every config, config 3 included
    total = native.sum_f32(lp).item()
    assert abs(total - float(lp.double().sum().item())) < 1e-6 * abs(total)


@pytest.mark.parametrize("name", ["LinearAffineCoupling", "LinearRQSCoupling", "LinearShiftCoupling",
                                  "AffineCoupling_ResidualFeedForward", "RQSCoupling_ResidualFeedForward"])
def test_sibling_couplings_on_hip(name):
    """SURVEY 8(f)-4 siblings: the coupling kernels consume h whatever predicted it
    (ResidualFeedForward on PyTorch-ROCm); outputs vs the reference's (tests/golden/siblings.npz)."""
    from test_host_cpu import _sibling
    from torchflows_amd import native
    fx = load_golden("siblings.npz")
    torch.manual_seed(0)
    layer = _sibling(name)
    layer.load_state_dict({k[len(name) + 4:]: torch.tensor(fx[k]) for k in fx.files if k.startswith(name + "/sd/")})
    layer = layer.cuda()
    x = torch.tensor(fx[f"{name}/x"]).cuda()
    before = native.calls
    with torch.no_grad():
        z, ld = layer.forward(x)
        xi, ldi = layer.inverse(x)
    assert native.calls - before == 2
    tol = 4e-5 if "RQS" in name else 1e-5
    for mine, key in ((z, "z"), (ld, "ld"), (xi, "xinv"), (ldi, "ldinv")):
        ref = fx[f"{name}/{key}"]
        err = np.max(np.abs(mine.cpu().numpy() - ref) / np.maximum(1.0, np.abs(ref)))
        assert err < tol, (name, key, err)


@pytest.mark.parametrize("arch", ["MAF", "IAF"])
@pytest.mark.parametrize("D,n_hidden", [(6, None), (64, None), (64, 24), (256, None), (256, 40), (7, 3)])
def test_made_sequential_map_in_one_launch(pkg, oracle, arch, D, n_hidden):
    """MAF.inverse / IAF.forward: the reference's D conditioner passes run as ONE libtfk launch per layer
    (tfk_made_affine_sequential); parity with the oracle, which restates the D-pass loop."""
    from torchflows_amd import native
    torch.manual_seed(D + (n_hidden or 0))
    kw = dict(n_layers=2)
    if n_hidden is not None:
        kw["conditioner_kwargs"] = dict(n_hidden=n_hidden)
    flow = pkg.Flow(getattr(pkg, arch)(D, **kw))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(2048, D))                # ActNorm statistics
    flow.eval()
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, 2, sd)
    x = torch.randn(513, D)
    flow = flow.cuda()
    before = native.calls
    with torch.no_grad():
        z, ld = flow.bijection.forward(x.cuda())
        xr, ldi = flow.bijection.inverse(x.cuda())
    launches = native.calls - before
    if not (D == 256 and n_hidden == 40):        # (that one exceeds the LDS: D-pass loop, same results)
        assert launches <= 2 * len(flow.bijection.layers), launches  # no D-pass loops
    z_ref, ld_ref = ref.forward(x.numpy())
    xr_ref, ldi_ref = ref.inverse(x.numpy())
    e = dict(z=rel(z.cpu().numpy(), z_ref), ld=rel(ld.cpu().numpy(), ld_ref),
             xr=rel(xr.cpu().numpy(), xr_ref), ldi=rel(ldi.cpu().numpy(), ldi_ref))
    print(arch, D, n_hidden, launches, {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 1e-5 * max(1.0, D / 64), e


_MADE_SPLINES = ["MaskedAutoregressiveRQNSF", "InverseAutoregressiveRQNSF", "MaskedAutoregressiveLRS", "InverseAutoregressiveLRS"]


# (D = 64: the host's fp64 D-pass truth takes 10-30 s per preset; the default run keeps one of the four, --runslow all)
@pytest.mark.parametrize("D,n_hidden,arch", [(D, h, a) for D, h in ((5, None), (16, 12)) for a in _MADE_SPLINES]
                         + [(64, None, "MaskedAutoregressiveRQNSF")]
                         + [pytest.param(64, None, a, marks=pytest.mark.slow) for a in _MADE_SPLINES[1:]])
def test_made_spline_sequential_map_in_one_launch(pkg, oracle, monkeypatch, arch, D, n_hidden):
    """The sequential map of MADE-based spline layers as ONE launch per layer (tfk_made_{rqs,lrs}_sequential),
    including the reference's last-pass log-det (layers_base.py:213-221): parity with the oracle's D-pass
    restatement and with this package's own D-pass loop on the device."""
    from torchflows_amd import native
    torch.manual_seed(D + (n_hidden or 0))
    kw = dict(n_layers=2)
    if n_hidden is not None:
        kw["conditioner_kwargs"] = dict(n_hidden=n_hidden)
    flow = pkg.Flow(getattr(pkg, arch)(D, **kw))
    flow.train()
    with torch.no_grad():          # (Inverse* presets: log_prob is the D-pass sequential map on the host -- a small batch)
        flow.log_prob(torch.randn(2048 if arch.startswith("Masked") else 192, D))
    flow.eval()
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, 2, sd)
    x = torch.randn(160 if D < 64 else 48, D) * 1.5        # (the host's fp64 D-pass loop is the slow part of this test)
    seq = "inverse" if arch.startswith("Masked") else "forward"
    import copy
    with torch.no_grad():                      # the ATen composite path on the host: fp64 truth, fp32 floor
        y64, ld64 = getattr(copy.deepcopy(flow).double().bijection, seq)(x.double())
        y32, ld32 = getattr(flow.bijection, seq)(x)
    floor = max(rel(y32.numpy(), y64.numpy()), rel(ld32.numpy(), ld64.numpy()))
    flow = flow.cuda()

    def run():
        before = native.calls
        with torch.no_grad():
            out = getattr(flow.bijection, seq)(x.cuda())
        return out, native.calls - before

    (y, ld), launches = run()
    assert launches <= 2 * len(flow.bijection.layers), launches          # no D-pass loops
    set_debug(monkeypatch, made_fused="0")
    (y_loop, ld_loop), launches_loop = run()
    set_debug(monkeypatch, made_fused="1")
    assert launches_loop >= 2 * D
    y_ref, ld_ref = getattr(ref, seq)(x.numpy())
    e = dict(y=rel(y.cpu().numpy(), y64.numpy()), ld=rel(ld.cpu().numpy(), ld64.numpy()),
             y_oracle=rel(y_ref, y64.numpy()), ld_oracle=rel(ld_ref, ld64.numpy()),
             y_loop=rel(y_loop.cpu().numpy(), y64.numpy()), ld_loop=rel(ld_loop.cpu().numpy(), ld64.numpy()))
    print(arch, D, n_hidden, launches, launches_loop, f"floor {floor:.1e}", {k: f"{v:.1e}" for k, v in e.items()})
    # every element's spline parameters depend on the elements inverted before it: rounding differences are
    # amplified along the row, so each implementation is held to the fp64 result at 3x the host's own
    # fp32-vs-fp64 distance (or 4e-5)
    assert max(e.values()) < max(4e-5, 3 * floor), (e, floor)
