"""Fused flow programs (tfk_flow_run: conditioner in-kernel, permutations / elementwise layers
folded) against the layer-by-layer HIP path, the CPU oracle and the golden fixtures.

Bars: log_prob / log_det within 1e-5 relative, rows within 1e-5 norm-wise and 2e-5
elementwise (the in-kernel dot products sum in a different order than rocBLAS; both are a
few ulp from the exact sum, and 3L+3 layers amplify that).
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


def data_init(flow, D, rows=2048):
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(rows, D))
    return flow.eval()


def run_both(flow, x, monkeypatch):
    """(fused results, layer-by-layer results) of log_prob, forward, inverse on the device."""
    from torchflows_amd import native
    out = []
    for fused_on, mfma_on in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", fused_on)
        monkeypatch.setenv("TORCHFLOWS_AMD_MFMA", mfma_on)
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        before = native.calls
        with torch.no_grad():
            z, lp = flow.forward_with_log_prob(x)
            lp_only = flow.log_prob(x)
            z2, ld = flow.bijection.forward(x)
            xr, ldi = flow.bijection.inverse(x)
        out.append(dict(z=z, lp=lp, lp_only=lp_only, z2=z2, ld=ld, xr=xr, ldi=ldi,
                        launches=native.calls - before))
    monkeypatch.setenv("TORCHFLOWS_AMD_MFMA", "1")
    return out


@pytest.mark.parametrize("arch", ["RealNVP", "NICE"])
@pytest.mark.parametrize("D", [16, 32, 64, 128, 256, 512])
@pytest.mark.parametrize("n_layers", [1, 2, 8])
def test_fused_vs_layerwise_vs_oracle(pkg, oracle, monkeypatch, arch, D, n_layers):
    torch.manual_seed(D + n_layers)
    ctor = getattr(pkg, arch)
    flow = data_init(pkg.Flow(ctor(D, n_layers=n_layers)), D)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, n_layers, sd)
    flow = flow.cuda()
    N = 1000 + D
    x = torch.randn(N, D)
    x[:50] *= 3
    fused, fused_valu, layer = run_both(flow, x.cuda(), monkeypatch)
    # D = 64 / 128 take the matrix-core kernel first, the others the vector-ALU one in both runs
    # the fused path really is a handful of launches, the other one is 3L+3 kernels + GEMMs
    assert fused["launches"] < layer["launches"]
    if D <= 64 and n_layers % 2 == 0:   # whole program in one LDS block, no final reordering
        assert fused["launches"] == 4, fused["launches"]
    z_ref, lp_ref = ref.log_prob(x.numpy(), return_z=True)
    _, ld_ref = ref.forward(x.numpy())
    xr_ref, ldi_ref = ref.inverse(x.numpy())
    for name, got in (("fused", fused), ("fused-valu", fused_valu), ("layerwise", layer)):
        e = dict(lp=rel(got["lp"].cpu().numpy(), lp_ref), lp_only=rel(got["lp_only"].cpu().numpy(), lp_ref),
                 z=normwise(got["z"].cpu().numpy(), z_ref), z2=normwise(got["z2"].cpu().numpy(), z_ref),
                 ld=rel(got["ld"].cpu().numpy(), ld_ref), xr=normwise(got["xr"].cpu().numpy(), xr_ref),
                 ldi=rel(got["ldi"].cpu().numpy(), ldi_ref))
        print(arch, D, n_layers, name, {k: f"{v:.1e}" for k, v in e.items()})
        assert max(e.values()) < 1e-5, (name, e)
    assert rel(fused["z"].cpu().numpy(), layer["z"].cpu().numpy()) < 2e-5
    assert rel(fused["xr"].cpu().numpy(), layer["xr"].cpu().numpy()) < 2e-5


@pytest.mark.parametrize("D", [16, 64, 128, 256])
@pytest.mark.parametrize("n_layers", [1, 2, 8])
def test_fused_spline_flow_vs_layerwise_vs_oracle(pkg, oracle, monkeypatch, D, n_layers):
    """CouplingRQNSF as fused flow programs (conditioner + 23-parameter GEMM + spline in one
    kernel).  Spline bar: 4e-5 (see tests/test_gpu_kernels.py for why not 1e-5)."""
    from torchflows_amd import fused
    torch.manual_seed(D + n_layers)
    flow = data_init(pkg.Flow(pkg.CouplingRQNSF(D, n_layers=n_layers)), D)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict("CouplingRQNSF", D, n_layers, sd)
    flow = flow.cuda()
    N = 700 + D
    x = torch.randn(N, D)
    x[:40] *= 30                                   # some rows leave the spline box
    got_fused, _, got_layer = run_both(flow, x.cuda(), monkeypatch)
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    flow.bijection.__dict__.pop("_tfk_compiled", None)
    # (D = 256, hidden width 17: two hidden tiles, operands streamed chunk by chunk -- round 1 ran it layer by layer)
    assert fused.get_compiled(flow.bijection, 0, torch.device("cuda", 0)) is not None
    assert got_fused["launches"] < got_layer["launches"]
    z_ref, lp_ref = ref.log_prob(x.numpy(), return_z=True)
    xr_ref, ldi_ref = ref.inverse(x.numpy())
    for name, got in (("fused", got_fused), ("layerwise", got_layer)):
        e = dict(lp=rel(got["lp"].cpu().numpy(), lp_ref), lp_only=rel(got["lp_only"].cpu().numpy(), lp_ref),
                 z=normwise(got["z"].cpu().numpy(), z_ref), xr=normwise(got["xr"].cpu().numpy(), xr_ref),
                 ldi=rel(got["ldi"].cpu().numpy(), ldi_ref))
        print("NSF", D, n_layers, name, {k: f"{v:.1e}" for k, v in e.items()})
        # log-dets sum T = D/2 spline terms per layer: bound per 32 terms, as in test_gpu_kernels
        tol = {k: 4e-5 * (max(1.0, D / 64) if k in ("lp", "lp_only", "ldi") else 1.0) for k in e}
        assert all(e[k] < tol[k] for k in e), (name, e)


@pytest.mark.parametrize("variant", ["fresh", "init"])
def test_fused_spline_golden(pkg, monkeypatch, variant):
    from torchflows_amd import fused
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    fx = load_golden("flow_nsf64.npz")
    flow = pkg.Flow(pkg.CouplingRQNSF(64, n_layers=8))
    flow.load_state_dict({k: torch.from_numpy(v) for k, v in state_dict_of(fx, variant).items()})
    flow = flow.cuda().eval()
    assert fused.get_compiled(flow.bijection, 0, torch.device("cuda", 0)) is not None
    with torch.no_grad():
        lp = flow.log_prob(torch.from_numpy(fx["x"]).cuda())
        xr, ldr = flow.bijection.inverse(torch.from_numpy(fx["z_in"]).cuda())
    g = lambda k: fx[f"{variant}/{k}"]
    assert rel(lp.cpu().numpy(), g("log_prob")) < max(4e-5, 3 * rel(g("log_prob"), g("log_prob64")))
    assert normwise(xr.cpu().numpy(), g("x_inv")) < max(4e-5, 3 * normwise(g("x_inv"), g("x_inv64")))
    assert rel(ldr.cpu().numpy(), g("log_det_inv")) < max(4e-5, 3 * rel(g("log_det_inv"), g("log_det_inv64")))


@pytest.mark.parametrize("name,arch,n_layers", [("flow_realnvp64.npz", "RealNVP", 8),
                                                ("flow_realnvp256.npz", "RealNVP", 8)])
@pytest.mark.parametrize("variant", ["fresh", "init"])
def test_fused_golden(pkg, monkeypatch, name, arch, n_layers, variant):
    from torchflows_amd import fused, native
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    fx = load_golden(name)
    D = int(fx["event_shape"][0])
    flow = pkg.Flow(pkg.RealNVP(D, n_layers=n_layers))
    flow.load_state_dict({k: torch.from_numpy(v) for k, v in state_dict_of(fx, variant).items()})
    flow = flow.cuda().eval()
    assert fused.get_compiled(flow.bijection, 0, torch.device("cuda", 0)) is not None
    before = native.calls
    with torch.no_grad():
        lp = flow.log_prob(torch.from_numpy(fx["x"]).cuda())
        launches = native.calls - before
        xr, ldr = flow.bijection.inverse(torch.from_numpy(fx["z_in"]).cuda())
    g = lambda k: fx[f"{variant}/{k}"]
    assert launches == (1 if D == 64 else launches)
    assert rel(lp.cpu().numpy(), g("log_prob")) < 1e-5
    assert normwise(xr.cpu().numpy(), g("x_inv")) < 1e-5
    # log-dets sum D/2 terms per layer that largely cancel: 1e-5 per 32 terms (as in
    # test_gpu_kernels), or 3x the reference's own fp32-vs-fp64 distance if that is larger
    assert rel(ldr.cpu().numpy(), g("log_det_inv")) < max(1e-5 * max(1.0, D / 64),
                                                          3 * rel(g("log_det_inv"), g("log_det_inv64")))


def test_segmented_program_matches_single_launch(pkg, monkeypatch):
    """A program split over several launches (small LDS budget) gives the same result."""
    from torchflows_amd import fused
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    monkeypatch.setenv("TORCHFLOWS_AMD_MFMA", "0")     # vector-ALU programs: bitwise invariant
    torch.manual_seed(0)
    flow = data_init(pkg.Flow(pkg.RealNVP(64, n_layers=5)), 64).cuda()
    x = torch.randn(777, 64, device="cuda")
    res = []
    for budget in (40 * 1024, 6 * 1024, 4 * 1024):
        monkeypatch.setattr(fused, "MAX_PARAM_BYTES", budget)
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        chain = fused.get_compiled(flow.bijection, 0, x.device)
        with torch.no_grad():
            z, lp = flow.forward_with_log_prob(x)
            lp2 = flow.log_prob(x)
            xr, ldi = flow.bijection.inverse(x)
        res.append((len(chain.segments), z, lp, lp2, xr, ldi))
    assert res[0][0] == 1 and res[1][0] > 1 and res[2][0] > res[1][0]
    for r in res[1:]:
        for a, b in zip(r[1:], res[0][1:]):
            assert torch.equal(a, b)          # same arithmetic, only the launch boundaries move
    # matrix-core programs keep per-lane log-det partial sums that are combined at the end of a
    # launch, so moving the launch boundaries changes the association of the fp32 sum: close,
    # not bitwise
    monkeypatch.setenv("TORCHFLOWS_AMD_MFMA", "1")
    res_m = []
    for budget in (52 * 1024, 12 * 1024):
        monkeypatch.setattr(fused, "MAX_PARAM_BYTES_MFMA", budget)
        monkeypatch.setattr(fused, "LEAN_BUDGET_64", budget)
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        chain = fused.get_compiled(flow.bijection, 0, x.device)
        assert all(seg.mfma for seg in chain.segments)
        with torch.no_grad():
            z, lp = flow.forward_with_log_prob(x)
            xr, ldi = flow.bijection.inverse(x)
        res_m.append((len(chain.segments), z, lp, xr, ldi))
    assert res_m[0][0] == 1 and res_m[1][0] > 1
    for a, b, ref in zip(res_m[1][1:], res_m[0][1:], (res[0][1], res[0][2], res[0][4], res[0][5])):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-6)
        assert torch.allclose(a, ref, rtol=1e-5, atol=1e-5)


def test_odd_layer_count_keeps_logical_order(pkg, oracle, monkeypatch):
    """With an odd number of reversals the physical order at the end is reversed: the rows
    handed back must still be in logical order, and the base density must see it too."""
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    torch.manual_seed(1)
    from torchflows_amd.base_distributions.gaussian import DiagonalGaussian
    D = 32
    b = pkg.RealNVP(D, n_layers=3)
    base = DiagonalGaussian(torch.randn(D), torch.rand(D) + 0.5)
    flow = data_init(pkg.Flow(b, base_distribution=base), D)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict("RealNVP", D, 3, sd)
    x = torch.randn(500, D)
    flow = flow.cuda()
    with torch.no_grad():
        z, lp = flow.forward_with_log_prob(x.cuda())
        lp_only = flow.log_prob(x.cuda())
    z_ref, lp_ref = ref.log_prob(x.numpy(), return_z=True)
    assert normwise(z.cpu().numpy(), z_ref) < 1e-5
    assert rel(lp.cpu().numpy(), lp_ref) < 1e-5 and rel(lp_only.cpu().numpy(), lp_ref) < 1e-5


def test_cache_follows_parameter_updates_and_train_mode(pkg, monkeypatch):
    from torchflows_amd import fused
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    torch.manual_seed(2)
    flow = pkg.Flow(pkg.RealNVP(16, n_layers=2)).cuda()
    x = torch.randn(300, 16, device="cuda")
    # train mode, ActNorm not initialised yet: not compilable, layer-by-layer path initialises it
    assert flow.training
    assert fused.get_compiled(flow.bijection, 0, x.device) is None
    with torch.no_grad():
        lp_train = flow.log_prob(x)
    flow.eval()
    assert fused.get_compiled(flow.bijection, 0, x.device) is not None
    with torch.no_grad():
        lp_eval = flow.log_prob(x)
        assert torch.allclose(lp_eval, lp_train, rtol=2e-5, atol=2e-5)
        flow.bijection.layers[0].value.mul_(0.5)          # in-place update (an optimizer step)
        lp_new = flow.log_prob(x)
        monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "0")
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        lp_layer = flow.log_prob(x)
    assert not torch.allclose(lp_new, lp_eval)
    assert torch.allclose(lp_new, lp_layer, rtol=2e-5, atol=2e-5)


def test_flow_run_argument_checks(pkg):
    from torchflows_amd import native
    x = torch.zeros(8, 64, device="cuda")
    params = torch.zeros(132, device="cuda")
    lp = torch.empty(8, device="cuda")
    native.flow_run(x, None, lp, None, None, None, [(0, 0, 0, 0)], params)       # fine
    with pytest.raises(native.NativeError):      # parameters run past the block
        native.flow_run(x, None, lp, None, None, None, [(0, 0, 0, 8)], params)
    with pytest.raises(native.NativeError):      # unknown op
        native.flow_run(x, None, lp, None, None, None, [(9, 0, 0, 0)], params)
    with pytest.raises(native.NativeError):      # D not supported
        native.flow_run(torch.zeros(8, 24, device="cuda"), None, lp, None, None, None, [], params)
    with pytest.raises(native.NativeError):      # no output
        native.flow_run(x, None, None, None, None, None, [(0, 0, 0, 0)], params)


@pytest.mark.parametrize("arch", ["RealNVP", "NICE"])
@pytest.mark.parametrize("D,n_hidden", [(64, 17), (64, 24), (64, 32), (64, 40), (64, 64), (128, 48), (256, 33),
                                        (64, 65), (64, 100), (64, 128), (128, 96), (30, 72)])
def test_fused_wide_hidden_layers(pkg, oracle, monkeypatch, arch, D, n_hidden):
    """Hidden widths 17..128 run on the matrix-core flow program too (2, 4 or 8 tiles of 16 units in
    GEMM 1, up to 32 k-steps in GEMM 2; above 64 units one coupling's operands fill a launch's LDS, so the chain is one
    launch per coupling -- still a quarter of the layer-by-layer route's); parity with the oracle and that route."""
    from torchflows_amd import fused as fz
    torch.manual_seed(D + n_hidden)
    ctor = getattr(pkg, arch)
    flow = data_init(pkg.Flow(ctor(D, n_layers=4, conditioner_kwargs=dict(n_hidden=n_hidden))), D)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, 4, sd)
    flow = flow.cuda()
    chain = fz.get_compiled(flow.bijection, 0, torch.device("cuda", 0))
    assert chain is not None and all(seg.mfma for seg in chain.segments)
    x = torch.randn(777, D)
    x[:40] *= 3
    fused, fused_valu, layer = run_both(flow, x.cuda(), monkeypatch)
    assert fused["launches"] < layer["launches"]
    if n_hidden > 64:
        # (four passes: forward_with_log_prob, log_prob, forward, inverse -- one launch per coupling each, and one for a
        # leading elementwise layer that does not fit beside the first)
        assert len(chain.segments) <= 5 and fused["launches"] <= 4 * (len(chain.segments) + 1), (len(chain.segments), fused["launches"])
    z_ref, lp_ref = ref.log_prob(x.numpy(), return_z=True)
    xr_ref, ldi_ref = ref.inverse(x.numpy())
    for name, got in (("fused", fused), ("layerwise", layer)):
        e = dict(lp=rel(got["lp"].cpu().numpy(), lp_ref), z=normwise(got["z"].cpu().numpy(), z_ref),
                 xr=normwise(got["xr"].cpu().numpy(), xr_ref), ldi=rel(got["ldi"].cpu().numpy(), ldi_ref))
        print(arch, D, n_hidden, name, {k: f"{v:.1e}" for k, v in e.items()})
        assert max(e.values()) < 1e-5, (name, e)


@pytest.mark.parametrize("arch,D", [("RealNVP", 64), ("RealNVP", 6), ("CouplingRQNSF", 64)])
def test_views_into_larger_buffers(pkg, oracle, monkeypatch, arch, D):
    """Ragged inputs the reference accepts: a view that starts 4 B into a buffer (not 16-byte aligned),
    a strided slice, a transposed batch, one row and no rows -- same values as the aligned copy."""
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    torch.manual_seed(3)
    flow = data_init(pkg.Flow(getattr(pkg, arch)(D, n_layers=2)), D)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, 2, sd)
    flow = flow.cuda()
    N = 301
    buf = torch.randn(N * D + 1, device="cuda")
    off = buf[1:].view(N, D)                                   # data_ptr % 16 == 4
    assert off.data_ptr() % 16 != 0
    wide = torch.randn(N, 2 * D, device="cuda")
    strided = wide[:, ::2]                                     # non-contiguous
    batch2 = torch.randn(7, 5, D, device="cuda").transpose(0, 1)   # batch shape (5, 7), permuted strides
    tol = 1e-5 if arch == "RealNVP" else 4e-5
    with torch.no_grad():
        for x in (off, strided, batch2, off[:1], off[:0]):
            lp = flow.log_prob(x)
            z, ld = flow.bijection.forward(x)
            xr, ldi = flow.bijection.inverse(z)
            assert lp.shape == x.shape[:-1] and z.shape == x.shape and ld.shape == x.shape[:-1]
            if x.numel() == 0:
                continue
            lp_ref = ref.log_prob(x.reshape(-1, D).cpu().numpy())
            assert rel(lp.reshape(-1).cpu().numpy(), lp_ref) < tol
            assert normwise(xr.cpu().numpy(), x.cpu().numpy()) < 1e-4
            assert rel((ld + ldi).cpu().numpy(), 0.0) < 1e-3
        # sampling noise handed over as a misaligned view (Flow.sample's inverse pass)
        x2, _ = flow.bijection.inverse(off)
        x3, _ = flow.bijection.inverse(off.clone())
        assert torch.equal(x2, x3)


@pytest.mark.parametrize("arch,D,n_hidden", [("MaskedAutoregressiveRQNSF", 64, None),
                                             pytest.param("InverseAutoregressiveRQNSF", 64, None, marks=pytest.mark.slow),
                                             ("MaskedAutoregressiveRQNSF", 128, None)])
def test_made_spline_parallel_map_as_flow_program(pkg, oracle, monkeypatch, arch, D, n_hidden):
    """The parallel map of MADE-based RQ-spline layers (MA-RQNSF density, IA-RQNSF sampling) as matrix-core
    flow-program ops (TFK_OP_MADE_RQS): fused vs layer by layer vs the oracle."""
    from torchflows_amd import native
    torch.manual_seed(D)
    kw = dict(n_layers=3)
    if n_hidden is not None:
        kw["conditioner_kwargs"] = dict(n_hidden=n_hidden)
    # (the data-dependent initialisation goes through log_prob: for the inverse-autoregressive flows that is the
    # element-by-element map on the host, D passes per layer -- 256 rows are plenty for ActNorm's statistics)
    flow = data_init(pkg.Flow(getattr(pkg, arch)(D, **kw)), D, rows=2048 if arch.startswith("Masked") else 256)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, 3, sd)
    x = torch.randn(777, D) * 1.3
    par = "forward" if arch.startswith("Masked") else "inverse"
    import copy
    with torch.no_grad():                      # ATen composite path on the host: fp64 truth and fp32 floor
        y64, ld64 = getattr(copy.deepcopy(flow).double().bijection, par)(x.double())
        y32, ld32 = getattr(flow.bijection, par)(x)
    floor_y, floor_ld = rel(y32.numpy(), y64.numpy()), rel(ld32.numpy(), ld64.numpy())
    flow = flow.cuda()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", mode)
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        before = native.calls
        with torch.no_grad():
            y, ld = getattr(flow.bijection, par)(x.cuda())
        res[mode] = (y.cpu().numpy(), ld.cpu().numpy(), native.calls - before)
    monkeypatch.setenv("TORCHFLOWS_AMD_FUSED", "1")
    assert res["1"][2] < res["0"][2] and res["1"][2] <= 6, (res["1"][2], res["0"][2])   # one launch per MADE layer (+ the rest)
    y_ref, ld_ref = getattr(ref, par)(x.numpy())
    e = dict(y=rel(res["1"][0], y64.numpy()), ld=rel(res["1"][1], ld64.numpy()),
             y_lw=rel(res["0"][0], y64.numpy()), ld_lw=rel(res["0"][1], ld64.numpy()),
             y_oracle=rel(y_ref, y64.numpy()), ld_oracle=rel(ld_ref, ld64.numpy()),
             y_norm=normwise(res["1"][0], y64.numpy()))
    print(arch, D, n_hidden, res["1"][2], res["0"][2], f"floor {floor_y:.1e} / {floor_ld:.1e}",
          {k: f"{v:.1e}" for k, v in e.items()})
    # three stacked spline layers amplify rounding differences elementwise (1 / bin width): every
    # implementation is held to the fp64 result at 3x the host's own fp32-vs-fp64 distance (or 4e-5)
    assert max(e["y"], e["y_lw"], e["y_oracle"]) < max(4e-5, 3 * floor_y)
    assert max(e["ld"], e["ld_lw"], e["ld_oracle"]) < max(4e-5 * max(1.0, D / 64), 3 * floor_ld)
    assert e["y_norm"] < 2e-5


@pytest.mark.parametrize("arch,D", [("RealNVP", 64), ("NICE", 64), ("MAF", 64)])
def test_sample_log_prob_rides_in_the_inverse_program(pkg, oracle, arch, D):
    """Flow.sample(return_log_prob=True) = (inverse(z), base_log_prob(z) + log_det) (flows.py:699-707): when the
    inverse chain is one matrix-core program the base density of the incoming rows is evaluated in the same
    launch (flag bit 2 of tfk_flow_run_mfma); same values as the two-step path and the oracle."""
    from torchflows_amd import native
    torch.manual_seed(9)
    flow = data_init(pkg.Flow(getattr(pkg, arch)(D, n_layers=3)), D)
    with torch.no_grad():
        flow.base.loc.normal_()
        flow.base.log_scale.uniform_(-0.5, 0.5)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, 3, sd)
    flow = flow.cuda()
    z = (torch.randn(1000, D) * torch.exp(torch.tensor(sd["base.log_scale"])) + torch.tensor(sd["base.loc"])).cuda()
    with torch.no_grad():
        before = native.calls
        got = flow._fused_sample(z)
        launches = native.calls - before
        x2, ld2 = flow.bijection.inverse(z)
        lp2 = flow.base_log_prob(z) + ld2
    if arch == "MAF":                      # its sequential map is not a flow program: the generic path stays
        assert got is None
        return
    x1, lp1 = got
    assert launches <= 2                  # the program (+ one permute: an odd number of reversals)
    assert torch.equal(x1, x2)
    assert rel(lp1.cpu().numpy(), lp2.cpu().numpy()) < 2e-6
    x_ref, ld_ref = ref.inverse(z.cpu().numpy())
    lp_ref = ref.base_log_prob(z.cpu().numpy()) + ld_ref if hasattr(ref, "base_log_prob") else None
    assert normwise(x1.cpu().numpy(), x_ref) < 1e-5
    if lp_ref is not None:
        assert rel(lp1.cpu().numpy(), lp_ref) < 1e-5
    torch.manual_seed(3)
    xs, lps = flow.sample((512,), return_log_prob=True)      # the public entry point takes the same route
    assert xs.shape == (512, D) and lps.shape == (512,) and torch.isfinite(lps).all()


@pytest.mark.parametrize("arch,D,n_layers", [("RealNVP", 6, 2), ("RealNVP", 22, 3), ("NICE", 40, 4), ("RealNVP", 62, 8),
                                             ("RealNVP", 100, 3), ("RealNVP", 200, 2),
                                             ("RealNVP", 3, 2), ("RealNVP", 7, 3), ("NICE", 21, 4), ("RealNVP", 43, 8),
                                             ("RealNVP", 63, 5), ("RealNVP", 15, 4), ("RealNVP", 31, 3), ("NICE", 5, 3),
                                             ("RealNVP", 99, 3), ("RealNVP", 127, 2), ("NICE", 201, 3), ("RealNVP", 201, 9), ("CouplingRQNSF", 22, 3), ("CouplingRQNSF", 8, 2),
                                             ("CouplingRQNSF", 21, 2), ("CouplingRQNSF", 100, 2), ("CouplingRQNSF", 7, 2),
                                             ("CouplingRQNSF", 63, 3), ("CouplingRQNSF", 99, 2),
                                             ("MAF", 6, 2), ("MAF", 21, 3), ("MAF", 43, 2), ("MAF", 100, 2), ("MAF", 7, 2), ("MAF", 99, 2),
                                             ("MaskedAutoregressiveRQNSF", 22, 2)])
def test_even_event_sizes_run_as_padded_flow_programs(pkg, oracle, monkeypatch, arch, D, n_layers):
    """Event sizes other than 64 / 128 / 256.  Even: both halves of the row are padded to the next supported
    plane width (zero weights => identity on the padding, a base log_scale of -0.5 log 2 pi => no density
    term).  Odd (HalfSplit moves one element across the halves at every reversal): every element keeps its
    own index in both planes and changes planes by TFK_OP_PLANE_SWAP.  The whole chain still runs as
    matrix-core flow programs.  Parity with the oracle and with the layer-by-layer path; the padded path
    must actually be taken."""
    from torchflows_amd import native
    torch.manual_seed(D)
    from torchflows_amd.base_distributions.gaussian import DiagonalGaussian
    base = DiagonalGaussian(torch.randn(D) * 0.3, torch.rand(D) + 0.5)
    flow = data_init(pkg.Flow(getattr(pkg, arch)(D, n_layers=n_layers), base_distribution=base), D)
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    ref = oracle.preset_from_state_dict(arch, D, n_layers, sd)
    x = torch.randn(700, D)
    flow = flow.cuda()
    res = {}
    for mode in ("1", "0"):
        set_debug(monkeypatch, fused_pad=mode)
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        before = native.calls
        with torch.no_grad():
            lp = flow.log_prob(x.cuda())
            z, ld = flow.bijection.forward(x.cuda())
            xr, ldi = flow.bijection.inverse(z)
        res[mode] = (lp, z, ld, xr, ldi, native.calls - before)
    set_debug(monkeypatch, fused_pad="1")
    assert res["1"][5] < res["0"][5] and res["1"][5] <= 3 * 4 + (n_layers if "M" == arch[0] else 0), (res["1"][5], res["0"][5])
    z_ref, lp_ref = ref.log_prob(x.numpy(), return_z=True)
    lp, z, ld, xr, ldi, _ = res["1"]
    e = dict(lp=rel(lp.cpu().numpy(), lp_ref), z=normwise(z.cpu().numpy(), z_ref),
             lp_lw=rel(lp.cpu().numpy(), res["0"][0].cpu().numpy()), ld_lw=rel(ld.cpu().numpy(), res["0"][2].cpu().numpy()),
             round_trip=normwise(xr.cpu().numpy(), x.numpy()), ld_sum=rel((ld + ldi).cpu().numpy(), 0.0))
    print(arch, D, n_layers, res["1"][5], res["0"][5], {k: f"{v:.1e}" for k, v in e.items()})
    assert z.shape == x.shape and xr.shape == x.shape
    if D % 2 and arch in ("RealNVP", "NICE", "CouplingRQNSF"):
        # odd event sizes of affine / shift chains: the straight-line kernel with the middle element changing planes
        # (round 3) -- ONE launch per pass at the narrowest row width that holds (D + 1) / 2 columns per plane; against the
        # interpreter's plane-per-element route (TORCHFLOWS_AMD_DEBUG=odd_lean=0) to fp32 rounding
        from torchflows_amd import fused as fz
        flow.invalidate_native_caches()            # (the cache remembers that the chain was declined without padding)
        chain = fz.get_compiled(flow.bijection, 0, torch.device("cuda", 0))
        want_w = next(w for w in ((32, 64, 128) if "RQ" in arch else (16, 32, 64, 128, 256)) if (D + 1) // 2 <= w // 2)
        # (RealNVP(201, 9 layers): 198 KB of operands at 256 columns -- odd sizes keep them resident, so two launches per pass)
        n_seg = 2 if (D, n_layers) == (201, 9) else 1
        assert chain is not None and chain.D == want_w and len(chain.segments) == n_seg, (chain.D, len(chain.segments))
        assert res["1"][5] == 3 * n_seg
        set_debug(monkeypatch, odd_lean="0")
        flow.invalidate_native_caches()
        with torch.no_grad():
            lp_i = flow.log_prob(x.cuda())
            z_i, ld_i = flow.bijection.forward(x.cuda())
            xs, lps = flow.sample((300,), return_log_prob=True)
        set_debug(monkeypatch, odd_lean=None)
        flow.invalidate_native_caches()
        tol_i = 4e-5 if "RQ" in arch else 1e-5
        assert rel(lp.cpu().numpy(), lp_i.cpu().numpy()) < tol_i and normwise(z.cpu().numpy(), z_i.cpu().numpy()) < tol_i
        with torch.no_grad():                      # the sampled rows evaluate to the density the other route returned
            x_back, ld_back = flow.bijection.inverse(flow.bijection.forward(xs)[0])
        assert normwise(x_back.cpu().numpy(), xs.cpu().numpy()) < 1e-4
    if D % 2 and arch == "MAF":
        # odd MAF: any layout serves a MADE layer (every element is read and transformed), so the density direction is the
        # lean chain at the next row width, read in place -- ONE launch per log_prob
        from torchflows_amd import fused as fz
        flow.invalidate_native_caches()
        chain = fz.get_compiled(flow.bijection, 0, torch.device("cuda", 0))
        assert chain is not None and chain.D == next(w for w in (32, 64, 128) if (D + 1) // 2 <= w // 2)
        assert len(chain.segments) == 1 and chain.segments[0].ops[0][0] == fz.OP_MADE_FWD_LEAN
        before = native.calls
        with torch.no_grad():
            lp_again = flow.log_prob(x.cuda())
        assert native.calls - before == 1 and torch.equal(lp_again, lp)
    tol = 4e-5 if "RQ" in arch else 1e-5
    assert max(e["lp"], e["lp_lw"], e["ld_lw"]) < tol * max(1.0, D / 64) and e["z"] < 2 * tol
    assert e["round_trip"] < 10 * tol and e["ld_sum"] < 1e-4


def test_data_edit_needs_invalidate_and_invalidate_refreshes_every_pack():
    """an in-place edit through ``.data`` does not move the version counters the packed-weight caches are
    keyed on; ``invalidate_native_caches()`` must refresh every pack (flow programs, MADE packs).  load_state_dict and
    train() / eval() refresh by themselves."""
    import torchflows_amd as tfa
    for ctor in (tfa.RealNVP, tfa.MAF):
        torch.manual_seed(0)
        flow = tfa.Flow(ctor(64, n_layers=2)).eval()
        x = torch.randn(300, 64)
        dev_flow = tfa.Flow(ctor(64, n_layers=2)).eval()
        dev_flow.load_state_dict(flow.state_dict())
        dev_flow = dev_flow.cuda()
        with torch.no_grad():
            before = dev_flow.log_prob(x.cuda()).cpu()
            for p, q in zip(flow.parameters(), dev_flow.parameters()):
                if p.requires_grad:
                    p.data.mul_(0.5)
                    q.data.mul_(0.5)          # invisible to _version
            want = flow.log_prob(x)
            dev_flow.invalidate_native_caches()
            got = dev_flow.log_prob(x.cuda()).cpu()
        assert not torch.allclose(before, want, atol=1e-3)
        assert rel(got.numpy(), want.numpy()) < 1e-5
        # load_state_dict refreshes without being asked
        with torch.no_grad():
            sd = {k: (v * 2 if v.is_floating_point() and "weight" in k else v) for k, v in flow.state_dict().items()}
            flow.load_state_dict(sd)
            dev_flow.load_state_dict(sd)
            assert rel(dev_flow.log_prob(x.cuda()).cpu().numpy(), flow.log_prob(x).numpy()) < 1e-5


def test_nested_composition_with_unsupported_layer_falls_back():
    """a BijectiveComposition nested in another one, holding a layer without a native step
    (ElementwiseRQSpline), must make the OUTER composition take the ATen loop instead of raising NativeError."""
    from torchflows_amd.bijections.base import BijectiveComposition
    from torchflows_amd.bijections.finite.autoregressive.layers import AffineCoupling, ElementwiseRQSpline, ActNorm
    torch.manual_seed(0)
    inner = BijectiveComposition([ElementwiseRQSpline((6,)), AffineCoupling((6,))])
    outer = BijectiveComposition([ActNorm((6,)), inner, AffineCoupling((6,))]).eval()
    x = torch.randn(50, 6)
    with torch.no_grad():
        z_h, ld_h = outer.forward(x)
        outer = outer.cuda()
        z_d, ld_d = outer.forward(x.cuda())
        xr, ldr = outer.inverse(z_d)
    assert rel(z_d.cpu().numpy(), z_h.numpy()) < 4e-5 and rel(ld_d.cpu().numpy(), ld_h.numpy()) < 4e-5
    assert torch.allclose(xr.cpu(), x, atol=1e-4)


def test_declined_composition_warns_once():
    """A composition the flow-program compiler declines (here: a conditioner deeper than Linear-Tanh-Linear) runs
    layer by layer -- about 10x slower -- and says so ONCE (fused.NativeRouteWarning); a compiled one stays silent."""
    import warnings
    import torchflows_amd as tfa
    from torchflows_amd import fused
    torch.manual_seed(0)
    deep = tfa.Flow(tfa.RealNVP(6, n_layers=2, conditioner_kwargs=dict(n_layers=3))).eval().cuda()
    x = torch.randn(40, 6, device="cuda")
    with torch.no_grad():
        with pytest.warns(fused.NativeRouteWarning, match="layer by layer"):
            deep.log_prob(x)
        with warnings.catch_warnings():
            warnings.simplefilter("error", fused.NativeRouteWarning)
            deep.log_prob(x)                                     # second call: silent
            plain = tfa.Flow(tfa.RealNVP(64, n_layers=2)).eval().cuda()
            plain.log_prob(torch.randn(40, 64, device="cuda"))   # compiled: silent


@pytest.mark.parametrize("name,arch,n_layers,ctx_shape", [("flow_realnvp5_ctx3.npz", "RealNVP", 2, (3,)),
                                                          ("flow_nsf6_ctx2.npz", "CouplingRQNSF", 2, (2,))])
@pytest.mark.parametrize("variant", ["fresh", "init"])
def test_context_conditioned_flows_run_as_flow_programs(name, arch, n_layers, ctx_shape, variant):
    """Context-conditioned flows (VERDICT r1 missing 5): couplings whose conditioner sees [x_A || context] and elementwise
    layers whose parameters are a Linear map of the context run as flow programs (tfk_flow_run_mfma_ctx; spline chains:
    the lean single-launch chain between two small elementwise launches) -- against the reference's golden outputs; the
    number of libtfk launches is the number of program launches."""
    import warnings
    import torchflows_amd as tfa
    from torchflows_amd import native, fused
    from conftest import load_golden, state_dict_of
    fx = load_golden(name)
    es = tuple(int(v) for v in fx["event_shape"])
    flow = tfa.Flow(getattr(tfa, arch)(es, n_layers=n_layers, context_shape=ctx_shape))
    flow.load_state_dict({k: torch.from_numpy(v) for k, v in state_dict_of(fx, variant).items()})
    flow = flow.cuda().eval()
    x, ctx = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["context"]).cuda()
    z_in = torch.from_numpy(fx["z_in"]).cuda()
    g = lambda k: fx[f"{variant}/{k}"]
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error", fused.NativeRouteWarning)      # nothing may fall to the layer-by-layer route
        before = native.calls
        lp = flow.log_prob(x, context=ctx)
        # program launches only: the context-conditioned elementwise layer in front, the lean chain (the context as further
        # GEMM-1 k-steps), the elementwise layers behind it
        # (odd event sizes: one interpreter program; even ones: the lean chain with the elementwise layers inside the launch)
        assert native.calls - before == 1
        before = native.calls
        z, lp2 = flow.forward_with_log_prob(x, context=ctx)
        xr, ldr = flow.bijection.inverse(z_in, context=ctx)
        assert native.calls - before <= 6
    tol = 4e-5 if "RQ" in arch else 1e-5
    assert rel(lp.cpu().numpy(), g("log_prob")) < 1e-5 and torch.equal(lp, lp2)
    assert normwise(z.cpu().numpy(), g("z")) < tol
    assert normwise(xr.cpu().numpy(), g("x_inv")) < tol and rel(ldr.cpu().numpy(), g("log_det_inv")) < tol


@pytest.mark.parametrize("arch,D", [("RealNVP", 22), ("RealNVP", 62), ("RealNVP", 100), ("NICE", 10), ("CouplingRQNSF", 22),
                                    ("CouplingRQNSF", 6), ("RealNVP", 16), ("RealNVP", 8), ("RealNVP", 4), ("NICE", 16),
                                    ("NICE", 6)])
def test_narrow_rows_are_read_in_place(monkeypatch, arch, D):
    """Event sizes that are not 64 / 128 / 256: lean programs read the caller's (N, D) rows themselves
    (tfk_flow_run_mfma_in) instead of a host-side padding pass; same values as with the padding pass, and as the
    host (ATen) path."""
    import torchflows_amd as tfa
    from torchflows_amd import native
    torch.manual_seed(1)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=3))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D))
    flow.eval()
    x = torch.randn(1000, D)
    with torch.no_grad():
        lp_h = flow.log_prob(x)
        flow = flow.cuda()
        xd = x.cuda()
        before = native.calls
        z1, lp1 = flow.forward_with_log_prob(xd)
        n_calls = native.calls - before
        xr1, ld1 = flow.bijection.inverse(z1)
        set_debug(monkeypatch, narrow_in="0")
        z0, lp0 = flow.forward_with_log_prob(xd)
    assert n_calls == 1                                         # one launch, no padding kernels
    tol = 4e-5 if "RQ" in arch else 1e-5
    assert rel(lp1.cpu().numpy(), lp_h.numpy()) < tol
    assert torch.equal(lp1, lp0) and torch.equal(z1, z0)
    assert torch.allclose(xr1.cpu(), x, atol=1e-3)
    if D <= 16 and "RQ" not in arch:
        # affine / shift chains on event sizes <= 16 run at row width 16 (two row elements per lane and plane); the same
        # values as at width 32 up to the order of the (zero) padding terms, and Flow.sample's inverse launch as well
        from torchflows_amd import fused as fz
        set_debug(monkeypatch, narrow_in=None)
        chain = fz.get_compiled(flow.bijection, 0, torch.device("cuda", 0))
        assert chain is not None and chain.D == 16 and len(chain.segments) == 1
        with torch.no_grad():
            before = native.calls
            torch.manual_seed(11)
            xs, lps = flow.sample((500,), return_log_prob=True)
            assert native.calls - before <= 2
            set_debug(monkeypatch, rows16="0")
            flow.invalidate_native_caches()
            z32, lp32 = flow.forward_with_log_prob(xd)
            xr32, _ = flow.bijection.inverse(z1)
            assert fz.get_compiled(flow.bijection, 0, torch.device("cuda", 0)).D == 32
            torch.manual_seed(11)                              # the same noise through the 32-wide program
            xs32, lps32 = flow.sample((500,), return_log_prob=True)
        assert rel(lp1.cpu().numpy(), lp32.cpu().numpy()) < 2e-6 and normwise(z1.cpu().numpy(), z32.cpu().numpy()) < 2e-6
        assert normwise(xr1.cpu().numpy(), xr32.cpu().numpy()) < 2e-6
        assert normwise(xs.cpu().numpy(), xs32.cpu().numpy()) < 2e-6 and rel(lps.cpu().numpy(), lps32.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("arch,D,n_layers", [("RealNVP", 64, 4), ("RealNVP", 256, 8), ("NICE", 256, 8), ("RealNVP", 256, 2)])
def test_affine_chain_bf16x3_operands(monkeypatch, arch, D, n_layers):
    """GEMM 2 of affine / shift chains on the bf16 matrix pipe at fp32 accuracy (three bf16 pieces per operand, six of the
    nine piece products).  D = 64: opt-in (TORCHFLOWS_AMD_DEBUG=lean_bf16x3=1, no faster than fp32 operands there); D = 256:
    the default since round 4 (streamed 42 KB blocks, 617 -> 545 us per RealNVP-256 launch).  Same values as the fp32 operand
    format to fp32 rounding, the host's fp64 evaluation within 1e-5, one launch per log_prob, the inverse undoes the forward."""
    import copy
    import torchflows_amd as tfa
    from torchflows_amd import fused, native
    torch.manual_seed(2)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=n_layers))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D))
    flow.eval()
    x = torch.randn(2000, D)
    x[:100] *= 4.0
    with torch.no_grad():
        lp_h = copy.deepcopy(flow).double().log_prob(x.double())
        flow = flow.cuda()
        set_debug(monkeypatch, lean_bf16x3="0")
        lp0 = flow.log_prob(x.cuda())
        set_debug(monkeypatch, lean_bf16x3="1")
        flow.invalidate_native_caches()
        chain = fused.get_compiled(flow.bijection, 0, torch.device("cuda", 0))
        is3 = any(len(op) > 4 and int(op[4]) == 256 for seg in chain.segments for op in seg.ops)
        # (D = 256: the format exists as streamed operands only -- a chain of two couplings keeps its fp32 blocks resident)
        assert is3 == (not (D == 256 and n_layers == 2)), "operand format"
        before = native.calls
        lp1 = flow.log_prob(x.cuda())
        assert native.calls - before == 1
        z1, ld1 = flow.bijection.forward(x.cuda())
        xr, ldr = flow.bijection.inverse(z1)
        if D == 256:                                        # the default picks the format by itself
            set_debug(monkeypatch, lean_bf16x3=None)
            flow.invalidate_native_caches()
            assert torch.equal(flow.log_prob(x.cuda()), lp1)
    e_h, e_0 = rel(lp1.cpu().numpy(), lp_h.numpy()), rel(lp1.cpu().numpy(), lp0.cpu().numpy())
    print(f"{arch}({D}) bf16 x 3 operands: log_prob vs fp64 host {e_h:.2e}, vs the fp32 operand format {e_0:.2e}")
    assert e_h < 1e-5 and e_0 < 2e-6
    assert torch.allclose(xr.cpu(), x, atol=2e-4) and torch.allclose(ld1, -ldr, atol=1e-3)


@pytest.mark.parametrize("D,n_hidden", [(64, 24), (64, 31), (128, None), (256, None), (22, 20)])
def test_spline_chain_two_hidden_tiles(D, n_hidden):
    """CouplingRQNSF chains whose conditioner is wider than 15 units -- a user-chosen n_hidden, or the default at D = 256
    (17 units) -- run as ONE single-launch spline chain with two hidden tiles (bf16 x 3 operands); D = 128 (15 units)
    takes one.  Against the host (ATen, fp32) path; one libtfk launch per log_prob."""
    import torchflows_amd as tfa
    from torchflows_amd import native
    torch.manual_seed(6)
    kw = {} if n_hidden is None else dict(conditioner_kwargs=dict(n_hidden=n_hidden))
    flow = tfa.Flow(tfa.CouplingRQNSF(D, n_layers=2, **kw))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D))
    flow.eval()
    x = torch.randn(700, D) * 1.5
    with torch.no_grad():
        lp_h = flow.log_prob(x)
        z_h, _ = flow.bijection.forward(x)
        flow = flow.cuda()
        before = native.calls
        lp_d = flow.log_prob(x.cuda())
        assert native.calls - before == 1
        z_d, ld_d = flow.bijection.forward(x.cuda())
        xr, ldr = flow.bijection.inverse(z_d)
    assert rel(lp_d.cpu().numpy(), lp_h.numpy()) < 1e-5 * max(1.0, D / 64)
    assert normwise(z_d.cpu().numpy(), z_h.numpy()) < 2e-5
    assert torch.allclose(xr.cpu(), x, atol=1e-3) and torch.allclose(ld_d, -ldr, atol=1e-3)


@pytest.mark.parametrize("D,n_hidden,n_layers", [(64, None, 8), (64, 24, 2), (128, None, 3), (22, None, 2), (16, None, 3)])
def test_linear_rational_spline_chain_is_one_launch(D, n_hidden, n_layers):
    """CouplingLRS chains (linear rational splines, TFK_OP_LRS_*_LEAN) run as ONE single-launch spline chain, one or two
    hidden tiles, padded event sizes included.  Against the host path in fp64 (the composition in double precision):
    log_prob within 1e-5; the inverse undoes the forward."""
    import copy
    import torchflows_amd as tfa
    from torchflows_amd import native
    torch.manual_seed(8)
    kw = {} if n_hidden is None else dict(conditioner_kwargs=dict(n_hidden=n_hidden))
    flow = tfa.Flow(tfa.CouplingLRS(D, n_layers=n_layers, **kw))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D))
    flow.eval()
    x = torch.randn(1500, D) * 1.5
    x[:200] *= 20.0                                        # rows with elements outside the +-50 box
    flow64 = copy.deepcopy(flow).double()
    with torch.no_grad():
        lp_h = flow64.log_prob(x.double())
        z_h, _ = flow64.bijection.forward(x.double())
        flow = flow.cuda()
        before = native.calls
        lp_d = flow.log_prob(x.cuda())
        assert native.calls - before == 1
        z_d, ld_d = flow.bijection.forward(x.cuda())
        xr, ldr = flow.bijection.inverse(z_d)
    e_lp, e_z = rel(lp_d.cpu().numpy(), lp_h.numpy()), normwise(z_d.cpu().numpy(), z_h.numpy())
    print(f"CouplingLRS({D}, hidden {n_hidden}, {n_layers} layers): log_prob {e_lp:.2e}, z nw {e_z:.2e}")
    assert e_lp < 1e-5
    assert e_z < 2e-5
    assert torch.allclose(xr.cpu(), x, atol=2e-3, rtol=1e-5) and torch.allclose(ld_d, -ldr, atol=1e-3)


@pytest.mark.parametrize("arch,D", [("RealNVP", 64), ("RealNVP", 128), ("RealNVP", 22), ("MAF", 64), ("IAF", 64)])
def test_affine_scale_logits_far_below_zero(arch, D):
    """alpha = exp(u / 2 + c0) + 1e-10 (affine.py:33-34) with logits down to -80: the chain kernels take log2(alpha) = u'
    (and 1 / alpha = 2^-u' in the inverse) only while every lane of the wave holds u' >= -8 (csrc/tfk_flow_chain.h:
    log2_scales) -- rows whose scales approach the 1e-10 floor must take the full logarithm.  Against the host path in
    fp64, in the direction(s) that run as a chain kernel (MAF: forward, IAF: inverse, RealNVP: both)."""
    import copy
    import torchflows_amd as tfa
    torch.manual_seed(9)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=4))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D))
    flow.eval()
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if name.endswith("sequential.2.bias"):          # (T, 2) interleaved (scale logit, shift)
                b = p.view(-1, 2)
                b[:, 0] += torch.linspace(-80.0, 0.0, b.shape[0])[torch.randperm(b.shape[0])]
    flow64 = copy.deepcopy(flow).double()
    flow = flow.cuda()
    if arch != "IAF":
        x = torch.randn(3000, D)
        with torch.no_grad():
            lp_h = flow64.log_prob(x.double())
            z_h, ld_h = flow64.bijection.forward(x.double())
            lp_d = flow.log_prob(x.cuda())
            z_d, ld_d = flow.bijection.forward(x.cuda())
        e_lp, e_ld = rel(lp_d.cpu().numpy(), lp_h.numpy()), rel(ld_d.cpu().numpy(), ld_h.numpy())
        print(f"{arch}({D}) scale logits down to -80: log_prob {e_lp:.2e}, log_det {e_ld:.2e}, "
              f"mean log_det {float(ld_h.mean()):.1f}")
        assert e_lp < 1e-5 and e_ld < 1e-5
        assert normwise(z_d.cpu().numpy(), z_h.numpy()) < 2e-5
    if arch != "MAF":
        # the inverse map (Flow.sample's direction; 1 / alpha up to 2e17 per layer)
        y = torch.randn(3000, D) * 1e-3
        with torch.no_grad():
            xi_h, ldi_h = flow64.bijection.inverse(y.double())
            xi_d, ldi_d = flow.bijection.inverse(y.cuda())
        e_ldi = rel(ldi_d.cpu().numpy(), ldi_h.numpy())
        ok = torch.isfinite(xi_h.float()).all(dim=1)                   # (rows that leave fp32's range are not compared)
        e_xi = normwise(xi_d.cpu()[ok].numpy(), xi_h[ok].numpy())
        print(f"{arch}({D}) inverse: log_det {e_ldi:.2e}, x nw {e_xi:.2e} on {int(ok.sum())} rows")
        assert e_ldi < 1e-5
        assert int(ok.sum()) > 100 and e_xi < 1e-4


@pytest.mark.parametrize("arch,D", [("CouplingRQNSF", 64), ("CouplingLRS", 64), ("CouplingRQNSF", 22)])
def test_spline_logits_beyond_the_fast_softmax_range(arch, D):
    """The spline chain kernels form the bin softmax WITHOUT subtracting the maximum while every logit of the workgroup's
    rows stays within +-64 / log2(e) (csrc/tfk_flow_rqs_chain.h: spline_knots); rows beyond that are re-run with the
    maximum subtracted.  Logits up to +-150 on a quarter of the elements, against the host path in fp64."""
    import copy
    import torchflows_amd as tfa
    torch.manual_seed(10)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=3))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D))
    flow.eval()
    P = 23 if arch == "CouplingRQNSF" else 32
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if name.endswith("sequential.2.bias"):
                b = p.view(-1, P)
                T = b.shape[0]
                pick = torch.randperm(T)[: max(1, T // 4)]
                b[pick, 0:8] += torch.empty(len(pick), 8).uniform_(-150.0, 150.0)
    flow64 = copy.deepcopy(flow).double()
    flow = flow.cuda()
    x = torch.randn(2000, D) * 2.0
    with torch.no_grad():
        lp_h = flow64.log_prob(x.double())
        z_h, _ = flow64.bijection.forward(x.double())
        lp_d = flow.log_prob(x.cuda())
        z_d, ld_d = flow.bijection.forward(x.cuda())
        xr, ldr = flow.bijection.inverse(z_d)
    e_lp, e_z = rel(lp_d.cpu().numpy(), lp_h.numpy()), normwise(z_d.cpu().numpy(), z_h.numpy())
    print(f"{arch}({D}) softmax logits up to +-150: log_prob {e_lp:.2e}, z nw {e_z:.2e}")
    assert torch.isfinite(lp_d).all()
    assert e_lp < 4e-5 and e_z < 4e-5
    assert torch.allclose(ld_d, -ldr, atol=2e-3)


@pytest.mark.parametrize("arch,D,N", [("RealNVP", 64, 100003), ("RealNVP", 64, 17), ("CouplingRQNSF", 64, 40001),
                                      ("RealNVP", 256, 30011), ("RealNVP", 22, 70001), ("MAF", 64, 50021),
                                      ("CouplingLRS", 16, 9001)])
def test_log_likelihood_sum_rides_in_the_log_prob_launch(arch, D, N, monkeypatch):
    """Flow.log_prob_and_sum: the fp64 sum of the log-probabilities comes out of the SAME launch (tfk_flow_run_mfma_sum:
    one partial per workgroup, the last workgroup adds them in index order and resets the workspace) -- equal to the
    fp64 sum of the returned vector to rounding, bitwise reproducible, and correct call after call.  (Opt-in:
    TORCHFLOWS_AMD_DEBUG=sum_in_kernel=1; by default log_prob_and_sum is log_prob + tfk_sum_f32.)"""
    import torchflows_amd as tfa
    from torchflows_amd import native
    set_debug(monkeypatch, sum_in_kernel="1")
    torch.manual_seed(11)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=4))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D))
    flow = flow.eval().cuda()
    x = torch.randn(N, D, device="cuda")
    with torch.no_grad():
        before = native.calls
        lp, total = flow.log_prob_and_sum(x)
        assert native.calls - before == 1                       # ONE launch: no separate reduction
        lp_ref = flow.log_prob(x)
        want = lp.double().sum()
        assert torch.equal(lp, lp_ref)
        assert total.dtype == torch.float64 and total.shape == (1,)
        assert abs(float(total) - float(want)) <= 1e-12 * abs(float(want)) + 1e-9
        for n in (N, max(1, N // 3), 1, N):                     # the workspace resets itself; other sizes in between
            lp2, t2 = flow.log_prob_and_sum(x[:n])
            w2 = lp2.double().sum()
            assert abs(float(t2) - float(w2)) <= 1e-12 * abs(float(w2)) + 1e-9
        assert float(t2) == float(total)                        # same rows, same order of additions
    # a second stream has its own workspace
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        lp3, t3 = flow.log_prob_and_sum(x)
    s.synchronize()
    assert float(t3) == float(total)


@pytest.mark.parametrize("arch,D,C,n_hidden", [("CouplingRQNSF", 64, 8, None), ("CouplingRQNSF", 64, 16, 24),
                                               ("CouplingLRS", 64, 5, None), ("CouplingRQNSF", 128, 3, None),
                                               ("CouplingRQNSF", 22, 8, None), ("RealNVP", 64, 8, None),
                                               ("NICE", 64, 3, None), ("RealNVP", 128, 16, None), ("RealNVP", 22, 5, None),
                                               ("RealNVP", 256, 8, None), ("CouplingRQNSF", 256, 4, None)])
def test_conditional_spline_chain_is_three_launches(arch, D, C, n_hidden):
    """Conditional coupling flows (log_prob(x, context=c)): [context-conditioned elementwise layer] + ONE lean chain launch
    whose GEMM 1 takes the context's columns of W1 as further k-steps + [the elementwise layers behind the chain].
    Against the host path in fp64; forward and inverse."""
    import copy
    import torchflows_amd as tfa
    from torchflows_amd import native
    torch.manual_seed(12)
    kw = {} if n_hidden is None else dict(conditioner_kwargs=dict(n_hidden=n_hidden))
    flow = tfa.Flow(getattr(tfa, arch)(D, context_shape=(C,), n_layers=4, **kw))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(1024, D), context=torch.randn(1024, C))
    flow.eval()
    x, c = torch.randn(1500, D) * 1.5, torch.randn(1500, C)
    flow64 = copy.deepcopy(flow).double()
    with torch.no_grad():
        lp_h = flow64.log_prob(x.double(), context=c.double())
        z_h, _ = flow64.bijection.forward(x.double(), context=c.double())
        flow = flow.cuda()
        before = native.calls
        lp_d = flow.log_prob(x.cuda(), context=c.cuda())
        # spline chains: [context-conditioned elementwise] + chain + [elementwise behind]; affine / shift chains take the
        # elementwise layers inside the lean launch (one launch per 8 couplings' worth of LDS)
        # (D = 256: the interpreter keeps conditional affine chains -- the context variant of the 256-wide kernel spills)
        n_launch = native.calls - before
        assert n_launch == 1 or (D == 256 and n_launch <= 3)
        z_d, ld_d = flow.bijection.forward(x.cuda(), context=c.cuda())
        xr, ldr = flow.bijection.inverse(z_d, context=c.cuda())
    e_lp, e_z = rel(lp_d.cpu().numpy(), lp_h.numpy()), normwise(z_d.cpu().numpy(), z_h.numpy())
    print(f"{arch}({D}) context {C}: log_prob {e_lp:.2e}, z nw {e_z:.2e}")
    assert e_lp < 1e-5 and e_z < 2e-5
    assert torch.allclose(xr.cpu(), x, atol=2e-3, rtol=1e-5) and torch.allclose(ld_d, -ldr, atol=1e-3)


@pytest.mark.parametrize("arch,D,n_layers", [("MaskedAutoregressiveRQNSF", 64, 4),
                                             pytest.param("InverseAutoregressiveRQNSF", 64, 3, marks=pytest.mark.slow),
                                             ("MaskedAutoregressiveLRS", 64, 3), ("MaskedAutoregressiveRQNSF", 128, 2),
                                             ("MaskedAutoregressiveRQNSF", 22, 3),
                                             pytest.param("InverseAutoregressiveLRS", 64, 2, marks=pytest.mark.slow)])
def test_made_spline_chain_is_one_launch(arch, D, n_layers):
    """The parallel map of MADE-based spline flows (MA-RQNSF / MA-LRS density, IA-* sampling direction) as ONE launch of
    the spline chain kernel (TFK_OP_MADE_{RQS,LRS}_FWD_LEAN: both planes feed GEMM 1, every element is a target).
    Against the host path in fp64."""
    import copy
    import torchflows_amd as tfa
    from torchflows_amd import native
    torch.manual_seed(13)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=n_layers))
    flow.train()
    with torch.no_grad():                                   # (Inverse* presets: log_prob is the SEQUENTIAL map on the host)
        flow.log_prob(torch.randn(1024 if arch.startswith("Masked") else 96, D))
    flow.eval()
    x = torch.randn(1500, D) * 1.5
    x[:100] *= 30.0                                         # rows with elements outside the spline box
    flow64 = copy.deepcopy(flow).double()
    parallel_is_forward = arch.startswith("Masked")
    with torch.no_grad():
        fn64 = flow64.bijection.forward if parallel_is_forward else flow64.bijection.inverse
        z_h, ld_h = fn64(x.double())
        flow = flow.cuda()
        fn = flow.bijection.forward if parallel_is_forward else flow.bijection.inverse
        before = native.calls
        z_d, ld_d = fn(x.cuda())
        assert native.calls - before <= 2      # (the chain + a permutation back to logical order after an odd number of reversals)
        if parallel_is_forward:
            lp_d = flow.log_prob(x.cuda())
            lp_h = flow64.log_prob(x.double())
            assert rel(lp_d.cpu().numpy(), lp_h.numpy()) < 1e-5
    e_z, e_ld = normwise(z_d.cpu().numpy(), z_h.numpy()), rel(ld_d.cpu().numpy(), ld_h.numpy())
    print(f"{arch}({D}, {n_layers} layers): z nw {e_z:.2e}, log_det {e_ld:.2e}")
    assert e_z < 2e-5 and e_ld < 4e-5                       # (D log-det terms per layer that largely cancel)


def test_data_edits_are_noticed_without_a_host_sync(monkeypatch):
    """``p.data.mul_(...)`` moves no version counter, so a compiled program would serve the old weights
    for ever.  Default guard (fused._Guard): every GUARD_EVERY-th hit enqueues a device-side checksum comparison whose
    verdict is read on a later hit -- no synchronisation on the hit path; within ~2 GUARD_EVERY calls the stale program is
    dropped (StaleProgramWarning) and the results are those of the live weights.  RealNVP-64 (flow program) and an image
    flow (image program)."""
    import warnings
    import torchflows_amd as tfa
    from torchflows_amd import fused
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    monkeypatch.setattr(fused, "GUARD_EVERY", 4)
    for make, shape in ((lambda: tfa.RealNVP(64, n_layers=2), (64,)), (lambda: AffineGlow((3, 8, 8), n_layers=1), (3, 8, 8))):
        torch.manual_seed(2)
        flow = tfa.Flow(make())
        flow.train()
        with torch.no_grad():
            flow.log_prob(torch.randn(256, *shape))
        flow = flow.eval().cuda()
        x = torch.randn(64, *shape, device="cuda")
        with torch.no_grad():
            before = flow.log_prob(x).clone()
            lin = [m for m in flow.modules() if isinstance(m, torch.nn.Linear)][-1]
            lin.bias.data.add_(0.05)                                   # no version counter moves
            ref = None
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                seen = None
                for i in range(40):
                    lp = flow.log_prob(x)
                    torch.cuda.synchronize()
                    if not torch.equal(lp, before):
                        seen = i
                        break
            assert seen is not None and seen <= 3 * 4 + 2, seen
            assert any(issubclass(m.category, fused.StaleProgramWarning) for m in w)
            flow.invalidate_native_caches()
            ref = flow.log_prob(x)
            assert torch.equal(lp, ref)
