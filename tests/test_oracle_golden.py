"""Pin the CPU oracle (oracle/) to the real reference through the golden fixtures.

The fixtures in tests/golden/ are outputs of davidnabergoj/torchflows v1.2.0 run
in the build container (tests/golden/make_golden.py).  The reference has no
golden vectors of its own for this path, only the
closed-form Gaussian check, which gauss.npz repeats.

Tolerances (fp32, relative to max(1, |ref|)):
  * integer work (masks, permutations, spline bin index): bit-exact;
  * affine / RealNVP / NICE: 5e-6 (measured 1e-7..1e-6; reference fp32-vs-fp64
    floor is 2e-7..7e-7);
  * RQ spline: 4e-5 elementwise, 2e-5 norm-wise.  The reference's own
    fp32-vs-fp64 distance on these fixtures is up to 1.3e-5 (knots are
    100*cumsum(softmax)-50, 1 ulp = 3.8e-6, amplified by 1/bin_width <= 10),
    and ATen's CPU softmax uses a reduced-accuracy vector exp, so agreement
    tighter than the floor is not defined.  Each test prints the floor next to
    the error it measured.
"""
import numpy as np
import pytest

from conftest import load_golden, state_dict_of


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def normwise(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ------------------------------------------------------------------ integers
def test_halfsplit_and_reverse_permutation_bit_exact(oracle):
    fx = load_golden("masks.npz")
    for tag in fx["shapes"]:
        D = int(np.prod([int(t) for t in str(tag).split("x")]))
        s, t = oracle.halfsplit_mask(D)
        assert np.array_equal(s, fx[f"src_{tag}"].reshape(-1).astype(bool))
        assert np.array_equal(t, fx[f"tgt_{tag}"].reshape(-1).astype(bool))
        assert int(s.sum()) == int(fx[f"S_{tag}"]) and int(t.sum()) == int(fx[f"T_{tag}"])
        src, tgt = oracle.mask_to_index(s), oracle.mask_to_index(t)
        assert np.array_equal(src, np.nonzero(fx[f"src_{tag}"].reshape(-1))[0])
        assert np.array_equal(tgt, np.nonzero(fx[f"tgt_{tag}"].reshape(-1))[0])
        f, i = oracle.reverse_permutation(D)
        assert np.array_equal(f, fx[f"pfwd_{tag}"])
        assert np.array_equal(i, fx[f"pinv_{tag}"])


# ------------------------------------------------------------------ affine
@pytest.mark.parametrize("T", [2, 32, 128])
def test_affine_golden(oracle, T):
    fx = load_golden("affine.npz")
    x, h = fx[f"T{T}_x"], fx[f"T{T}_h"]
    z, ld = oracle.affine(x, h)
    xi, ldi = oracle.affine(x, h, inverse=True)
    assert rel(z, fx[f"T{T}_z"]) < 5e-6
    assert rel(ld, fx[f"T{T}_ld"]) < 5e-6
    assert rel(xi, fx[f"T{T}_xinv"]) < 5e-6
    assert rel(ldi, fx[f"T{T}_ldinv"]) < 5e-6
    # round trip and sign convention of the reference's own tests
    # (test/test_reconstruction_transformers.py:31-58)
    xr, ldr = oracle.affine(z, h, inverse=True)
    ok = np.isfinite(z).all(axis=1)
    assert rel(xr[ok][3:], x[ok][3:]) < 1e-3
    assert rel(ldr, -ld) < 1e-6


# ------------------------------------------------------------------ spline
def _rqs_cases():
    fx = load_golden("rqs.npz")
    return [str(c) for c in fx["cases"]]


@pytest.mark.parametrize("tag", _rqs_cases())
def test_rqs_golden(oracle, tag):
    fx = load_golden("rqs.npz")
    K = int(tag.split("K")[1])
    B = float(tag.split("_")[0][1:])
    x, h = fx[f"{tag}_x"], fx[f"{tag}_h"]
    for inverse, zk, lk, lek, kk, knots in ((False, "z", "ld", "ld_el", "k", "bin_x"),
                                            (True, "xinv", "ldinv", "ldinv_el", "kinv", "bin_y")):
        out, ld, ld_el, k = oracle.rqs(x, h, K, B, inverse=inverse)
        assert rel(out, fx[f"{tag}_{zk}"]) < 4e-5
        assert rel(ld_el, fx[f"{tag}_{lek}"]) < 4e-5
        assert rel(ld, fx[f"{tag}_{lk}"]) < 4e-5
        # bin index: bit-exact wherever the input is not within 4 ulp of a knot of
        # the reference (there a 1-ulp knot difference legitimately moves the bin)
        kn = fx[f"{tag}_{knots}"]
        gap = np.min(np.abs(kn - x[..., None]), axis=-1)
        safe = gap > 4 * np.spacing(np.float32(B))
        assert np.array_equal(k[safe], fx[f"{tag}_{kk}"][safe])
        assert (~safe).mean() < 0.05
        # outside the strict box: identity, zero log-det, k = -1 (spline/base.py:29-33)
        outside = ~((x > -B) & (x < B))
        assert np.array_equal(out[outside], x[outside])
        assert np.all(ld_el[outside] == 0) and np.all(k[outside] == -1)
        assert outside.any()


def test_rqs_exact_knot_goes_to_left_bin(oracle):
    """searchsorted(..., right=False) - 1: an input equal to an interior knot lands in
    the bin on its left (rational_quadratic.py:82,147).  Uses the oracle's own knots."""
    rng = np.random.default_rng(0)
    K, B, T = 8, 50.0, 7
    h = rng.standard_normal((5, T, 3 * K - 1)).astype(np.float32)
    bx, by, _ = oracle.rqs_knots(h, K, B)
    for j in range(1, K):
        _, _, _, k = oracle.rqs(bx[..., j], h, K, B)
        assert np.all(k == j - 1)
        _, _, _, k = oracle.rqs(by[..., j], h, K, B, inverse=True)
        assert np.all(k == j - 1)
    # boundary knots are outside the strict box
    out, _, ld_el, k = oracle.rqs(bx[..., 0], h, K, B)
    assert np.all(k == -1) and np.all(ld_el == 0) and np.array_equal(out, bx[..., 0])


def test_rqs_roundtrip_grid(oracle):
    """The reference's own property (test/test_spline.py:30-136): inverse(forward(x)) = x,
    ld_f = -ld_i, over boundaries, bin counts and input scales."""
    rng = np.random.default_rng(1)
    for B in (1.0, 5.0, 50.0):
        for K in (2, 4, 8, 16, 32):
            for scale in (1e-2, 1.0, 1e1, 1e2):
                x = (rng.standard_normal((16, 5)) * scale).astype(np.float32)
                h = rng.standard_normal((16, 5, 3 * K - 1)).astype(np.float32)
                z, ld, _, _ = oracle.rqs(x, h, K, B)
                xr, ldi, _, _ = oracle.rqs(z, h, K, B, inverse=True)
                assert np.all(np.isfinite(z)) and np.all(np.isfinite(ld))
                assert np.max(np.abs(xr - x)) < 1e-3 * max(1.0, scale)
                assert np.max(np.abs(ld + ldi)) < 1e-3


# ------------------------------------------------------------------ gaussian
def test_diag_gauss_golden(oracle):
    fx = load_golden("gauss.npz")
    for D in (3, 64):
        lp = oracle.diag_gauss_logprob(fx[f"D{D}_value"], fx[f"D{D}_loc"], fx[f"D{D}_log_scale"])
        assert rel(lp, fx[f"D{D}_log_prob"]) < 2e-6
    v = fx["std2_value"]
    lp = oracle.diag_gauss_logprob(v, np.zeros(2), np.zeros(2))
    closed = -0.5 * (v.astype(np.float64) ** 2).sum(-1) - np.log(2 * np.pi)
    assert rel(lp, fx["std2_log_prob"]) < 1e-6 and rel(lp, closed) < 1e-6


def test_actnorm_init_golden(oracle):
    fx = load_golden("layers.npz")
    for tag in ("n100", "n1"):
        v = oracle.actnorm_init(fx[f"actnorm_{tag}_x"])
        assert rel(v, fx[f"actnorm_{tag}_value"]) < 5e-6


# ------------------------------------------------------------------ whole flows
FLOWS = [
    # fixture, arch, n_layers, context size, is-spline
    ("flow_realnvp3.npz", "RealNVP", 2, 0, False),
    ("flow_realnvp64.npz", "RealNVP", 8, 0, False),
    ("flow_nsf64.npz", "CouplingRQNSF", 8, 0, True),
    ("flow_realnvp256.npz", "RealNVP", 8, 0, False),
    ("flow_nice7.npz", "NICE", 2, 0, False),
    ("flow_realnvp_7x11.npz", "RealNVP", 2, 0, False),
    ("flow_realnvp5_ctx3.npz", "RealNVP", 2, 3, False),
    ("flow_nsf6_ctx2.npz", "CouplingRQNSF", 2, 2, True),
    ("flow_nsf_3x5x2.npz", "CouplingRQNSF", 2, 0, True),
    ("flow_lrs16.npz", "CouplingLRS", 3, 0, True),
    ("flow_maf6.npz", "MAF", 2, 0, False),
    ("flow_iaf6.npz", "IAF", 2, 0, False),
    ("flow_marqnsf5.npz", "MaskedAutoregressiveRQNSF", 2, 0, True),
    ("flow_iarqnsf5.npz", "InverseAutoregressiveRQNSF", 2, 0, True),
    ("flow_malrs5.npz", "MaskedAutoregressiveLRS", 2, 0, True),
]


@pytest.mark.parametrize("variant", ["fresh", "init"])
@pytest.mark.parametrize("name,arch,n_layers,C,spline", FLOWS)
def test_flow_golden(oracle, name, arch, n_layers, C, spline, variant):
    fx = load_golden(name)
    D = int(np.prod(fx["event_shape"]))
    assert int(fx["n_bijection_layers"]) == 3 * n_layers + 3
    flow = oracle.preset_from_state_dict(arch, D, n_layers, state_dict_of(fx, variant),
                                         context_size=C)
    ctx = fx["context"] if C else None
    x = fx["x"].reshape(-1, D)
    z_in = fx["z_in"].reshape(-1, D)
    g = lambda k: fx[f"{variant}/{k}"]

    z, ld, tz, tl = flow.forward(x, ctx, trace=True)
    lp = flow.log_prob(x, ctx)
    xi, ldi = flow.inverse(z_in, ctx)
    _, slp = flow.sample_log_prob(z_in, ctx)

    tol_el, tol_nw = (4e-5, 2e-5) if spline else (5e-6, 2e-6)
    # log-dets are signed sums of 3L+3 layer terms that largely cancel (fresh weights:
    # sum |ld_i| ~ 1e2 against a result of a few units), so their error is bounded
    # relative to that conditioning, not to the result: 1e-5 (the north-star bound)
    tol_ld = 4e-5 if spline else 1e-5
    floor = rel(g("log_prob"), g("log_prob64"))
    err = rel(lp, g("log_prob"))
    print(f"{name} {variant}: log_prob rel err {err:.2e}, reference fp32-vs-fp64 floor {floor:.2e}")
    assert err < max(tol_el, 3 * floor)
    assert rel(ld, g("log_det")) < max(tol_ld, 3 * rel(g("log_det"), g("log_det64")))
    assert normwise(z, g("z").reshape(-1, D)) < tol_nw
    assert normwise(xi, g("x_inv").reshape(-1, D)) < max(tol_nw, 3 * normwise(g("x_inv"), g("x_inv64")))
    assert rel(ldi, g("log_det_inv")) < max(tol_ld, 3 * rel(g("log_det_inv"), g("log_det_inv64")))
    # Flow.sample(return_log_prob=True) convention: log p(z) + log|dx/dz|  (flows.py:710-712)
    assert rel(slp, g("sample_log_prob")) < max(tol_ld, 3 * rel(g("log_det_inv"), g("log_det_inv64")))
    # per-layer trace, layer order of BijectiveComposition.forward (bijections/base.py:211-222)
    nt = g("trace_z").shape[1]
    assert normwise(tz[:, :nt], g("trace_z")) < tol_nw
    assert rel(tl[:, :nt], g("trace_ld")) < tol_el
    # permutation layers have exactly zero log-det
    for i, t in enumerate(fx["layer_types"]):
        if str(t) == "ReversePermutationMatrix":
            assert np.all(tl[i] == 0)


# ------------------------------------------------------------------ image path (config 5)
def test_image_masks_and_squeeze_bit_exact(oracle):
    fx = load_golden("image_masks.npz")
    for tag in fx["shapes"]:
        es = tuple(int(t) for t in str(tag).split("x"))
        for inv in (0, 1):
            s, t = oracle.image_mask("checkerboard", es, bool(inv))
            assert np.array_equal(s, fx[f"ckb{inv}_src_{tag}"].astype(bool)) and np.array_equal(t, ~s)
            if es[0] > 1:
                s, t = oracle.image_mask("channel_wise", es, bool(inv))
                assert np.array_equal(s, fx[f"chw{inv}_src_{tag}"].astype(bool)) and np.array_equal(t, ~s)
        assert np.array_equal(oracle.squeeze_index(es), fx[f"squeeze_fwd_{tag}"])


def test_conv1x1_golden(oracle):
    fx = load_golden("image_layers.npz")
    for key in ("lu1", "lu2", "lu3", "lu6", "lu12", "conv3", "conv6"):
        x, h = fx[f"{key}_x"], fx[f"{key}_h"]
        y, ld = oracle.conv1x1(x, h)
        xi, ldi = oracle.conv1x1(x, h, inverse=True)
        assert rel(y, fx[f"{key}_y"]) < 2e-6 and rel(ld, fx[f"{key}_ld"]) < 2e-6
        assert rel(xi, fx[f"{key}_xinv"]) < 2e-6 and rel(ldi, fx[f"{key}_ldinv"]) < 2e-6
        # round trip (reference test/test_lu_matrix_transformer.py:7-30)
        xr, ldr = oracle.conv1x1(y, h, inverse=True)
        assert rel(xr, x) < 1e-4 and rel(ldr, -ld) < 1e-6
    # the log-det does not scale with the number of pixels (reference quirk Q9)
    x, h = fx["conv6_x"], fx["conv6_h"]
    _, ld_img = oracle.conv1x1(x, h)
    _, ld_vec = oracle.conv1x1(x[:, :, 0, 0], h)
    assert np.array_equal(ld_img, ld_vec)


# ------------------------------------------------------------------ linear rational spline (8f-4)
def test_lrs_golden(oracle):
    """orc_lrs_fwd / _inv against LinearRational.forward / inverse of the reference
    (tests/golden/lrs.npz), at the reference's own fp32-vs-fp64 distance."""
    fx = load_golden("lrs.npz")
    for tag in fx["cases"]:
        B = float(str(tag).split("_")[0][1:])
        K = int(str(tag).split("K")[1])
        x, h = fx[f"{tag}_x"], fx[f"{tag}_h"]
        z, ld = oracle.lrs(x, h, K, B)
        xi, ldi = oracle.lrs(x, h, K, B, inverse=True)
        for mine, key in ((z, "z"), (ld, "ld"), (xi, "xinv"), (ldi, "ldinv")):
            e = rel(mine, fx[f"{tag}_{key}64"])
            floor = rel(fx[f"{tag}_{key}"], fx[f"{tag}_{key}64"])
            assert e < max(1e-5, 2 * floor), (tag, key, e, floor)
        outside = np.abs(x) >= B                       # strict box: identity, zero log-det
        assert np.array_equal(z[outside], x[outside]) and np.array_equal(xi[outside], x[outside])
