"""HIP kernels vs the CPU oracle and the reference's golden vectors, through the C-ABI.

Every call below goes ``torchflows_amd.native`` -> ctypes -> ``libtfk.so`` (include/tfk.h).
Bars: integer / index work bit-exact; fp32 transforms within 1e-5 relative
(``|a-b| / max(1,|b|)``) of the oracle for the affine family (measured ~3e-7).
For the RQ spline the bar is 4e-5 against both the oracle and the golden vectors: the
knots are 2B*cumsum(softmax)-B, so at B = 50 one ulp of a knot is 3.8e-6 and
xi = (x - knot)/width amplifies it by 1/width (width >= 0.1); device expf (ocml, <= 1 ulp)
and glibc expf differ in the last bit on ~1/3 of the arguments, which is enough to move
z by 1-2e-5 on narrow bins (measured: 1.9e-5 max at B = 50, ~1e-6 at B = 5 with O(1)
parameters).  The reference's own fp32-vs-fp64 distance is 1.3e-5 and its CPU softmax
uses a reduced-accuracy vector exp, so nothing tighter is defined (tests/test_oracle_golden.py).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.fixture(scope="module")
def native():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from torchflows_amd import native as nat
    nat.lib()          # raises if libtfk.so is missing -- there is no fallback
    name, cus = nat.device_info()
    print("device:", name, cus)
    return nat


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def full_row_case(rng, N, D, T, masked):
    """rows (N, D) + target index list (None = contiguous tail)."""
    x = (rng.standard_normal((N, D)) * 2).astype(np.float32)
    if masked:
        tgt = np.sort(rng.choice(D, size=T, replace=False)).astype(np.int32)
    else:
        tgt = np.arange(D - T, D, dtype=np.int32)
    return x, tgt


SHAPES = [  # (N, D, T)
    (1000, 64, 32),     # RealNVP D=64: vectorised HalfSplit kernel
    (513, 256, 128),    # config 4 shape
    (257, 3, 2),        # config 1 shape: generic kernel
    (100, 77, 39),      # (7, 11) event
    (64, 3072, 1536),   # image-sized rows (G = 64 lanes, looped)
    (1, 8, 4),
]


@pytest.mark.parametrize("N,D,T", SHAPES)
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_coupling_vs_oracle(native, oracle, N, D, T, masked, inverse):
    rng = np.random.default_rng(N + D + T + masked)
    x, tgt = full_row_case(rng, N, D, T, masked)
    h = rng.standard_normal((N, T, 2)).astype(np.float32)
    h[0, :, 0] = 30.0 if N > 1 else 1.0
    zb, ld = oracle.affine(x[:, tgt], h, inverse=inverse)
    expect = x.copy()
    expect[:, tgt] = zb

    xd, hd = dev(x), dev(h)
    tgt_d = dev(tgt, torch.int32) if masked else None
    # out of place, overwrite
    out = torch.full_like(xd, float("nan"))
    logdet = torch.full((N,), float("nan"), device="cuda")
    native.affine_coupling(xd, hd, out, logdet, tgt_d, T, accumulate=False, inverse=inverse)
    # the row log-det is a sum of T O(1) terms that largely cancel; its rounding error
    # (the oracle sums sequentially, the kernel per lane then by a shuffle tree) scales with
    # the number of terms, so the bound is 1e-5 per 32 terms (T = 32 for D = 64)
    ld_tol = 1e-5 * max(1.0, T / 32)
    assert rel(out.cpu().numpy(), expect) < 1e-5
    assert rel(logdet.cpu().numpy(), ld) < ld_tol
    assert torch.equal(xd.cpu(), torch.from_numpy(x)), "input was mutated"
    untouched = np.setdiff1d(np.arange(D), tgt)
    assert np.array_equal(out.cpu().numpy()[:, untouched], x[:, untouched])   # bit-exact copy
    # in place + accumulate on top of a running log-det
    run = rng.standard_normal(N).astype(np.float32)
    logdet2 = dev(run)
    buf = xd.clone()
    native.affine_coupling(buf, hd, buf, logdet2, tgt_d, T, accumulate=True, inverse=inverse)
    assert torch.equal(buf, out)
    assert rel(logdet2.cpu().numpy(), run + ld) < ld_tol


@pytest.mark.parametrize("N,D,T", [(300, 7, 4), (64, 64, 32)])
@pytest.mark.parametrize("inverse", [False, True])
def test_shift_coupling(native, N, D, T, inverse):
    rng = np.random.default_rng(5)
    x, tgt = full_row_case(rng, N, D, T, False)
    h = rng.standard_normal((N, T, 1)).astype(np.float32)
    expect = x.copy()
    expect[:, tgt] = x[:, tgt] - h[..., 0] if inverse else x[:, tgt] + h[..., 0]
    out = torch.empty(N, D, device="cuda")
    logdet = torch.full((N,), 7.0, device="cuda")
    native.shift_coupling(dev(x), dev(h), out, logdet, None, T, accumulate=False, inverse=inverse)
    assert np.array_equal(out.cpu().numpy(), expect)          # one fp32 add: bit-exact
    assert torch.all(logdet == 0)
    native.shift_coupling(dev(x), dev(h), out, None, None, T, accumulate=True, inverse=inverse)


RQS_SHAPES = [  # (N, D, T, K, boundary)
    (1024, 64, 32, 8, 50.0),    # NSF D=64: shuffle reduce, 8 rows per LDS tile
    (1001, 64, 32, 8, 5.0),     # ragged last tile
    (300, 3, 2, 8, 50.0),
    (77, 30, 15, 8, 50.0),      # T not a power of two: LDS reduce
    (9, 600, 300, 8, 50.0),     # T > 256: chunked rows
    (200, 16, 8, 4, 5.0),       # compile-time K = 4
    (200, 16, 8, 5, 3.0),       # run-time K (streamed from LDS)
    (50, 8, 4, 16, 50.0),
    (3, 512, 256, 8, 50.0),
]


@pytest.mark.parametrize("N,D,T,K,B", RQS_SHAPES)
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_coupling_vs_oracle(native, oracle, N, D, T, K, B, masked, inverse):
    rng = np.random.default_rng(N + D + K + masked)
    x, tgt = full_row_case(rng, N, D, T, masked)
    x *= B / 4
    x[0, tgt[0]] = B                       # on the edge: identity
    x[-1, tgt[-1]] = -B * 1.5              # outside
    h = rng.standard_normal((N, T, 3 * K - 1)).astype(np.float32)
    h[N // 2] *= 5.0
    zb, ld, ld_el, k = oracle.rqs(x[:, tgt], h, K, B, inverse=inverse)
    expect = x.copy()
    expect[:, tgt] = zb

    xd, hd = dev(x), dev(h)
    tgt_d = dev(tgt, torch.int32) if masked else None
    out = torch.full_like(xd, float("nan"))
    logdet = torch.full((N,), float("nan"), device="cuda")
    native.rqs_coupling(xd, hd, out, logdet, tgt_d, T, K, B, accumulate=False, inverse=inverse)
    got = out.cpu().numpy()
    tol = 4e-5
    print(f"rqs N={N} T={T} K={K} B={B}: out err {rel(got, expect):.2e}, "
          f"logdet err {rel(logdet.cpu().numpy(), ld):.2e}")
    assert rel(got, expect) < tol, rel(got, expect)
    assert rel(logdet.cpu().numpy(), ld) < tol * max(1.0, T / 32)
    outside = ~((x[:, tgt] > -B) & (x[:, tgt] < B))
    assert np.array_equal(got[:, tgt][outside], x[:, tgt][outside])       # identity, bit-exact
    untouched = np.setdiff1d(np.arange(D), tgt)
    assert np.array_equal(got[:, untouched], x[:, untouched])
    assert torch.equal(xd.cpu(), torch.from_numpy(x))
    # in place + accumulate
    run = rng.standard_normal(N).astype(np.float32)
    logdet2 = dev(run)
    buf = xd.clone()
    native.rqs_coupling(buf, hd, buf, logdet2, tgt_d, T, K, B, accumulate=True, inverse=inverse)
    assert torch.equal(buf, out)
    assert rel(logdet2.cpu().numpy(), run + ld) < tol * max(1.0, T / 32)


def test_rqs_exact_knot_goes_left_and_bin_choice(native, oracle):
    """Inputs exactly on the oracle's interior knots must land in the left bin
    (searchsorted right=False): the kernel must then agree with the oracle, whose bin
    index test_oracle_golden pins.  Parity of outputs at knots implies the same bin."""
    rng = np.random.default_rng(3)
    K, B, T, N = 8, 50.0, 32, 64
    h = rng.standard_normal((N, T, 3 * K - 1)).astype(np.float32)
    bx, by, _ = oracle.rqs_knots(h, K, B)
    for inverse, kn in ((False, bx), (True, by)):
        for j in range(1, K):
            x = np.ascontiguousarray(kn[..., j])
            zb, ld, _, k = oracle.rqs(x, h, K, B, inverse=inverse)
            assert np.all(k == j - 1)
            out = torch.empty(N, T, device="cuda")
            logdet = torch.empty(N, device="cuda")
            native.rqs_coupling(dev(x), dev(h), out, logdet, None, T, K, B, inverse=inverse)
            assert rel(out.cpu().numpy(), zb) < 4e-5
            assert rel(logdet.cpu().numpy(), ld) < 4e-5


@pytest.mark.parametrize("T", [2, 32, 128])
def test_affine_golden_vectors(native, T):
    fx = load_golden("affine.npz")
    x, h = fx[f"T{T}_x"], fx[f"T{T}_h"]
    N = x.shape[0]
    for inverse, zk, lk in ((False, "z", "ld"), (True, "xinv", "ldinv")):
        out = torch.empty(N, T, device="cuda")
        logdet = torch.empty(N, device="cuda")
        native.affine_coupling(dev(x), dev(h), out, logdet, None, T, inverse=inverse)
        assert rel(out.cpu().numpy(), fx[f"T{T}_{zk}"]) < 1e-5
        assert rel(logdet.cpu().numpy(), fx[f"T{T}_{lk}"]) < 1e-5


def test_rqs_golden_vectors(native):
    fx = load_golden("rqs.npz")
    for tag in fx["cases"]:
        tag = str(tag)
        K = int(tag.split("K")[1])
        B = float(tag.split("_")[0][1:])
        x, h = fx[f"{tag}_x"], fx[f"{tag}_h"]
        N, T = x.shape
        for inverse, zk, lk in ((False, "z", "ld"), (True, "xinv", "ldinv")):
            out = torch.empty(N, T, device="cuda")
            logdet = torch.empty(N, device="cuda")
            native.rqs_coupling(dev(x), dev(h), out, logdet, None, T, K, B, inverse=inverse)
            assert rel(out.cpu().numpy(), fx[f"{tag}_{zk}"]) < 4e-5      # reference noise floor
            assert rel(logdet.cpu().numpy(), fx[f"{tag}_{lk}"]) < 4e-5


@pytest.mark.parametrize("N,D", [(1000, 64), (77, 3), (10, 77), (33, 3072), (1, 1)])
@pytest.mark.parametrize("inverse_affine", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_elementwise_affine(native, oracle, N, D, inverse_affine, inverse):
    rng = np.random.default_rng(D)
    x = rng.standard_normal((N, D)).astype(np.float32) * 3
    value = rng.standard_normal((D, 2)).astype(np.float32)
    h = np.broadcast_to(value, (N, D, 2)).copy()
    z_ref, ld_ref = oracle.affine(x, h, inverse=(inverse != inverse_affine))
    out = torch.empty(N, D, device="cuda")
    run = rng.standard_normal(N).astype(np.float32)
    logdet = dev(run)
    native.elementwise_affine(dev(x), dev(value), out, logdet, inverse_affine, accumulate=True,
                              inverse=inverse)
    assert rel(out.cpu().numpy(), z_ref) < 1e-5
    assert rel(logdet.cpu().numpy(), run + ld_ref) < 1e-5 * max(1.0, D / 64)
    buf = dev(x)
    native.elementwise_affine(buf, dev(value), buf, logdet, inverse_affine, accumulate=False,
                              inverse=inverse)
    assert torch.equal(buf, out)
    assert rel(logdet.cpu().numpy(), ld_ref) < 1e-5 * max(1.0, D / 64)


@pytest.mark.parametrize("N,D", [(1000, 64), (5, 3), (17, 77), (3, 3072)])
def test_permute_bit_exact(native, oracle, N, D):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((N, D)).astype(np.float32)
    fwd, inv = oracle.reverse_permutation(D)
    out = torch.empty(N, D, device="cuda")
    native.permute(dev(x), None, out)
    assert np.array_equal(out.cpu().numpy(), x[:, fwd])
    perm = rng.permutation(D).astype(np.int32)
    native.permute(dev(x), dev(perm, torch.int32), out)
    assert np.array_equal(out.cpu().numpy(), x[:, perm])
    with pytest.raises(native.NativeError):
        xd = dev(x)
        native.permute(xd, None, xd)


def test_diag_gauss_and_sum(native, oracle):
    fx = load_golden("gauss.npz")
    for D in (3, 64):
        v = fx[f"D{D}_value"]
        out = torch.empty(v.shape[0], device="cuda")
        native.diag_gauss_logprob(dev(v), dev(fx[f"D{D}_loc"]), dev(fx[f"D{D}_log_scale"]), None, out)
        assert rel(out.cpu().numpy(), fx[f"D{D}_log_prob"]) < 1e-5
        ld = np.linspace(-3, 3, v.shape[0]).astype(np.float32)
        native.diag_gauss_logprob(dev(v), dev(fx[f"D{D}_loc"]), dev(fx[f"D{D}_log_scale"]), dev(ld), out)
        assert rel(out.cpu().numpy(), fx[f"D{D}_log_prob"] + ld) < 1e-5
    rng = np.random.default_rng(0)
    for n in (1, 255, 256, 100003, 1 << 20):
        a = rng.standard_normal(n).astype(np.float32) * 100
        s = native.sum_f32(dev(a))
        assert s.dtype == torch.float64
        assert abs(float(s.item()) - float(a.astype(np.float64).sum())) <= 1e-9 * max(1.0, np.abs(a).sum())
    s1 = native.sum_f32(dev(a)).item()
    assert s1 == native.sum_f32(dev(a)).item()        # deterministic


def test_empty_and_errors(native):
    e = torch.empty(0, 64, device="cuda")
    h = torch.empty(0, 32, 2, device="cuda")
    ld = torch.empty(0, device="cuda")
    native.affine_coupling(e, h, torch.empty_like(e), ld, None, 32)           # N == 0: no-op
    native.rqs_coupling(e, torch.empty(0, 32, 23, device="cuda"), torch.empty_like(e), ld, None, 32, 8, 50.0)
    native.elementwise_affine(e, torch.zeros(64, 2, device="cuda"), torch.empty_like(e), ld, False)
    x = torch.zeros(4, 64, device="cuda")
    with pytest.raises(native.NativeError):       # wrong dtype
        native.affine_coupling(x.double(), torch.zeros(4, 32, 2, device="cuda"), x, torch.zeros(4, device="cuda"), None, 32)
    with pytest.raises(native.NativeError):       # host tensor
        native.affine_coupling(x.cpu(), torch.zeros(4, 32, 2), x.cpu(), torch.zeros(4), None, 32)
    with pytest.raises(native.NativeError):       # h size mismatch
        native.affine_coupling(x, torch.zeros(4, 31, 2, device="cuda"), x, torch.zeros(4, device="cuda"), None, 32)
    with pytest.raises(native.NativeError):       # K out of range is rejected by the library
        native.rqs_coupling(x, torch.zeros(4, 32, 3 * 40 - 1, device="cuda"), x.clone(), torch.zeros(4, device="cuda"), None, 32, 40, 50.0)
    with pytest.raises(native.NativeError):       # non-contiguous
        native.permute(torch.zeros(4, 128, device="cuda")[:, ::2], None, torch.zeros(4, 64, device="cuda"))


# ------------------------------------------------------------------ linear rational spline (8f-4)
@pytest.mark.parametrize("N,D,T", [(1000, 64, 32), (257, 3, 2), (100, 77, 39), (8, 1024, 512), (1, 8, 4)])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("n_bins", [8, 4])
def test_lrs_coupling_vs_oracle(native, oracle, N, D, T, masked, inverse, n_bins):
    """tfk_lrs_coupling_* vs orc_lrs_* on seeded inputs (box +-5, O(1) parameters): 2e-5 relative
    (device expf / logf differ from glibc's in the last bit; 1/bin-width amplifies, as for RQS)."""
    rng = np.random.default_rng(N + D + 3 * masked + inverse + n_bins)
    x, tgt = full_row_case(rng, N, D, T, masked)
    x[:3] *= 10                                          # leave the box
    h = rng.standard_normal((N, T, 4 * n_bins)).astype(np.float32)
    out = torch.empty(N, D, device="cuda")
    ld = torch.empty(N, device="cuda")
    native.lrs_coupling(dev(x), dev(h), out, ld, dev(tgt, torch.int32) if masked else None, T, n_bins, 5.0,
                        inverse=inverse)
    z_o, ld_o = oracle.lrs(x[:, tgt], h, n_bins, 5.0, inverse=inverse)
    expect = x.copy()
    expect[:, tgt] = z_o
    e_z, e_l = rel(out.cpu().numpy(), expect), rel(ld.cpu().numpy(), ld_o)
    assert e_z < 2e-5 and e_l < 2e-5 * max(1.0, T / 32), (e_z, e_l)
    keep = np.ones(D, bool)
    keep[tgt] = False
    assert np.array_equal(out.cpu().numpy()[:, keep], x[:, keep])
    # in place + accumulate
    buf = dev(x).clone()
    acc = torch.full((N,), 2.5, device="cuda")
    native.lrs_coupling(buf, dev(h), buf, acc, dev(tgt, torch.int32) if masked else None, T, n_bins, 5.0,
                        accumulate=True, inverse=inverse)
    assert torch.equal(buf, out) and torch.allclose(acc, ld + 2.5, rtol=1e-6, atol=1e-5)


def test_lrs_golden_on_hip(native):
    fx = load_golden("lrs.npz")
    for tag in fx["cases"]:
        B = float(str(tag).split("_")[0][1:])
        K = int(str(tag).split("K")[1])
        x, h = fx[f"{tag}_x"], fx[f"{tag}_h"]
        N, T = x.shape
        for inverse, zk, lk in ((False, "z", "ld"), (True, "xinv", "ldinv")):
            out = torch.empty(N, T, device="cuda")
            ld = torch.empty(N, device="cuda")
            native.lrs_coupling(dev(x), dev(h), out, ld, None, T, K, B, inverse=inverse)
            e_z = rel(out.cpu().numpy(), fx[f"{tag}_{zk}64"])
            e_l = rel(ld.cpu().numpy(), fx[f"{tag}_{lk}64"])
            f_z = rel(fx[f"{tag}_{zk}"], fx[f"{tag}_{zk}64"])
            f_l = rel(fx[f"{tag}_{lk}"], fx[f"{tag}_{lk}64"])
            print(f"lrs {tag} inverse={inverse}: out {e_z:.2e} (floor {f_z:.2e}), log-det {e_l:.2e} (floor {f_l:.2e})")
            assert e_z < max(1e-5, 3 * f_z) and e_l < max(1e-5, 3 * f_l)


def test_same_abi_two_libraries(oracle):
    """libtfk.so (device pointers, a HIP stream) and the CPU restatement (host pointers, stream NULL) export
    the SAME tfk_* symbols: one ctypes call sequence, run against both, must agree."""
    import ctypes as C
    from torchflows_amd import native
    native.lib()
    gpu, cpu = C.CDLL(native.LIB_PATH), C.CDLL(oracle.build())
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
    sigs = {"tfk_affine_coupling_fwd": [vp, vp, vp, vp, i64, i32, vp, i32, i32, vp],
            "tfk_rqs_coupling_inv": [vp, vp, vp, vp, i64, i32, vp, i32, i32, f32, i32, vp],
            "tfk_diag_gauss_logprob": [vp, vp, vp, vp, vp, i64, i32, vp], "tfk_sum_f32": [vp, vp, i64, vp]}
    for L in (gpu, cpu):
        for n, a in sigs.items():
            getattr(L, n).argtypes = a
    rng = np.random.default_rng(3)
    N, D, T, K = 777, 24, 12, 8
    x = rng.standard_normal((N, D)).astype(np.float32)
    h2 = rng.standard_normal((N, T, 2)).astype(np.float32)
    h23 = rng.standard_normal((N, T, 3 * K - 1)).astype(np.float32)
    loc, ls = rng.standard_normal(D).astype(np.float32), (0.1 * rng.standard_normal(D)).astype(np.float32)

    def run(L, to_dev):
        bufs = {k: to_dev(v) for k, v in dict(x=x, h2=h2, h23=h23, loc=loc, ls=ls, z=np.empty_like(x), y=np.empty_like(x),
                                              ld=np.empty(N, np.float32), lp=np.empty(N, np.float32),
                                              total=np.zeros(1, np.float64)).items()}
        p = {k: (v.data_ptr() if hasattr(v, "data_ptr") else v.ctypes.data) for k, v in bufs.items()}
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if to_dev is dev_any else None
        assert L.tfk_affine_coupling_fwd(p["x"], p["h2"], p["z"], p["ld"], N, D, None, T, 0, stream) == 0
        assert L.tfk_rqs_coupling_inv(p["z"], p["h23"], p["y"], p["ld"], N, D, None, T, K, f32(50.0), 1, stream) == 0
        assert L.tfk_diag_gauss_logprob(p["y"], p["loc"], p["ls"], p["ld"], p["lp"], N, D, stream) == 0
        assert L.tfk_sum_f32(p["lp"], p["total"], N, stream) == 0
        if to_dev is dev_any:
            torch.cuda.synchronize()
            return bufs["y"].cpu().numpy(), bufs["lp"].cpu().numpy(), float(bufs["total"].cpu()[0])
        return bufs["y"], bufs["lp"], float(bufs["total"][0])

    def dev_any(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    y_d, lp_d, tot_d = run(gpu, dev_any)
    y_h, lp_h, tot_h = run(cpu, lambda a: np.ascontiguousarray(a))
    assert rel(y_d, y_h) < 4e-5 and rel(lp_d, lp_h) < 1e-5
    assert abs(tot_d - tot_h) < 1e-5 * abs(tot_h)
