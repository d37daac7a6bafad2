"""The C-ABI of include/tfk.h as exported by the CPU restatement (oracle/oracle_tfk.c, host pointers, stream = NULL;
SURVEY.md 8(b)): the same ctypes call sequence a reference-side binding would make against libtfk.so, here run on the
host against the REFERENCE's golden outputs -- and that libtfk.so itself exports every symbol the header declares
(no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_vp, _i64, _i32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_float


def rel(a, b):
    return float(np.nanmax(np.abs(np.asarray(a, np.float64) - b) / np.maximum(1.0, np.abs(b))))


def ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


@pytest.fixture(scope="module")
def cpu_abi(oracle):
    L = C.CDLL(oracle.build())
    L.tfk_last_error.restype = C.c_char_p
    coupling = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _vp]
    for d in ("fwd", "inv"):
        getattr(L, f"tfk_affine_coupling_{d}").argtypes = coupling
        getattr(L, f"tfk_shift_coupling_{d}").argtypes = coupling
        getattr(L, f"tfk_rqs_coupling_{d}").argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _f, _i32, _vp]
        getattr(L, f"tfk_elementwise_affine_{d}").argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]
    L.tfk_permute.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_diag_gauss_logprob.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_sum_f32.argtypes = [_vp, _vp, _i64, _vp]
    return L


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tfk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfk_[a-z0-9_]+)\s*\(", text)))


def test_libtfk_exports_every_declared_symbol():
    from torchflows_amd import native
    L = C.CDLL(native.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 47 and set(names) == set(native.SYMBOLS)
    for n in names:
        assert hasattr(L, n), n
    L.tfk_abi_version.restype = C.c_int
    assert L.tfk_abi_version() == native.ABI_VERSION == 29


def test_cpu_restatement_exports_the_minimum_set(cpu_abi):
    assert cpu_abi.tfk_abi_version() == 29
    for n in ("tfk_affine_coupling_fwd", "tfk_affine_coupling_inv", "tfk_rqs_coupling_fwd", "tfk_rqs_coupling_inv",
              "tfk_elementwise_affine_fwd", "tfk_elementwise_affine_inv", "tfk_diag_gauss_logprob", "tfk_sum_f32",
              "tfk_permute", "tfk_shift_coupling_fwd", "tfk_shift_coupling_inv"):
        assert hasattr(cpu_abi, n), n
    # a stream on the host build is an error, reported through the ABI's own channel
    x = np.zeros((1, 2), np.float32)
    rc = cpu_abi.tfk_permute(ptr(x), None, ptr(np.zeros_like(x)), 1, 2, C.c_void_p(1))
    assert rc == 1 and b"stream" in cpu_abi.tfk_last_error()


@pytest.mark.parametrize("T", [2, 32, 128])
def test_affine_coupling_host_abi_vs_reference(cpu_abi, T):
    """x = [untouched half | target half]: tfk_affine_coupling_* with tgt_idx = NULL (HalfSplit tail) must give the
    reference's Affine.forward / inverse on the target half, leave the rest, and honour `accumulate`."""
    fx = load_golden("affine.npz")
    xb, h = fx[f"T{T}_x"], np.ascontiguousarray(fx[f"T{T}_h"])
    N, D = xb.shape[0], 2 * T
    x = np.ascontiguousarray(np.concatenate([np.full_like(xb, 7.0), xb], axis=1))
    for d, zk, lk in (("fwd", "z", "ld"), ("inv", "xinv", "ldinv")):
        z = np.empty_like(x)
        ld = np.full(N, 3.0, np.float32)
        rc = getattr(cpu_abi, f"tfk_affine_coupling_{d}")(ptr(x), ptr(h), ptr(z), ptr(ld), N, D, None, T, 1, None)
        assert rc == 0, cpu_abi.tfk_last_error()
        assert np.array_equal(z[:, :T], x[:, :T])
        assert rel(z[:, T:], fx[f"T{T}_{zk}"]) < 5e-6 and rel(ld - 3.0, fx[f"T{T}_{lk}"]) < 1e-5
    # an explicit index list (every other column) in place
    idx = np.arange(0, D, 2, dtype=np.int32)
    x2 = np.ascontiguousarray(np.zeros((N, D), np.float32))
    x2[:, idx] = xb
    ld = np.empty(N, np.float32)
    assert cpu_abi.tfk_affine_coupling_fwd(ptr(x2), ptr(h), ptr(x2), ptr(ld), N, D, ptr(idx), T, 0, None) == 0
    assert rel(x2[:, idx], fx[f"T{T}_z"]) < 5e-6 and np.all(x2[:, 1::2] == 0)


@pytest.mark.parametrize("tag", ["B50_K8", "B5_K8", "B50_K4"])
def test_rqs_coupling_host_abi_vs_reference(cpu_abi, tag):
    fx = load_golden("rqs.npz")
    xb, h = fx[f"{tag}_x"], np.ascontiguousarray(fx[f"{tag}_h"])
    N, T = xb.shape
    K = (h.shape[-1] + 1) // 3
    boundary = float(tag.split("_")[0][1:])
    x = np.ascontiguousarray(np.concatenate([np.zeros_like(xb), xb], axis=1))
    for d, zk, lk in (("fwd", "z", "ld"), ("inv", "xinv", "ldinv")):
        z, ld = np.empty_like(x), np.empty(N, np.float32)
        rc = getattr(cpu_abi, f"tfk_rqs_coupling_{d}")(ptr(x), ptr(h), ptr(z), ptr(ld), N, 2 * T, None, T, K,
                                                       _f(boundary), 0, None)
        assert rc == 0, cpu_abi.tfk_last_error()
        assert rel(z[:, T:], fx[f"{tag}_{zk}"]) < 4e-5 and rel(ld, fx[f"{tag}_{lk}"]) < 4e-5


def test_gauss_sum_permute_elementwise_host_abi(cpu_abi):
    fx = load_golden("gauss.npz")
    z = np.ascontiguousarray(fx["D64_value"])
    N, D = z.shape
    out = np.empty(N, np.float32)
    ldin = np.full(N, 2.0, np.float32)
    assert cpu_abi.tfk_diag_gauss_logprob(ptr(z), ptr(fx["D64_loc"]), ptr(fx["D64_log_scale"]), ptr(ldin), ptr(out),
                                          N, D, None) == 0
    assert rel(out - 2.0, fx["D64_log_prob"]) < 1e-5
    total = np.zeros(1, np.float64)
    assert cpu_abi.tfk_sum_f32(ptr(out), ptr(total), N, None) == 0
    assert abs(total[0] - out.astype(np.float64).sum()) < 1e-9 * abs(total[0])
    zr = np.empty_like(z)
    assert cpu_abi.tfk_permute(ptr(z), None, ptr(zr), N, D, None) == 0
    assert np.array_equal(zr, z[:, ::-1])
    # ElementwiseAffine / ActNorm with global parameters: fwd then inv is the identity, log-dets cancel
    rng = np.random.default_rng(0)
    value = np.ascontiguousarray(rng.standard_normal((D, 2)).astype(np.float32))
    for inverse_affine in (0, 1):
        y, ld = np.empty_like(z), np.empty(N, np.float32)
        assert cpu_abi.tfk_elementwise_affine_fwd(ptr(z), ptr(value), ptr(y), ptr(ld), N, D, inverse_affine, 0, None) == 0
        back = np.empty_like(z)
        assert cpu_abi.tfk_elementwise_affine_inv(ptr(y), ptr(value), ptr(back), ptr(ld), N, D, inverse_affine, 1, None) == 0
        assert rel(back, z) < 1e-5 and np.abs(ld).max() < 1e-4
