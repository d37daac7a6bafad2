"""ctypes/numpy front-end of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg -- never from ``torchflows_amd``.

The oracle restates the reference's algorithm (davidnabergoj/torchflows v1.2.0)
for the coupling-flow hot path in scalar fp32 C; this module only marshals numpy
arrays into it and rebuilds the *preset recipe* of the reference
(``bijections/finite/autoregressive/architectures.py:46-53``) from a state dict
that uses the reference's key names.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ELEMENTWISE_AFFINE = 0
ELEMENTWISE_INVERSE_AFFINE = 1
PERMUTATION = 2
AFFINE_COUPLING = 3
RQS_COUPLING = 4
SHIFT_COUPLING = 5
LRS_COUPLING = 6

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


class _OrcLayer(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("value", _f32p),
        ("perm_fwd", _i32p),
        ("perm_inv", _i32p),
        ("src_idx", _i32p),
        ("S", C.c_int32),
        ("tgt_idx", _i32p),
        ("T", C.c_int32),
        ("n_linear", C.c_int32),
        ("dims", _i32p),
        ("W", C.POINTER(_f32p)),
        ("b", C.POINTER(_f32p)),
        ("K", C.c_int32),
        ("boundary", C.c_float),
        ("autoregressive", C.c_int32),
        ("swap_transformer", C.c_int32),
    ]


def _sources_sha256() -> str:
    import hashlib
    h = hashlib.sha256()
    for p in (os.path.join(_HERE, "oracle.c"), os.path.join(_HERE, "oracle.h"), os.path.join(_HERE, "oracle_tfk.c"),
              os.path.join(_HERE, "..", "include", "tfk.h"), os.path.join(_HERE, "Makefile")):
        h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile).  Returns its path.
    Staleness is judged by a hash of the sources kept beside the library, not by modification times: a copy of the
    tree (the snapshot sent to the GPU box) does not preserve their order, and a rebuild there would fork a compiler
    out of a process that may already have initialised the GPU."""
    stamp = _LIB_PATH + ".sha256"
    want = _sources_sha256()
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if force or not os.path.exists(_LIB_PATH) or have != want:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_halfsplit_mask.argtypes = [C.c_int, _u8p, _u8p]
        L.orc_mask_to_index.argtypes = [_u8p, C.c_int, _i32p]
        L.orc_mask_to_index.restype = C.c_int
        L.orc_reverse_permutation.argtypes = [C.c_int, _i32p, _i32p]
        L.orc_checkerboard_mask.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p]
        L.orc_channelwise_mask.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p]
        L.orc_squeeze_index.argtypes = [C.c_int, C.c_int, C.c_int, _i32p]
        for name in ("orc_conv1x1_fwd", "orc_conv1x1_inv"):
            getattr(L, name).argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int]
        for name in ("orc_affine_fwd", "orc_affine_inv"):
            getattr(L, name).argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int]
        for name in ("orc_rqs_fwd", "orc_rqs_inv"):
            getattr(L, name).argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _i32p,
                                         C.c_int64, C.c_int, C.c_int, C.c_float]
        L.orc_affine_bwd.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int]
        L.orc_rqs_bwd.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int,
                                  C.c_int, C.c_float, C.c_int]
        for name in ("orc_lrs_fwd", "orc_lrs_inv"):
            getattr(L, name).argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_float]
        L.orc_rqs_knots.argtypes = [_f32p, C.c_int64, C.c_int, C.c_float, _f32p, _f32p, _f32p]
        L.orc_diag_gauss_logprob.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int]
        L.orc_composition_forward.argtypes = [C.POINTER(_OrcLayer), C.c_int, _f32p, _f32p,
                                              C.c_int, _f32p, _f32p, _f32p, _f32p,
                                              C.c_int64, C.c_int]
        L.orc_composition_inverse.argtypes = [C.POINTER(_OrcLayer), C.c_int, _f32p, _f32p,
                                              C.c_int, _f32p, _f32p, C.c_int64, C.c_int]
        L.orc_flow_log_prob.argtypes = [C.POINTER(_OrcLayer), C.c_int, _f32p, _f32p,
                                        _f32p, _f32p, C.c_int, _f32p, _f32p,
                                        C.c_int64, C.c_int]
        L.orc_actnorm_init.argtypes = [_f32p, C.c_int64, C.c_int, _f32p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _fp(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _ip(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_i32p) if a is not None else None


# --------------------------------------------------------------------------
# integer rules
# --------------------------------------------------------------------------

def halfsplit_mask(D: int):
    s = np.zeros(D, np.uint8)
    t = np.zeros(D, np.uint8)
    lib().orc_halfsplit_mask(D, s.ctypes.data_as(_u8p), t.ctypes.data_as(_u8p))
    return s.astype(bool), t.astype(bool)


def mask_to_index(mask) -> np.ndarray:
    m = np.ascontiguousarray(np.asarray(mask).reshape(-1).astype(np.uint8))
    idx = np.zeros(m.size, np.int32)
    n = lib().orc_mask_to_index(m.ctypes.data_as(_u8p), m.size, _ip(idx))
    return idx[:n].copy()


def reverse_permutation(D: int):
    f = np.zeros(D, np.int32)
    i = np.zeros(D, np.int32)
    lib().orc_reverse_permutation(D, _ip(f), _ip(i))
    return f, i


def image_mask(kind: str, shape, invert: bool = False):
    """(source, target) boolean masks of shape (C, H, W); kind = 'checkerboard' | 'channel_wise'."""
    Cc, H, W = (int(v) for v in shape)
    s = np.zeros(Cc * H * W, np.uint8)
    t = np.zeros_like(s)
    fn = lib().orc_checkerboard_mask if kind == "checkerboard" else lib().orc_channelwise_mask
    fn(Cc, H, W, int(invert), s.ctypes.data_as(_u8p), t.ctypes.data_as(_u8p))
    return s.reshape(Cc, H, W).astype(bool), t.reshape(Cc, H, W).astype(bool)


def squeeze_index(shape) -> np.ndarray:
    Cc, H, W = (int(v) for v in shape)
    idx = np.zeros(Cc * H * W, np.int32)
    lib().orc_squeeze_index(Cc, H, W, _ip(idx))
    return idx


def conv1x1(x, h, inverse: bool = False):
    """x: (N, n, *pixels) channel-major, h: (N, n + n(n-1)).  Returns (y, logdet (N,))."""
    x = _f32(x)
    h = _f32(h)
    N, n = x.shape[0], x.shape[1]
    HW = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    assert h.shape == (N, n + n * (n - 1))
    y = np.empty_like(x)
    ld = np.empty(N, np.float32)
    fn = lib().orc_conv1x1_inv if inverse else lib().orc_conv1x1_fwd
    fn(_fp(x), _fp(h), _fp(y), _fp(ld), N, n, HW)
    return y, ld


# --------------------------------------------------------------------------
# transformers
# --------------------------------------------------------------------------

def affine(x, h, inverse: bool = False):
    x = _f32(x)
    h = _f32(h)
    N, T = x.shape
    assert h.shape == (N, T, 2)
    out = np.empty_like(x)
    ld = np.empty(N, np.float32)
    fn = lib().orc_affine_inv if inverse else lib().orc_affine_fwd
    fn(_fp(x), _fp(h), _fp(out), _fp(ld), N, T)
    return out, ld


def rqs(x, h, n_bins: int = 8, boundary: float = 50.0, inverse: bool = False):
    """Returns (out, logdet (N,), logdet_el (N,T), bin index (N,T))."""
    x = _f32(x)
    h = _f32(h)
    N, T = x.shape
    assert h.shape == (N, T, 3 * n_bins - 1)
    out = np.empty_like(x)
    ld = np.empty(N, np.float32)
    ld_el = np.empty_like(x)
    k = np.empty((N, T), np.int32)
    fn = lib().orc_rqs_inv if inverse else lib().orc_rqs_fwd
    fn(_fp(x), _fp(h), _fp(out), _fp(ld), _fp(ld_el), _ip(k), N, T, n_bins, boundary)
    return out, ld, ld_el, k


def lrs(x, h, n_bins: int = 8, boundary: float = 50.0, inverse: bool = False):
    """Linear rational spline on (N,T) with h (N,T,4K): (out, logdet (N,))."""
    x, h = _f32(x), _f32(h)
    N, T = x.shape
    assert h.shape == (N, T, 4 * n_bins)
    out = np.empty_like(x)
    ld = np.empty(N, np.float32)
    (lib().orc_lrs_inv if inverse else lib().orc_lrs_fwd)(_fp(x), _fp(h), _fp(out), _fp(ld), N, T, n_bins, boundary)
    return out, ld


def affine_bwd(x, h, gz, gld, inverse: bool = False):
    """Reverse mode of ``affine``: (gx (N,T), gh (N,T,2)) for upstream gz (N,T), gld (N,)."""
    x, h, gz, gld = _f32(x), _f32(h), _f32(gz), _f32(gld)
    N, T = x.shape
    assert h.shape == (N, T, 2) and gz.shape == (N, T) and gld.shape == (N,)
    gx = np.empty_like(x)
    gh = np.empty_like(h)
    lib().orc_affine_bwd(_fp(x), _fp(h), _fp(gz), _fp(gld), _fp(gx), _fp(gh), N, T, int(inverse))
    return gx, gh


def rqs_bwd(x, h, gz, gld, n_bins: int = 8, boundary: float = 50.0, inverse: bool = False):
    """Reverse mode of ``rqs``: (gx (N,T), gh (N,T,3K-1))."""
    x, h, gz, gld = _f32(x), _f32(h), _f32(gz), _f32(gld)
    N, T = x.shape
    assert h.shape == (N, T, 3 * n_bins - 1) and gz.shape == (N, T) and gld.shape == (N,)
    gx = np.empty_like(x)
    gh = np.empty_like(h)
    lib().orc_rqs_bwd(_fp(x), _fp(h), _fp(gz), _fp(gld), _fp(gx), _fp(gh), N, T, n_bins,
                      boundary, int(inverse))
    return gx, gh


def rqs_knots(h, n_bins: int = 8, boundary: float = 50.0):
    """bin_x, bin_y, delta, each (..., K+1), for parameters h (..., 3K-1)."""
    h = _f32(h)
    lead = h.shape[:-1]
    M = int(np.prod(lead)) if lead else 1
    bx = np.empty((M, n_bins + 1), np.float32)
    by = np.empty_like(bx)
    dl = np.empty_like(bx)
    lib().orc_rqs_knots(_fp(h), M, n_bins, boundary, _fp(bx), _fp(by), _fp(dl))
    shp = (*lead, n_bins + 1)
    return bx.reshape(shp), by.reshape(shp), dl.reshape(shp)


def diag_gauss_logprob(z, loc, log_scale):
    z = _f32(z)
    N, D = z.shape
    out = np.empty(N, np.float32)
    lib().orc_diag_gauss_logprob(_fp(z), _fp(_f32(loc)), _fp(_f32(log_scale)), _fp(out), N, D)
    return out


def actnorm_init(x):
    x = _f32(x)
    N, D = x.shape
    value = np.empty((D, 2), np.float32)
    lib().orc_actnorm_init(_fp(x), N, D, _fp(value))
    return value


# --------------------------------------------------------------------------
# flow description
# --------------------------------------------------------------------------

@dataclass
class Layer:
    kind: int
    value: Optional[np.ndarray] = None          # (D, 2)
    perm_fwd: Optional[np.ndarray] = None
    perm_inv: Optional[np.ndarray] = None
    src_idx: Optional[np.ndarray] = None
    tgt_idx: Optional[np.ndarray] = None
    weights: List[np.ndarray] = field(default_factory=list)   # nn.Linear (out, in)
    biases: List[np.ndarray] = field(default_factory=list)
    n_bins: int = 8
    boundary: float = 50.0
    autoregressive: int = 0        # 1: forward parallel / inverse sequential (MADE); 2: exchanged
    swap_transformer: bool = False


class OracleFlow:
    """A composition of layers + a diagonal Gaussian base, evaluated by the C oracle."""

    def __init__(self, layers: Sequence[Layer], D: int, loc=None, log_scale=None,
                 context_size: int = 0):
        self.layers = list(layers)
        self.D = int(D)
        self.C = int(context_size)
        self.loc = _f32(np.zeros(D) if loc is None else loc)
        self.log_scale = _f32(np.zeros(D) if log_scale is None else log_scale)
        self._keep = []
        arr = (_OrcLayer * len(self.layers))()
        for i, L in enumerate(self.layers):
            o = arr[i]
            o.kind = L.kind
            if L.value is not None:
                L.value = _f32(L.value).reshape(-1, 2)
                o.value = _fp(L.value)
            elif L.kind in (ELEMENTWISE_AFFINE, ELEMENTWISE_INVERSE_AFFINE):
                # context-conditioned elementwise layer (h predicted from the context)
                L.weights = [_f32(w) for w in L.weights]
                L.biases = [_f32(b) for b in L.biases]
                dims = _i32([L.weights[0].shape[1]] + [w.shape[0] for w in L.weights])
                Wp = (_f32p * len(L.weights))(*[_fp(w) for w in L.weights])
                bp = (_f32p * len(L.biases))(*[_fp(b) for b in L.biases])
                self._keep += [dims, Wp, bp]
                o.n_linear = len(L.weights)
                o.dims = _ip(dims)
                o.W = Wp
                o.b = bp
                assert dims[0] == self.C and dims[-1] == 2 * self.D
            if L.perm_fwd is not None:
                L.perm_fwd = _i32(L.perm_fwd)
                L.perm_inv = _i32(L.perm_inv)
                o.perm_fwd = _ip(L.perm_fwd)
                o.perm_inv = _ip(L.perm_inv)
            if L.src_idx is not None:
                L.src_idx = _i32(L.src_idx)
                L.tgt_idx = _i32(L.tgt_idx)
                o.src_idx = _ip(L.src_idx)
                o.S = L.src_idx.size
                o.tgt_idx = _ip(L.tgt_idx)
                o.T = L.tgt_idx.size
                L.weights = [_f32(w) for w in L.weights]
                L.biases = [_f32(b) for b in L.biases]
                dims = _i32([L.weights[0].shape[1]] + [w.shape[0] for w in L.weights])
                Wp = (_f32p * len(L.weights))(*[_fp(w) for w in L.weights])
                bp = (_f32p * len(L.biases))(*[_fp(b) for b in L.biases])
                self._keep += [dims, Wp, bp]
                o.n_linear = len(L.weights)
                o.dims = _ip(dims)
                o.W = Wp
                o.b = bp
                assert dims[0] == o.S + self.C, (dims[0], o.S, self.C)
            o.K = L.n_bins
            o.boundary = L.boundary
            o.autoregressive = L.autoregressive
            o.swap_transformer = 1 if L.swap_transformer else 0
        self._arr = arr

    # -- evaluation -------------------------------------------------------
    def _ctx(self, context, N):
        if context is None:
            assert self.C == 0
            return None
        c = _f32(context).reshape(N, self.C)
        return c

    def forward(self, x, context=None, trace: bool = False):
        x = _f32(x).reshape(-1, self.D)
        N = x.shape[0]
        ctx = self._ctx(context, N)
        z = np.empty_like(x)
        ld = np.empty(N, np.float32)
        tz = np.empty((len(self.layers), N, self.D), np.float32) if trace else None
        tl = np.empty((len(self.layers), N), np.float32) if trace else None
        lib().orc_composition_forward(self._arr, len(self.layers), _fp(x), _fp(ctx), self.C,
                                      _fp(z), _fp(ld), _fp(tz), _fp(tl), N, self.D)
        return (z, ld, tz, tl) if trace else (z, ld)

    def inverse(self, z, context=None):
        z = _f32(z).reshape(-1, self.D)
        N = z.shape[0]
        ctx = self._ctx(context, N)
        x = np.empty_like(z)
        ld = np.empty(N, np.float32)
        lib().orc_composition_inverse(self._arr, len(self.layers), _fp(z), _fp(ctx), self.C,
                                      _fp(x), _fp(ld), N, self.D)
        return x, ld

    def log_prob(self, x, context=None, return_z: bool = False):
        x = _f32(x).reshape(-1, self.D)
        N = x.shape[0]
        ctx = self._ctx(context, N)
        z = np.empty_like(x) if return_z else None
        lp = np.empty(N, np.float32)
        lib().orc_flow_log_prob(self._arr, len(self.layers), _fp(self.loc), _fp(self.log_scale),
                                _fp(x), _fp(ctx), self.C, _fp(z), _fp(lp), N, self.D)
        return (z, lp) if return_z else lp

    def log_prob_grad(self, x, g_lp=None):
        """Reverse mode of ``log_prob`` (flows.py:628-658 as torch.autograd differentiates it):
        returns ``(gx (N,D), grads)`` for the upstream ``g_lp`` (N,) = d loss / d log_prob
        (default: ones).  ``grads[i]`` is a dict for layer i with ``value`` (D,2) and / or
        ``weights`` / ``biases`` lists.  Conditioners without a context only."""
        assert self.C == 0
        x = _f32(x).reshape(-1, self.D)
        N, D = x.shape
        g = np.ones(N, np.float32) if g_lp is None else _f32(g_lp).reshape(N)
        z, _, tz, _ = self.forward(x, trace=True)
        scale2 = np.exp(self.log_scale).astype(np.float32) ** 2
        g_rows = (-(z - self.loc) / scale2 * g[:, None]).astype(np.float32)   # gaussian.py:46-54
        g_ld = g.copy()                                                       # flows.py:647-648
        grads = [dict() for _ in self.layers]
        for i in range(len(self.layers) - 1, -1, -1):
            L = self.layers[i]
            x_in = x if i == 0 else tz[i - 1]
            if L.kind in (ELEMENTWISE_AFFINE, ELEMENTWISE_INVERSE_AFFINE):
                h = np.broadcast_to(L.value[None], (N, D, 2))
                g_rows, gh = affine_bwd(x_in, h, g_rows, g_ld,
                                        inverse=(L.kind == ELEMENTWISE_INVERSE_AFFINE))
                grads[i]["value"] = gh.sum(axis=0, dtype=np.float32)
            elif L.kind == PERMUTATION:
                g_rows = np.ascontiguousarray(g_rows[:, L.perm_inv])          # z_j = x[perm_fwd[j]]
            else:
                xa = x_in[:, L.src_idx]
                acts = [xa]                                   # conditioner forward, transforms.py:293-307
                for li, (W, b) in enumerate(zip(L.weights, L.biases)):
                    pre = acts[-1] @ W.T + b
                    acts.append(pre if li == len(L.weights) - 1 else np.tanh(pre).astype(np.float32))
                hflat = _f32(acts[-1])
                T = L.tgt_idx.size
                xb = _f32(x_in[:, L.tgt_idx])
                gzb = _f32(g_rows[:, L.tgt_idx])
                if L.kind == AFFINE_COUPLING:
                    gxb, gh = affine_bwd(xb, hflat.reshape(N, T, 2), gzb, g_ld)
                elif L.kind == RQS_COUPLING:
                    gxb, gh = rqs_bwd(xb, hflat.reshape(N, T, -1), gzb, g_ld, L.n_bins, L.boundary)
                else:                                         # Shift: z = x + h, log-det 0
                    gxb, gh = gzb, gzb
                gcur = _f32(gh).reshape(N, -1)
                gW, gb = [None] * len(L.weights), [None] * len(L.weights)
                for li in range(len(L.weights) - 1, -1, -1):
                    gW[li] = (gcur.T @ acts[li]).astype(np.float32)
                    gb[li] = gcur.sum(axis=0, dtype=np.float32)
                    gcur = gcur @ L.weights[li]
                    if li > 0:
                        gcur = (gcur * (1.0 - acts[li] ** 2)).astype(np.float32)   # tanh'
                grads[i]["weights"], grads[i]["biases"] = gW, gb
                g_new = g_rows.copy()
                g_new[:, L.tgt_idx] = gxb
                g_new[:, L.src_idx] += gcur
                g_rows = g_new
        return g_rows, grads

    def sample_log_prob(self, z, context=None):
        """Flow.sample(return_log_prob=True) on a given base draw z:
        x = inverse(z);  value = log p_base(z) + log|dx/dz|   (flows.py:699-712)."""
        z = _f32(z).reshape(-1, self.D)
        x, ld = self.inverse(z, context)
        return x, diag_gauss_logprob(z, self.loc, self.log_scale) + ld


_COUPLING_KIND = {"RealNVP": AFFINE_COUPLING, "CouplingRQNSF": RQS_COUPLING, "NICE": SHIFT_COUPLING,
                  "CouplingLRS": LRS_COUPLING,
                  # MADE-based presets (architectures.py:106-223): (kind, autoregressive mode, swap)
                  "MAF": (AFFINE_COUPLING, 1, False), "IAF": (AFFINE_COUPLING, 2, True),
                  "MaskedAutoregressiveRQNSF": (RQS_COUPLING, 1, False),
                  "InverseAutoregressiveRQNSF": (RQS_COUPLING, 2, False),
                  "MaskedAutoregressiveLRS": (LRS_COUPLING, 1, False),
                  "InverseAutoregressiveLRS": (LRS_COUPLING, 2, False)}


def preset_from_state_dict(arch: str, D: int, n_layers: int, sd: Dict[str, np.ndarray],
                           context_size: int = 0, n_bins: int = 8, boundary: float = 50.0,
                           permute: bool = True) -> OracleFlow:
    """Rebuild the reference preset recipe
    [ElementwiseAffine] + n_layers x [ReversePermutation, Coupling, ActNorm] +
    [ElementwiseAffine, ActNorm]   (architectures.py:46-53)
    from a reference-keyed state dict (``bijection.layers.{i}.…``).
    ``permute=False`` reproduces the ``edge_list=`` quirk (no permutation layers).
    """
    kind = _COUPLING_KIND[arch]
    ar_mode, swap = 0, False
    if isinstance(kind, tuple):
        kind, ar_mode, swap = kind
    src_mask, tgt_mask = halfsplit_mask(D)
    src, tgt = mask_to_index(src_mask), mask_to_index(tgt_mask)
    if ar_mode:
        src = tgt = np.arange(D, dtype=np.int32)
    fwd, inv = reverse_permutation(D)
    g = lambda k: np.asarray(sd[k], dtype=np.float32)

    def mlp(i):
        pre = f"bijection.layers.{i}.conditioner_transform.sequential."
        lin = sorted({int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre)})
        # MADE.MaskedLinear (transforms.py:192-198): the effective weight is weight * mask
        W = [g(f"{pre}{j}.weight") * g(f"{pre}{j}.mask") if f"{pre}{j}.mask" in sd else g(f"{pre}{j}.weight")
             for j in lin]
        return W, [g(f"{pre}{j}.bias") for j in lin]

    def elementwise(i):
        # ElementwiseAffine takes the flow's context_shape (architectures.py:46, :53):
        # with a context its h comes from a Linear conditioner, `value` is an empty buffer.
        if g(f"bijection.layers.{i}.value").ndim == 0:
            W, b = mlp(i)
            return Layer(ELEMENTWISE_AFFINE, weights=W, biases=b)
        return Layer(ELEMENTWISE_AFFINE, value=g(f"bijection.layers.{i}.value"))

    layers: List[Layer] = []
    i = 0
    layers.append(elementwise(i))
    i += 1
    for _ in range(n_layers):
        if permute:
            layers.append(Layer(PERMUTATION, perm_fwd=fwd, perm_inv=inv))
            i += 1
        W, b = mlp(i)
        layers.append(Layer(kind, src_idx=src, tgt_idx=tgt, weights=W, biases=b,
                            n_bins=n_bins, boundary=boundary, autoregressive=ar_mode, swap_transformer=swap))
        i += 1
        layers.append(Layer(ELEMENTWISE_INVERSE_AFFINE, value=g(f"bijection.layers.{i}.value")))
        i += 1
    layers.append(elementwise(i))
    i += 1
    layers.append(Layer(ELEMENTWISE_INVERSE_AFFINE, value=g(f"bijection.layers.{i}.value")))
    return OracleFlow(layers, D, loc=g("base.loc"), log_scale=g("base.log_scale"),
                      context_size=context_size)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))
