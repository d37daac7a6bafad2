"""ctypes binding of libtfk.so (C-ABI declared in include/tfk.h).

PyTorch is used here only as plumbing: device memory (``tensor.data_ptr()``), the
current HIP stream and the device guard.  No torch types cross the boundary.

The product path never falls back: if the library is missing or a call fails the
error is raised (``NativeError``).  Nothing in this package imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# TORCHFLOWS_AMD_LIB: load another build of the same ABI (kernel experiments / ablations)
LIB_PATH = os.environ.get("TORCHFLOWS_AMD_LIB") or os.path.join(_PKG, "lib", "libtfk.so")
CSRC = os.path.join(_PKG, "csrc")

# every symbol include/tfk.h declares (tests check the built library exports all of them)
SYMBOLS = (
    "tfk_abi_version", "tfk_last_error", "tfk_device_info",
    "tfk_affine_coupling_fwd", "tfk_affine_coupling_inv",
    "tfk_shift_coupling_fwd", "tfk_shift_coupling_inv",
    "tfk_rqs_coupling_fwd", "tfk_rqs_coupling_inv",
    "tfk_lrs_coupling_fwd", "tfk_lrs_coupling_inv",
    "tfk_conv1x1_coupling_fwd", "tfk_conv1x1_coupling_inv", "tfk_conv1x1_coupling_bwd",
    "tfk_elementwise_affine_fwd", "tfk_elementwise_affine_inv",
    "tfk_permute", "tfk_diag_gauss_logprob",
    "tfk_sum_workspace_bytes", "tfk_sum_f32", "tfk_sum_f32_ws",
    "tfk_flow_supported", "tfk_flow_run",
    "tfk_flow_mfma_supported", "tfk_flow_lean_supported", "tfk_flow_run_mfma", "tfk_flow_run_mfma_in", "tfk_flow_run_mfma_ctx",
    "tfk_flow_sum_workspace_bytes", "tfk_flow_run_mfma_sum",
    "tfk_affine_coupling_bwd", "tfk_shift_coupling_bwd",
    "tfk_rqs_coupling_bwd_supported", "tfk_rqs_coupling_bwd", "tfk_lrs_coupling_bwd",
    "tfk_elementwise_affine_bwd_workspace_bytes", "tfk_elementwise_affine_bwd",
    "tfk_diag_gauss_logprob_bwd",
    "tfk_coupling_train_bwd_supported", "tfk_coupling_train_bwd_out_floats",
    "tfk_coupling_train_bwd_workspace_bytes", "tfk_affine_coupling_train_bwd",
    "tfk_rqs_coupling_train_bwd_supported", "tfk_rqs_coupling_train_bwd", "tfk_rqs_coupling_train_bwd_hid",
    "tfk_rows_outer_workspace_bytes", "tfk_rows_outer",
    "tfk_made_affine_sequential", "tfk_made_rqs_sequential_lds_bytes", "tfk_made_rqs_sequential",
    "tfk_made_lrs_sequential_lds_bytes", "tfk_made_lrs_sequential",
    "tfk_conv3x3_block_supported", "tfk_conv3x3_relu_pool_affine", "tfk_conv1x1_frame", "tfk_bounded_sigmoid", "tfk_bounded_sigmoid_bwd",
    "tfk_glow_weight_floats", "tfk_glow_plan", "tfk_glow_coupling", "tfk_rows_fma",
    "tfk_glow_level_blob_bytes", "tfk_glow_level_pack", "tfk_glow_level_info", "tfk_glow_level",
    "tfk_rows_fma_gauss_logprob",
    "tfk_convnet_train_workspace_bytes", "tfk_convnet_train_block_supported", "tfk_convnet_train_block_fwd",
    "tfk_convnet_train_block_bwd", "tfk_convnet_train_frame_fwd", "tfk_convnet_train_frame_bwd",
    "tfk_convnet_train_linear_wgrad", "tfk_convnet_train_linear_fwd", "tfk_convnet_train_linear_bwd_input",
    "tfk_convnet_train_linear_prep", "tfk_convnet_train_forward", "tfk_convnet_train_backward",
    "tfk_convnet_train_sums_floats",
)

ABI_VERSION = 29


class NativeError(RuntimeError):
    """libtfk is missing, mismatched, or a kernel call was rejected."""


_vp = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64



class GlowLayer(C.Structure):
    """``tfk_glow_layer`` of include/tfk.h: one convolutional coupling of an image flow, everything that does not
    depend on the sample prepared by the caller (torchflows_amd/image_program.py)."""
    _fields_ = [("kind", _i32), ("c_in", _i32), ("hi", _i32), ("wi", _i32), ("oy", _i32), ("ox", _i32),
                ("kh", _i32), ("kw", _i32), ("T", _i32), ("n_params", _i32), ("n_ch", _i32), ("hw", _i32),
                ("slots", _i32), ("block", _i32), ("cg1", _i32), ("cg2", _i32), ("grid", _i32),
                ("src_idx", _vp), ("src_st", _vp), ("tgt_idx", _vp), ("tgt_st", _vp),
                ("weights", _vp), ("bg1", _vp), ("bg2", _vp), ("w_eff", _vp), ("b_eff", _vp)]


class ConvNetTrainPlan(C.Structure):
    """``tfk_convnet_train_plan`` (include/tfk.h): the ConvNet conditioner's parameters and the activations its two passes
    share, as device pointers."""
    _fields_ = [("mod1_w", C.c_void_p), ("mod1_b", C.c_void_p),
                ("conv_w", C.c_void_p * 3), ("conv_b", C.c_void_p * 3),
                ("bn_w", C.c_void_p * 3), ("bn_b", C.c_void_p * 3),
                ("bn_mean", C.c_void_p * 3), ("bn_var", C.c_void_p * 3), ("bn_count", C.c_void_p * 3),
                ("bn_eps", C.c_float * 3), ("bn_momentum", C.c_float * 3),
                ("mod2_w", C.c_void_p), ("mod2_b", C.c_void_p), ("lin_w", C.c_void_p), ("lin_b", C.c_void_p),
                ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("M", C.c_int32),
                ("a0", C.c_void_p), ("y", C.c_void_p * 3), ("amax", C.c_void_p * 3), ("stats", C.c_void_p * 3),
                ("a16", C.c_void_p), ("lin_fold", C.c_void_p), ("workspace", C.c_void_p)]


class GlowLevelStep(C.Structure):
    """``tfk_glow_level_step`` of include/tfk.h: a coupling inside a level program (tfk_glow_level)."""
    _fields_ = [("layer", GlowLayer), ("inverse", _i32), ("reserved", _i32), ("src_loc", _vp), ("tgt_loc", _vp),
                ("w4", _vp), ("b4", _vp), ("weights_host", _vp)]


_lib: Optional[C.CDLL] = None
calls = 0   # number of kernel entry points invoked (tests assert the HIP path really ran)


def build(verbose: bool = False) -> str:
    """Compile libtfk.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", CSRC, "-B", "-j", str(min(os.cpu_count() or 1, 8))], check=True, stdout=out)
    global _lib
    _lib = None
    return LIB_PATH


def _bind(L: C.CDLL) -> None:
    coupling = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _vp]
    for n in ("tfk_affine_coupling_fwd", "tfk_affine_coupling_inv",
              "tfk_shift_coupling_fwd", "tfk_shift_coupling_inv"):
        getattr(L, n).argtypes = coupling
    rqs = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, C.c_float, _i32, _vp]
    L.tfk_rqs_coupling_fwd.argtypes = rqs
    L.tfk_rqs_coupling_inv.argtypes = rqs
    L.tfk_lrs_coupling_fwd.argtypes = rqs
    L.tfk_lrs_coupling_inv.argtypes = rqs
    conv = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp]
    L.tfk_conv1x1_coupling_fwd.argtypes = conv
    L.tfk_conv1x1_coupling_inv.argtypes = conv
    L.tfk_conv1x1_coupling_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp]
    ew = [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]
    L.tfk_elementwise_affine_fwd.argtypes = ew
    L.tfk_elementwise_affine_inv.argtypes = ew
    L.tfk_permute.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_diag_gauss_logprob.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_sum_workspace_bytes.argtypes = [_i64]
    L.tfk_sum_workspace_bytes.restype = _i64
    L.tfk_sum_f32.argtypes = [_vp, _vp, _i64, _vp]
    L.tfk_sum_f32_ws.argtypes = [_vp, _vp, _vp, _i64, _vp]
    L.tfk_flow_supported.argtypes = [_i32]
    L.tfk_flow_run.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, C.POINTER(_i32), _i32,
                               _vp, _i64, _i32, _vp]
    L.tfk_flow_mfma_supported.argtypes = [_i32]
    L.tfk_flow_lean_supported.argtypes = [_i32]
    L.tfk_flow_run_mfma.argtypes = L.tfk_flow_run.argtypes
    L.tfk_flow_run_mfma_in.argtypes = [_vp, _i32] + L.tfk_flow_run.argtypes[1:]
    L.tfk_flow_run_mfma_ctx.argtypes = [_vp, _vp, _i32] + L.tfk_flow_run.argtypes[1:]
    L.tfk_flow_sum_workspace_bytes.argtypes = []
    L.tfk_flow_sum_workspace_bytes.restype = _i64
    L.tfk_flow_run_mfma_sum.argtypes = [_vp, _i32] + L.tfk_flow_run.argtypes[1:-1] + [_vp, _vp, _vp]
    L.tfk_affine_coupling_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _vp]
    L.tfk_shift_coupling_bwd.argtypes = [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _vp]
    L.tfk_rqs_coupling_bwd_supported.argtypes = [_i32]
    L.tfk_rqs_coupling_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32,
                                       C.c_float, _i32, _vp]
    L.tfk_lrs_coupling_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32,
                                       C.c_float, _i32, _vp]
    L.tfk_elementwise_affine_bwd_workspace_bytes.argtypes = [_i64, _i32]
    L.tfk_elementwise_affine_bwd_workspace_bytes.restype = _i64
    L.tfk_elementwise_affine_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]
    L.tfk_diag_gauss_logprob_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_coupling_train_bwd_supported.argtypes = [_i32]
    L.tfk_coupling_train_bwd_out_floats.argtypes = [_i32]
    L.tfk_coupling_train_bwd_out_floats.restype = _i64
    L.tfk_coupling_train_bwd_workspace_bytes.argtypes = [_i32]
    L.tfk_coupling_train_bwd_workspace_bytes.restype = _i64
    L.tfk_affine_coupling_train_bwd.argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32,
                                                _vp, _i32, _vp]
    L.tfk_rqs_coupling_train_bwd_supported.argtypes = [_i32, _i32]
    L.tfk_rqs_coupling_train_bwd.argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32,
                                             C.c_float, _i32, _vp, _i32, _vp]
    L.tfk_rqs_coupling_train_bwd_hid.argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _i32,
                                                 C.c_float, _i32, _vp, _i32, _vp]
    L.tfk_rows_outer_workspace_bytes.argtypes = [_i32]
    L.tfk_rows_outer_workspace_bytes.restype = _i64
    L.tfk_rows_outer.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp]
    L.tfk_made_affine_sequential.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]
    L.tfk_made_rqs_sequential_lds_bytes.argtypes = [_i32, _i32, _i32]
    L.tfk_made_rqs_sequential_lds_bytes.restype = _i64
    L.tfk_made_rqs_sequential.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _i32, _vp]
    L.tfk_made_lrs_sequential_lds_bytes.argtypes = [_i32, _i32, _i32]
    L.tfk_made_lrs_sequential_lds_bytes.restype = _i64
    L.tfk_made_lrs_sequential.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _i32, _vp]
    L.tfk_conv3x3_block_supported.argtypes = [_i32, _i32]
    L.tfk_conv3x3_relu_pool_affine.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]
    L.tfk_bounded_sigmoid.argtypes = [_vp, _vp, _i64, C.c_float, C.c_float, _vp]
    L.tfk_bounded_sigmoid_bwd.argtypes = [_vp, _vp, _vp, _i64, C.c_float, C.c_float, _vp]
    L.tfk_conv1x1_frame.argtypes = [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]
    L.tfk_glow_weight_floats.argtypes = [_i32]
    L.tfk_glow_weight_floats.restype = _i64
    pi = C.POINTER(_i32)
    L.tfk_glow_plan.argtypes = [C.POINTER(GlowLayer), _i32, pi, pi, pi, pi, pi, pi]
    L.tfk_glow_coupling.argtypes = [_vp, _vp, _i64, _i32, C.POINTER(GlowLayer), _i32, _vp]
    L.tfk_rows_fma.argtypes = [_vp, _vp, _i64, _i32, _vp]
    L.tfk_rows_fma_gauss_logprob.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]
    ps = C.POINTER(GlowLevelStep)
    L.tfk_glow_level_blob_bytes.argtypes = [ps, _i32, _i32, _i32, _i32, _i32]
    L.tfk_glow_level_blob_bytes.restype = _i64
    L.tfk_glow_level_pack.argtypes = [ps, _i32, _i32, _i32, _i32, _i32, _vp, _i64]
    L.tfk_glow_level_info.argtypes = [_vp, pi, pi, pi, pi]
    L.tfk_glow_level.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]
    L.tfk_convnet_train_workspace_bytes.argtypes = []
    L.tfk_convnet_train_workspace_bytes.restype = _i64
    L.tfk_convnet_train_block_supported.argtypes = [_i32, _i32, _i32]
    L.tfk_convnet_train_block_fwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float,
                                              _i32, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]
    L.tfk_convnet_train_block_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp,
                                              _i64, _i32, _i32, _i32, _vp]
    L.tfk_convnet_train_frame_fwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                              _vp]
    L.tfk_convnet_train_frame_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32,
                                              _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]
    L.tfk_convnet_train_linear_wgrad.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]
    L.tfk_convnet_train_linear_prep.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]
    pp = C.POINTER(ConvNetTrainPlan)
    L.tfk_convnet_train_forward.argtypes = [pp, _vp, _vp, _i64, _i32, _i32, _vp]
    L.tfk_convnet_train_backward.argtypes = [pp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_convnet_train_sums_floats.argtypes = [_i32, _i32, _i32, _i32]
    L.tfk_convnet_train_sums_floats.restype = _i64
    L.tfk_convnet_train_linear_fwd.argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_convnet_train_linear_bwd_input.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp]
    L.tfk_last_error.restype = C.c_char_p
    L.tfk_device_info.argtypes = [C.c_char_p, _i32, C.POINTER(_i32)]


def lib() -> C.CDLL:
    """The loaded library; raises NativeError (never falls back) if it cannot be used."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950). There is no fallback path.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:   # e.g. libamdhip64 not loadable
            raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
        missing = [s for s in SYMBOLS if not hasattr(L, s)]
        if missing:
            raise NativeError(f"{LIB_PATH} lacks symbols {missing}; rebuild it")
        if L.tfk_abi_version() != ABI_VERSION:
            raise NativeError(f"{LIB_PATH} has ABI {L.tfk_abi_version()}, expected {ABI_VERSION}")
        _bind(L)
        _lib = L
    return _lib


def available() -> bool:
    try:
        lib()
        return True
    except NativeError:
        return False


def eligible(*tensors: Optional[torch.Tensor]) -> bool:
    """True when the HIP path applies: fp32 tensors on a HIP device, no autograd needed.
    (Training / fp64 / host tensors take the ATen composite path, as SURVEY.md app. B.)"""
    seen = False
    for t in tensors:
        if t is None:
            continue
        seen = True
        if t.device.type != "cuda" or t.dtype != torch.float32:
            return False
        if t.requires_grad and torch.is_grad_enabled():
            return False
    return seen


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeError(f"{what} failed (code {rc}): {lib().tfk_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor], what: str) -> Optional[int]:
    if t is None:
        return None
    if t.device.type != "cuda":
        raise NativeError(f"{what}: tensor is on {t.device}, the kernels need a HIP device")
    if not t.is_contiguous():
        raise NativeError(f"{what}: tensor must be contiguous")
    return t.data_ptr()


def _f32(t: Optional[torch.Tensor], what: str) -> Optional[int]:
    if t is not None and t.dtype != torch.float32:
        raise NativeError(f"{what}: expected float32, got {t.dtype}")
    return _ptr(t, what)


def _idx(t: Optional[torch.Tensor], what: str) -> Optional[int]:
    if t is not None and t.dtype != torch.int32:
        raise NativeError(f"{what}: expected int32 indices, got {t.dtype}")
    return _ptr(t, what)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_guard(t: torch.Tensor):
    """Make the tensor's device current for the launch -- a no-op (and no context-manager cost on
    the launch path) when it already is."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor) -> int:
    """The caller's current HIP stream on the tensor's device, as the integer handle the C-ABI takes."""
    if _raw_stream is not None:             # one C call (torch.cuda.current_stream() builds a Stream object: ~5 us)
        idx = t.device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(t.device).cuda_stream


def _rows(x: torch.Tensor, what: str):
    if x.dim() != 2:
        raise NativeError(f"{what}: expected a (N, D) tensor, got shape {tuple(x.shape)}")
    return x.shape[0], x.shape[1]


def _coupling(name, x, h, out, logdet, tgt_idx, T, P, accumulate, extra=()):
    global calls
    N, D = _rows(x, name)
    if out.shape != x.shape:
        raise NativeError(f"{name}: out shape {tuple(out.shape)} != x shape {tuple(x.shape)}")
    if h.numel() != N * T * P:
        raise NativeError(f"{name}: h has {h.numel()} elements, expected N*T*P = {N}*{T}*{P}")
    if logdet is not None and logdet.numel() != N:
        raise NativeError(f"{name}: logdet has {logdet.numel()} elements, expected {N}")
    if tgt_idx is not None and tgt_idx.numel() != T:
        raise NativeError(f"{name}: tgt_idx has {tgt_idx.numel()} entries, expected T = {T}")
    fn = getattr(lib(), name)
    args = (_f32(x, name), _f32(h, name), _f32(out, name), _f32(logdet, name), N, D,
            _idx(tgt_idx, name), T, *extra, 1 if accumulate else 0)
    with _device_guard(x):
        rc = fn(*args, _stream(x))
    calls += 1
    _check(rc, name)


def affine_coupling(x, h, out, logdet, tgt_idx, T, accumulate=False, inverse=False):
    """x, out: (N, D); h: (N, T, 2); logdet: (N,).  tgt_idx None = contiguous tail."""
    _coupling("tfk_affine_coupling_inv" if inverse else "tfk_affine_coupling_fwd",
              x, h, out, logdet, tgt_idx, T, 2, accumulate)


def shift_coupling(x, h, out, logdet, tgt_idx, T, accumulate=False, inverse=False):
    _coupling("tfk_shift_coupling_inv" if inverse else "tfk_shift_coupling_fwd",
              x, h, out, logdet, tgt_idx, T, 1, accumulate)


def rqs_coupling(x, h, out, logdet, tgt_idx, T, n_bins, boundary, accumulate=False,
                 inverse=False):
    """h: (N, T, 3*n_bins - 1)."""
    _coupling("tfk_rqs_coupling_inv" if inverse else "tfk_rqs_coupling_fwd",
              x, h, out, logdet, tgt_idx, T, 3 * n_bins - 1, accumulate,
              extra=(int(n_bins), C.c_float(float(boundary))))


def lrs_coupling(x, h, out, logdet, tgt_idx, T, n_bins, boundary, accumulate=False, inverse=False):
    """Linear rational spline coupling; h: (N, T, 4*n_bins)."""
    _coupling("tfk_lrs_coupling_inv" if inverse else "tfk_lrs_coupling_fwd",
              x, h, out, logdet, tgt_idx, T, 4 * n_bins, accumulate,
              extra=(int(n_bins), C.c_float(float(boundary))))


def conv1x1_coupling(x, h, out, logdet, tgt_idx, T, n_channels, accumulate=False, inverse=False):
    """Invertible 1x1 convolution on the T target positions (n_channels x T/n_channels pixels,
    channel-major); h: (N, n + n(n-1)) LU parameters per sample."""
    global calls
    name = "tfk_conv1x1_coupling_inv" if inverse else "tfk_conv1x1_coupling_fwd"
    N, D = _rows(x, name)
    n = int(n_channels)
    if out.shape != x.shape or logdet.numel() != N or h.numel() != N * (n + n * (n - 1)):
        raise NativeError(f"{name}: bad out / logdet / h shape")
    if tgt_idx is not None and tgt_idx.numel() != T:
        raise NativeError(f"{name}: tgt_idx has {tgt_idx.numel()} entries, expected T = {T}")
    args = (_f32(x, name), _f32(h, name), _f32(out, name), _f32(logdet, name), N, D,
            _idx(tgt_idx, name), T, n, 1 if accumulate else 0)
    with _device_guard(x):
        rc = getattr(lib(), name)(*args, _stream(x))
    calls += 1
    _check(rc, name)


def elementwise_affine(x, value, out, logdet, inverse_affine, accumulate=False, inverse=False):
    """value: (D, 2) global parameters; inverse_affine selects the ActNorm transformer."""
    global calls
    name = "tfk_elementwise_affine_inv" if inverse else "tfk_elementwise_affine_fwd"
    N, D = _rows(x, name)
    if value.numel() != 2 * D:
        raise NativeError(f"{name}: value has {value.numel()} elements, expected 2*D = {2 * D}")
    if out.shape != x.shape or logdet.numel() != N:
        raise NativeError(f"{name}: bad out/logdet shape")
    args = (_f32(x, name), _f32(value, name), _f32(out, name), _f32(logdet, name), N, D,
            1 if inverse_affine else 0, 1 if accumulate else 0)
    with _device_guard(x):
        rc = getattr(lib(), name)(*args, _stream(x))
    calls += 1
    _check(rc, name)


def permute(x, perm, out):
    """out[n, j] = x[n, perm[j]]; perm None = reversal."""
    global calls
    N, D = _rows(x, "tfk_permute")
    if out.shape != x.shape:
        raise NativeError("tfk_permute: bad out shape")
    if perm is not None and perm.numel() != D:
        raise NativeError(f"tfk_permute: perm has {perm.numel()} entries, expected D = {D}")
    args = (_f32(x, "tfk_permute"), _idx(perm, "tfk_permute"), _f32(out, "tfk_permute"), N, D)
    with _device_guard(x):
        rc = lib().tfk_permute(*args, _stream(x))
    calls += 1
    _check(rc, "tfk_permute")


def diag_gauss_logprob(z, loc, log_scale, logdet_in, out):
    global calls
    name = "tfk_diag_gauss_logprob"
    N, D = _rows(z, name)
    if loc.numel() != D or log_scale.numel() != D or out.numel() != N:
        raise NativeError(f"{name}: bad parameter/out shape")
    args = (_f32(z, name), _f32(loc, name), _f32(log_scale, name), _f32(logdet_in, name),
            _f32(out, name), N, D)
    with _device_guard(z):
        rc = lib().tfk_diag_gauss_logprob(*args, _stream(z))
    calls += 1
    _check(rc, name)


# ---- reverse mode (csrc/tfk_bwd.hip) ------------------------------------------------------
def _bwd_common(name, x, g, gld, tgt_idx, T):
    N, D = _rows(g, name)
    if x is not None and x.shape != g.shape:
        raise NativeError(f"{name}: x shape {tuple(x.shape)} != g shape {tuple(g.shape)}")
    if gld is not None and gld.numel() != N:
        raise NativeError(f"{name}: gld has {gld.numel()} elements, expected {N}")
    if tgt_idx is not None and tgt_idx.numel() != T:
        raise NativeError(f"{name}: tgt_idx has {tgt_idx.numel()} entries, expected T = {T}")
    return N, D


def affine_coupling_bwd(x, h, g, gld, gh, tgt_idx, T, inverse=False):
    """In place on g (N, D): target columns dL/d out -> dL/d x; gh (N, T, 2) written."""
    global calls
    name = "tfk_affine_coupling_bwd"
    N, D = _bwd_common(name, x, g, gld, tgt_idx, T)
    if h.numel() != N * T * 2 or gh.numel() != N * T * 2:
        raise NativeError(f"{name}: h / gh must hold N*T*2 = {N * T * 2} elements")
    args = (_f32(x, name), _f32(h, name), _f32(g, name), _f32(gld, name), _f32(gh, name), N, D,
            _idx(tgt_idx, name), T, 1 if inverse else 0)
    with _device_guard(g):
        rc = lib().tfk_affine_coupling_bwd(*args, _stream(g))
    calls += 1
    _check(rc, name)


def conv1x1_coupling_bwd(x, h, g, gld, gh, tgt_idx, T, n_channels, inverse=False):
    """Reverse mode of ``conv1x1_coupling``: in place on g (N, D) at the T target positions, gh (N, n + n(n-1)) written."""
    global calls
    name = "tfk_conv1x1_coupling_bwd"
    N, D = _bwd_common(name, x, g, gld, tgt_idx, T)
    n = int(n_channels)
    if h.numel() != N * n * n or gh.numel() != N * n * n:
        raise NativeError(f"{name}: h / gh must hold N * n^2 = {N * n * n} elements")
    args = (_f32(x, name), _f32(h, name), _f32(g, name), _f32(gld, name), _f32(gh, name), N, D,
            _idx(tgt_idx, name), T, n, 1 if inverse else 0)
    with _device_guard(g):
        rc = lib().tfk_conv1x1_coupling_bwd(*args, _stream(g))
    calls += 1
    _check(rc, name)


def shift_coupling_bwd(g, gh, tgt_idx, T, inverse=False):
    global calls
    name = "tfk_shift_coupling_bwd"
    N, D = _bwd_common(name, None, g, None, tgt_idx, T)
    if gh.numel() != N * T:
        raise NativeError(f"{name}: gh must hold N*T = {N * T} elements")
    args = (_f32(g, name), _f32(gh, name), N, D, _idx(tgt_idx, name), T, 1 if inverse else 0)
    with _device_guard(g):
        rc = lib().tfk_shift_coupling_bwd(*args, _stream(g))
    calls += 1
    _check(rc, name)


def rqs_coupling_bwd(x, h, g, gld, gh, tgt_idx, T, n_bins, boundary, inverse=False):
    global calls
    name = "tfk_rqs_coupling_bwd"
    N, D = _bwd_common(name, x, g, gld, tgt_idx, T)
    P = 3 * int(n_bins) - 1
    if h.numel() != N * T * P or gh.numel() != N * T * P:
        raise NativeError(f"{name}: h / gh must hold N*T*P = {N * T * P} elements")
    args = (_f32(x, name), _f32(h, name), _f32(g, name), _f32(gld, name), _f32(gh, name), N, D,
            _idx(tgt_idx, name), T, int(n_bins), C.c_float(float(boundary)), 1 if inverse else 0)
    with _device_guard(g):
        rc = lib().tfk_rqs_coupling_bwd(*args, _stream(g))
    calls += 1
    _check(rc, name)


def lrs_coupling_bwd(x, h, g, gld, gh, tgt_idx, T, n_bins, boundary, inverse=False):
    global calls
    name = "tfk_lrs_coupling_bwd"
    N, D = _bwd_common(name, x, g, gld, tgt_idx, T)
    P = 4 * int(n_bins)
    if h.numel() != N * T * P or gh.numel() != N * T * P:
        raise NativeError(f"{name}: h / gh must hold N*T*P = {N * T * P} elements")
    args = (_f32(x, name), _f32(h, name), _f32(g, name), _f32(gld, name), _f32(gh, name), N, D,
            _idx(tgt_idx, name), T, int(n_bins), C.c_float(float(boundary)), 1 if inverse else 0)
    with _device_guard(g):
        rc = lib().tfk_lrs_coupling_bwd(*args, _stream(g))
    calls += 1
    _check(rc, name)


def elementwise_affine_bwd(x, value, g, gld, want_param, inverse=False, out=None):
    """In place on g (all D columns).  Returns dL/dvalue (D, 2) when ``want_param`` (written into ``out``, 2 D
    contiguous floats, when given)."""
    global calls
    name = "tfk_elementwise_affine_bwd"
    N, D = _rows(g, name)
    if value.numel() != 2 * D:
        raise NativeError(f"{name}: value has {value.numel()} elements, expected 2*D = {2 * D}")
    gvalue = ws = None
    if want_param:
        if x is None or x.shape != g.shape or gld is None or gld.numel() != N:
            raise NativeError(f"{name}: the parameter gradient needs x (N, D) and gld (N,)")
        if out is not None and (out.numel() != 2 * D or not out.is_contiguous() or out.dtype != torch.float32):
            raise NativeError(f"{name}: out must hold 2*D = {2 * D} contiguous floats")
        gvalue = torch.empty(D, 2, dtype=torch.float32, device=g.device) if out is None else out.view(D, 2)
        nbytes = int(lib().tfk_elementwise_affine_bwd_workspace_bytes(N, D))
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=g.device)
    args = (_f32(x if want_param else None, name), _f32(value, name), _f32(g, name),
            _f32(gld if want_param else None, name), _f32(gvalue, name), _f32(ws, name), N, D,
            1 if inverse else 0)
    with _device_guard(g):
        rc = lib().tfk_elementwise_affine_bwd(*args, _stream(g))
    calls += 1
    _check(rc, name)
    return gvalue


def diag_gauss_logprob_bwd(z, loc, log_scale, glp, g):
    global calls
    name = "tfk_diag_gauss_logprob_bwd"
    N, D = _rows(z, name)
    if loc.numel() != D or log_scale.numel() != D or glp.numel() != N or g.shape != z.shape:
        raise NativeError(f"{name}: bad parameter / gradient shape")
    args = (_f32(z, name), _f32(loc, name), _f32(log_scale, name), _f32(glp, name), _f32(g, name), N, D)
    with _device_guard(z):
        rc = lib().tfk_diag_gauss_logprob_bwd(*args, _stream(z))
    calls += 1
    _check(rc, name)


def affine_coupling_train_bwd(x, g, gld, params, gemm2_steps, out, workspace, inverse_form=False,
                              gscale=None, g_reversed=False):
    """Fused conditioner + transform + MLP backward of one HalfSplit affine coupling (in place on
    g, accumulator-layout weight gradients into ``out``)."""
    global calls
    name = "tfk_affine_coupling_train_bwd"
    N, D = _rows(g, name)
    if x.shape != g.shape or gld.numel() != N:
        raise NativeError(f"{name}: bad x / gld shape")
    if out.numel() != int(lib().tfk_coupling_train_bwd_out_floats(D)):
        raise NativeError(f"{name}: out must hold tfk_coupling_train_bwd_out_floats(D) floats")
    if workspace.numel() * 4 < int(lib().tfk_coupling_train_bwd_workspace_bytes(D)):
        raise NativeError(f"{name}: workspace too small")
    args = (_f32(x, name), _f32(g, name), _f32(gld, name), _f32(params, name), params.numel(),
            int(gemm2_steps), _f32(out, name), _f32(workspace, name), N, D, 1 if inverse_form else 0,
            _f32(gscale, name), 1 if g_reversed else 0)
    if gscale is not None and gscale.numel() != D:
        raise NativeError(f"{name}: gscale must hold D = {D} floats")
    with _device_guard(g):
        rc = lib().tfk_affine_coupling_train_bwd(*args, _stream(g))
    calls += 1
    _check(rc, name)


def rqs_coupling_train_bwd(x, g, gld, params, gemm2_steps, gh_perm, gpre_perm, n_bins, boundary,
                           inverse=False, gscale=None, g_reversed=False, hid_perm=None):
    """Conditioner re-evaluation + RQ-spline backward + dL/dhidden + dL/dx_A in one launch (in place on
    g); gh_perm (N, 768) and gpre_perm (N, 16) in accumulator order; ``hid_perm`` (N, 16): also the hidden activations,
    column 15 == 1 (tfk_rqs_coupling_train_bwd_hid)."""
    global calls
    name = "tfk_rqs_coupling_train_bwd" if hid_perm is None else "tfk_rqs_coupling_train_bwd_hid"
    N, D = _rows(g, name)
    if x.shape != g.shape or gld.numel() != N or gh_perm.shape != (N, 768) or gpre_perm.shape != (N, 16):
        raise NativeError(f"{name}: bad x / gld / gh_perm / gpre_perm shape")
    if hid_perm is not None and hid_perm.shape != (N, 16):
        raise NativeError(f"{name}: hid_perm must be (N, 16)")
    if gscale is not None and gscale.numel() != D:
        raise NativeError(f"{name}: gscale must hold D = {D} floats")
    head = (_f32(x, name), _f32(g, name), _f32(gld, name), _f32(params, name), params.numel(),
            int(gemm2_steps), _f32(gh_perm, name), _f32(gpre_perm, name))
    tail = (N, D, int(n_bins), C.c_float(float(boundary)), 1 if inverse else 0, _f32(gscale, name), 1 if g_reversed else 0)
    with _device_guard(g):
        if hid_perm is None:
            rc = lib().tfk_rqs_coupling_train_bwd(*head, *tail, _stream(g))
        else:
            rc = lib().tfk_rqs_coupling_train_bwd_hid(*head, _f32(hid_perm, name), *tail, _stream(g))
    calls += 1
    _check(rc, name)


_rows_outer_ws = {}


def _rows_outer_workspace(device, M: int, stream: int):
    """ONE workspace per (device, M) -- about 25 MB at M = 768 -- whatever the stream: every graph-mode ``Flow.fit`` runs
    on a fresh side stream, and a copy per stream pinned up to ~0.8 GB for the life of the process.  The
    buffer is never freed (a captured training step bakes in its address).  A use on ANOTHER stream than the last one
    first waits for that stream's tail, so two streams never hold the partial sums at the same time; only a stream that
    is being captured and finds the buffer last used elsewhere gets a private copy (a capture cannot wait on work outside
    it)."""
    key = (device.index, M)
    entry = _rows_outer_ws.get(key)
    if entry is None:
        entry = _rows_outer_ws[key] = [torch.empty(int(lib().tfk_rows_outer_workspace_bytes(M)) // 4,
                                                   dtype=torch.float32, device=device), stream]
    elif entry[1] != stream:
        if torch.cuda.is_current_stream_capturing():
            pkey = (device.index, M, stream)
            private = _rows_outer_ws.get(pkey)
            if private is None:
                private = _rows_outer_ws[pkey] = [torch.empty_like(entry[0]), stream]
            return private[0]
        prev = torch.cuda.ExternalStream(entry[1], device=device) if entry[1] else torch.cuda.default_stream(device)
        torch.cuda.current_stream(device).wait_stream(prev)
        entry[1] = stream
    return entry[0]


def rows_outer(A, M, B, out):
    """out[M * 16] (accumulator order, include/tfk.h) = sum over the rows of A[n, :M]^T B[n, :16] (tfk_rows_outer):
    the weight-gradient products that contract over the batch rows, deterministic, no GEMM-library call."""
    global calls
    name = "tfk_rows_outer"
    N, lda = _rows(A, name)
    if B.shape != (N, 16) or out.numel() != M * 16 or not out.is_contiguous():
        raise NativeError(f"{name}: B must be (N, 16) and out hold M * 16 = {M * 16} contiguous floats")
    ws = _rows_outer_workspace(A.device, int(M), _stream(A))
    with _device_guard(A):
        rc = lib().tfk_rows_outer(_f32(A, name), int(lda), int(M), _f32(B, name), _f32(out, name), _f32(ws, name), N,
                                  _stream(A))
    calls += 1
    _check(rc, name)


def made_affine_sequential(z, out, logdet, W1t, b1, W2, b2, divide, accumulate=False):
    """The sequential map of a MADE-based affine layer in one launch; W1t (D, HP), b1 (HP,),
    W2 (D, 2, HP), b2 (D, 2) masked and zero-padded to HP hidden units."""
    global calls
    name = "tfk_made_affine_sequential"
    N, D = _rows(z, name)
    HP = b1.numel()
    if out.shape != z.shape or logdet.numel() != N or W1t.numel() != D * HP or W2.numel() != 2 * D * HP \
            or b2.numel() != 2 * D:
        raise NativeError(f"{name}: bad tensor shapes")
    args = (_f32(z, name), _f32(out, name), _f32(logdet, name), N, D, _f32(W1t, name), _f32(b1, name),
            _f32(W2, name), _f32(b2, name), HP, 1 if divide else 0, 1 if accumulate else 0)
    with _device_guard(z):
        rc = lib().tfk_made_affine_sequential(*args, _stream(z))
    calls += 1
    _check(rc, name)


def made_rqs_sequential(z, out, logdet, W1t, b1, W2, b2, n_bins, boundary, accumulate=False, lrs=False):
    """The sequential map of a MADE-based RQ-spline (or, ``lrs``, linear-rational-spline) layer in one
    launch; W1t (D, HP), b1 (HP,), W2 (D, P, HP), b2 (D, P) masked and zero-padded to HP hidden units,
    P = 3 n_bins - 1 (RQ) or 4 n_bins (LRS)."""
    global calls
    name = "tfk_made_lrs_sequential" if lrs else "tfk_made_rqs_sequential"
    N, D = _rows(z, name)
    HP, P = b1.numel(), (4 * int(n_bins) if lrs else 3 * int(n_bins) - 1)
    if out.shape != z.shape or logdet.numel() != N or W1t.numel() != D * HP or W2.numel() != P * D * HP \
            or b2.numel() != P * D:
        raise NativeError(f"{name}: bad tensor shapes")
    args = (_f32(z, name), _f32(out, name), _f32(logdet, name), N, D, _f32(W1t, name), _f32(b1, name),
            _f32(W2, name), _f32(b2, name), HP, int(n_bins), C.c_float(float(boundary)), 1 if accumulate else 0)
    with _device_guard(z):
        rc = getattr(lib(), name)(*args, _stream(z))
    calls += 1
    _check(rc, name)


def conv3x3_relu_pool_affine(x, weight, bias, scale, shift):
    """conv3x3(pad 1) -> ReLU -> MaxPool2d(2) -> per-channel scale / shift in one launch (NCHW)."""
    global calls
    name = "tfk_conv3x3_relu_pool_affine"
    if x.dim() != 4 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] != x.shape[1]:
        raise NativeError(f"{name}: x (N, C, H, W) and weight (C_out, C, 3, 3) expected")
    N, c_in, H, W = x.shape
    c_out = weight.shape[0]
    if bias.numel() != c_out or scale.numel() != c_out or shift.numel() != c_out:
        raise NativeError(f"{name}: bias / scale / shift must hold c_out = {c_out} values")
    out = torch.empty(N, c_out, H // 2, W // 2, dtype=torch.float32, device=x.device)
    args = (_f32(x, name), _f32(weight, name), _f32(bias, name), _f32(scale, name), _f32(shift, name),
            _f32(out, name), N, c_in, c_out, H, W)
    with _device_guard(x):
        rc = lib().tfk_conv3x3_relu_pool_affine(*args, _stream(x))
    calls += 1
    _check(rc, name)
    return out


def bounded_sigmoid(h: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """``lo + (hi - lo) * sigmoid(h)`` in one pass (ATen: three kernels, two temporaries)."""
    global calls
    name = "tfk_bounded_sigmoid"
    out = torch.empty_like(h)
    with _device_guard(h):
        rc = lib().tfk_bounded_sigmoid(_f32(h, name), _f32(out, name), h.numel(), float(lo), float(hi), _stream(h))
    calls += 1
    _check(rc, name)
    return out


def bounded_sigmoid_bwd(out: torch.Tensor, g: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """d(loss)/dh of ``out = lo + (hi - lo) * sigmoid(h)`` from ``out`` and d(loss)/d(out), one pass."""
    global calls
    name = "tfk_bounded_sigmoid_bwd"
    g_in = torch.empty_like(out)
    with _device_guard(out):
        rc = lib().tfk_bounded_sigmoid_bwd(_f32(out, name), _f32(g, name), _f32(g_in, name), out.numel(), float(lo),
                                           float(hi), _stream(out))
    calls += 1
    _check(rc, name)
    return g_in


def conv1x1_frame(x, weight, bias, h_out: int, w_out: int):
    """1x1 convolution c_in -> c_out placed in the middle of an (h_out, w_out) frame that holds the bias.
    ``x`` (N, C, H, W) may be a view whose images are contiguous but further apart than C*H*W."""
    global calls
    name = "tfk_conv1x1_frame"
    if x.dim() != 4 or x.dtype != torch.float32 or x.device.type != "cuda":
        raise NativeError(f"{name}: x must be a float32 (N, C, H, W) tensor on a HIP device")
    N, c_in, H, W = x.shape
    if N > 1 and (x.stride(3) != 1 or x.stride(2) != W or x.stride(1) != H * W):
        raise NativeError(f"{name}: the images of x must be contiguous")
    c_out = weight.shape[0]
    if weight.numel() != c_out * c_in or bias.numel() != c_out:
        raise NativeError(f"{name}: weight (c_out, c_in[, 1, 1]) and bias (c_out,) expected")
    x_stride = x.stride(0) if N > 1 else c_in * H * W
    out = torch.empty(N, c_out, h_out, w_out, dtype=torch.float32, device=x.device)
    with _device_guard(x):
        rc = lib().tfk_conv1x1_frame(x.data_ptr(), x_stride, _f32(weight, name), _f32(bias, name), _f32(out, name),
                                     N, c_in, c_out, H, W, h_out, w_out, _stream(x))
    calls += 1
    _check(rc, name)
    return out


_convnet_ws = {}      # (device index, stream handle) -> zeroed workspace of the tfk_convnet_train_* launches


def convnet_train_workspace(t: torch.Tensor) -> torch.Tensor:
    """The partial-sum rows + ticket counter the ConvNet training launches share, one per (device, stream): zeroed
    once, every launch leaves the counter at zero and consumes its own partial sums before it ends."""
    key = (t.device.index, _stream(t))
    ws = _convnet_ws.get(key)
    if ws is None:
        ws = _convnet_ws[key] = torch.zeros(int(lib().tfk_convnet_train_workspace_bytes()) // 4, dtype=torch.float32,
                                            device=t.device)
    return ws


# floats per sample the two passes of the whole-network calls keep / use
CONVNET_ACT_FLOATS = 4 * 32 * 32 + 8 * 16 * 16 + 8 * 8 * 8 + 4 * 4 * 4 + 16       # a0, y1, y2, y3, a16
CONVNET_AMAX_BYTES = 8 * 16 * 16 + 8 * 8 * 8 + 4 * 4 * 4
CONVNET_SCRATCH_FLOATS = 16 + 64 + 512 + 2048 + 4096                             # g16, gz3, gz2, gz1, g_a0


def convnet_train_forward(plan: "ConvNetTrainPlan", params, bns, x, training: bool, update_running: bool):
    """The whole ConvNet conditioner in ONE call (tfk_convnet_train_forward): fills ``plan`` from ``params`` (the 18
    tensors in convnet_train._params order), the three BatchNorm modules and fresh activation buffers; returns (theta,
    acts, amax) -- the two buffers are what the backward call needs besides ``plan``."""
    global calls
    name = "tfk_convnet_train_forward"
    N, c, h, w = x.shape
    (w_m1, b_m1, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, g3, be3, w_m2, b_m2, w_lin, b_lin) = params
    M = w_lin.shape[0]
    dev = x.device
    acts = torch.empty(N * CONVNET_ACT_FLOATS + 80 + 18 * M, dtype=torch.float32, device=dev)
    amax = torch.empty(N * CONVNET_AMAX_BYTES, dtype=torch.uint8, device=dev)
    theta = torch.empty(N, M, dtype=torch.float32, device=dev)
    p = plan
    p.mod1_w, p.mod1_b, p.mod2_w, p.mod2_b = w_m1.data_ptr(), b_m1.data_ptr(), w_m2.data_ptr(), b_m2.data_ptr()
    p.lin_w, p.lin_b = w_lin.data_ptr(), b_lin.data_ptr()
    for k, (cw, cb, bw, bb, bn) in enumerate(((w1, b1, g1, be1, bns[0]), (w2, b2, g2, be2, bns[1]), (w3, b3, g3, be3, bns[2]))):
        p.conv_w[k], p.conv_b[k], p.bn_w[k], p.bn_b[k] = cw.data_ptr(), cb.data_ptr(), bw.data_ptr(), bb.data_ptr()
        p.bn_mean[k], p.bn_var[k] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        p.bn_count[k] = None if bn.num_batches_tracked is None else bn.num_batches_tracked.data_ptr()
        p.bn_eps[k], p.bn_momentum[k] = float(bn.eps), float(bn.momentum)
    p.c, p.h, p.w, p.kh, p.kw, p.M = c, h, w, w_m1.shape[2], w_m1.shape[3], M
    base, fs = acts.data_ptr(), 4
    off = 0
    p.a0 = base
    off += N * 4096
    for k, n_el in enumerate((2048, 512, 64)):
        p.y[k] = base + off * fs
        off += N * n_el
    p.a16 = base + off * fs
    off += N * 16
    for k, n_el in enumerate((32, 32, 16)):
        p.stats[k] = base + off * fs
        off += n_el
    p.lin_fold = base + off * fs
    ab = amax.data_ptr()
    p.amax[0], p.amax[1], p.amax[2] = ab, ab + N * 2048, ab + N * (2048 + 512)
    p.workspace = convnet_train_workspace(x).data_ptr()
    with _device_guard(x):
        rc = lib().tfk_convnet_train_forward(C.byref(p), _f32(x, name), theta.data_ptr(), N, 1 if training else 0,
                                             1 if update_running else 0, _stream(x))
    calls += 7
    _check(rc, name)
    return theta, acts, amax


def convnet_train_backward(plan: "ConvNetTrainPlan", x, g_theta, training: bool):
    """The reverse pass of the whole network in ONE call (tfk_convnet_train_backward; ``plan`` as the forward call left it,
    its buffers alive): (g_x, bn_out (100 floats), sums) -- the layout of ``sums`` is in include/tfk.h."""
    global calls
    name = "tfk_convnet_train_backward"
    N = x.shape[0]
    p = plan
    n_sums = int(lib().tfk_convnet_train_sums_floats(p.c, p.kh, p.kw, p.M))
    g_x = torch.empty_like(x)
    scratch = torch.empty(N * CONVNET_SCRATCH_FLOATS, dtype=torch.float32, device=x.device)
    out = torch.empty(100 + n_sums, dtype=torch.float32, device=x.device)     # (its slices become the .grad tensors)
    base = out.data_ptr()
    with _device_guard(x):
        rc = lib().tfk_convnet_train_backward(C.byref(p), _f32(x, name), _f32(g_theta, name), g_x.data_ptr(),
                                              scratch.data_ptr(), base, base + 400, N, 1 if training else 0, _stream(x))
    calls += 7
    _check(rc, name)
    return g_x, out[:100], out[100:]


def convnet_train_block_fwd(x, in_affine, weight, bias, bn, training: bool, update_running: bool):
    """One ConvNetBlock (conv3x3 -> ReLU -> MaxPool2d(2)) ahead of its BatchNorm ``bn``: returns (y before
    normalisation, argmax bytes, stats = scale | shift | mean | 1/std of that BatchNorm); ``in_affine``: the scale | shift
    of the BatchNorm in front of the block (applied on load) or None."""
    global calls
    name = "tfk_convnet_train_block_fwd"
    N, c_in, H, W = x.shape
    c_out = weight.shape[0]
    y = torch.empty(N, c_out, H // 2, W // 2, dtype=torch.float32, device=x.device)
    amax = torch.empty(N, c_out, H // 2, W // 2, dtype=torch.uint8, device=x.device)
    stats = torch.empty(4 * c_out, dtype=torch.float32, device=x.device)
    nb = bn.num_batches_tracked
    momentum = 0.0 if bn.momentum is None else float(bn.momentum)
    ws = convnet_train_workspace(x)
    with _device_guard(x):
        rc = lib().tfk_convnet_train_block_fwd(
            _f32(x, name), _f32(in_affine, name), _f32(weight, name), _f32(bias, name), _f32(y, name), amax.data_ptr(),
            _f32(bn.weight, name), _f32(bn.bias, name), _f32(bn.running_mean, name), _f32(bn.running_var, name),
            None if nb is None else nb.data_ptr(), float(bn.eps), momentum, 1 if training else 0,
            1 if update_running else 0, _f32(stats, name), ws.data_ptr(), N, c_in, c_out, H, W, _stream(x))
    calls += 1
    _check(rc, name)
    return y, amax, stats


def convnet_train_block_bwd(gz, coef, y, amax, x, in_affine, weight, bn_stats, bn_training: bool):
    """Reverse mode of the block: returns (g_in, weight gradient, bias gradient, and -- when the input came through a
    BatchNorm with ``bn_stats`` -- (coef, d weight, d bias) of that BatchNorm, else None)."""
    global calls
    name = "tfk_convnet_train_block_bwd"
    N, c_in, H, _ = x.shape
    c_out = weight.shape[0]
    g_in = torch.empty_like(x)
    sums = torch.empty(c_out * c_in * 9 + c_out, dtype=torch.float32, device=x.device)
    bnout = None
    if bn_stats is not None:
        bnout = torch.empty(5 * c_in, dtype=torch.float32, device=x.device)
    ws = convnet_train_workspace(x)
    with _device_guard(x):
        rc = lib().tfk_convnet_train_block_bwd(
            _f32(gz, name), _f32(coef, name), _f32(y, name), amax.data_ptr(), _f32(x, name), _f32(in_affine, name),
            _f32(weight, name), _f32(g_in, name), _f32(sums, name), _f32(bn_stats, name),
            None if bnout is None else bnout.data_ptr(), None if bnout is None else bnout[3 * c_in:].data_ptr(),
            None if bnout is None else bnout[4 * c_in:].data_ptr(), 1 if bn_training else 0, ws.data_ptr(), N, c_in,
            c_out, H, _stream(x))
    calls += 1
    _check(rc, name)
    dW, db = sums[:c_out * c_in * 9].view(c_out, c_in, 3, 3), sums[c_out * c_in * 9:]
    prev = None if bnout is None else (bnout[:3 * c_in], bnout[3 * c_in:4 * c_in], bnout[4 * c_in:])
    return g_in, dW, db, prev


def convnet_train_frame_fwd(x, in_affine, weight, bias, h_out: int, w_out: int):
    """ConvModifier (weight (c_out, c_in, kh, kw), kh / kw 1 or 2), the BatchNorm in front applied on load."""
    global calls
    name = "tfk_convnet_train_frame_fwd"
    N, c_in, H, W = x.shape
    c_out, _, kh, kw = weight.shape
    out = torch.empty(N, c_out, h_out, w_out, dtype=torch.float32, device=x.device)
    with _device_guard(x):
        rc = lib().tfk_convnet_train_frame_fwd(_f32(x, name), _f32(in_affine, name), _f32(weight, name), _f32(bias, name),
                                               _f32(out, name), N, c_in, c_out, H, W, h_out, w_out, kh, kw, _stream(x))
    calls += 1
    _check(rc, name)
    return out


def convnet_train_frame_bwd(g_out, x, in_affine, weight, bn_stats, bn_training: bool):
    """Reverse mode of the ConvModifier: (g_in, weight gradient (c_out, c_in, kh, kw), bias gradient (c_out), BatchNorm
    triple or None as in ``convnet_train_block_bwd``)."""
    global calls
    name = "tfk_convnet_train_frame_bwd"
    N, c_in, H, W = x.shape
    _, c_out, h_out, w_out = g_out.shape
    kh, kw = weight.shape[2:]
    g_in = torch.empty_like(x)
    K0 = c_out * c_in * kh * kw
    K1 = K0 + 2 * c_in
    sums = torch.empty(K1 + c_out, dtype=torch.float32, device=x.device)
    bnout = None
    if bn_stats is not None:
        bnout = torch.empty(5 * c_in, dtype=torch.float32, device=x.device)
    ws = convnet_train_workspace(x)
    with _device_guard(x):
        rc = lib().tfk_convnet_train_frame_bwd(
            _f32(g_out, name), _f32(x, name), _f32(in_affine, name), _f32(weight, name), _f32(g_in, name),
            _f32(sums, name), _f32(bn_stats, name), None if bnout is None else bnout.data_ptr(),
            None if bnout is None else bnout[3 * c_in:].data_ptr(), None if bnout is None else bnout[4 * c_in:].data_ptr(),
            1 if bn_training else 0, ws.data_ptr(), N, c_in, c_out, H, W, h_out, w_out, kh, kw, _stream(x))
    calls += 1
    _check(rc, name)
    prev = None if bnout is None else (bnout[:3 * c_in], bnout[3 * c_in:4 * c_in], bnout[4 * c_in:])
    return g_in, sums[:K0].view(c_out, c_in, kh, kw), sums[K1:], prev


def convnet_train_linear_prep(weight, bias, frame_bias, h_out: int, w_out: int):
    """The Linear layer behind the second ConvModifier folded to its 16 interior inputs: (W16 (M, 16), b_eff (M), w_frame
    (M)); weight (M, h_out * w_out), ``frame_bias`` the modifier's bias (1 element)."""
    global calls
    name = "tfk_convnet_train_linear_prep"
    M = weight.shape[0]
    buf = torch.empty(M * 18, dtype=torch.float32, device=weight.device)
    W16, b_eff, w_frame = buf[:M * 16].view(M, 16), buf[M * 16:M * 17], buf[M * 17:]
    with _device_guard(weight):
        rc = lib().tfk_convnet_train_linear_prep(_f32(weight, name), _f32(bias, name), _f32(frame_bias, name),
                                                 W16.data_ptr(), b_eff.data_ptr(), w_frame.data_ptr(), M, h_out, w_out,
                                                 _stream(weight))
    calls += 1
    _check(rc, name)
    return W16, b_eff, w_frame


def convnet_train_linear_fwd(a16, W16, b_eff):
    """b_eff + a16 (N, 16) W16 (M, 16)^T -> (N, M)."""
    global calls
    name = "tfk_convnet_train_linear_fwd"
    N, M = a16.shape[0], W16.shape[0]
    out = torch.empty(N, M, dtype=torch.float32, device=a16.device)
    with _device_guard(a16):
        rc = lib().tfk_convnet_train_linear_fwd(_f32(a16, name), W16.data_ptr(), b_eff.data_ptr(), _f32(out, name), N, M,
                                                _stream(a16))
    calls += 1
    _check(rc, name)
    return out


def convnet_train_linear_bwd_input(g, W16):
    """g (N, M) W16 (M, 16) -> (N, 16)."""
    global calls
    name = "tfk_convnet_train_linear_bwd_input"
    N, M = g.shape
    g16 = torch.empty(N, 16, dtype=torch.float32, device=g.device)
    with _device_guard(g):
        rc = lib().tfk_convnet_train_linear_bwd_input(_f32(g, name), W16.data_ptr(), _f32(g16, name), N, M, _stream(g))
    calls += 1
    _check(rc, name)
    return g16


def convnet_train_linear_wgrad(g, a16, frame_bias, h_out: int, w_out: int):
    """(dW (M, h_out * w_out), db (M)) of the Linear layer behind the second ConvModifier, whose input equals
    ``frame_bias`` outside the 4 x 4 interior a16 (N, 16)."""
    global calls
    name = "tfk_convnet_train_linear_wgrad"
    N, M = g.shape
    dW = torch.empty(M, h_out * w_out, dtype=torch.float32, device=g.device)
    db = torch.empty(M, dtype=torch.float32, device=g.device)
    with _device_guard(g):
        rc = lib().tfk_convnet_train_linear_wgrad(_f32(g, name), _f32(a16, name), _f32(frame_bias, name), _f32(dW, name),
                                                  _f32(db, name), N, M, h_out, w_out, _stream(g))
    calls += 1
    _check(rc, name)
    return dW, db


def _pack_ops(ops):
    """Host-side op records: int32[8] each = kind, src_plane, H (or GEMM-2 steps), offset, K and
    the fp32 bit patterns of boundary, scale, c (spline ops only).  An array that is already
    packed (``fused.Segment.packed_ops``) passes through."""
    if isinstance(ops, C.Array):
        return ops
    key = tuple(ops)                      # (the training step packs the same few one- and two-op programs every step)
    hit = _OPS_CACHE.get(key)
    if hit is not None:
        return hit
    arr = _pack_ops_uncached(ops)
    if len(_OPS_CACHE) > 256:
        _OPS_CACHE.clear()
    _OPS_CACHE[key] = arr
    return arr


_OPS_CACHE = {}


def _pack_ops_uncached(ops):
    flat = []
    for op in ops:
        kind, plane, H, off = (int(v) for v in op[:4])
        K = int(op[4]) if len(op) > 4 else 0
        fl = [float(v) for v in op[5:8]] + [0.0] * (3 - len(op[5:8]))
        bits = struct.unpack("<3i", struct.pack("<3f", *fl))
        flat += [kind, plane, H, off, K, *bits]
    return (_i32 * max(len(flat), 1))(*flat)


def _n_ops(ops) -> int:
    return len(ops) // 8 if isinstance(ops, C.Array) else len(ops)


def flow_run(x, z, logdet, gauss_loc, gauss_log_scale, logprob, ops, params, accumulate=False):
    """Fused flow program (tfk_flow_run).  ops: list of (kind, src_plane, H, offset[, K,
    boundary, scale, c]) tuples (host side), params: fp32 device block.  z / logdet / logprob
    may be None."""
    global calls
    name = "tfk_flow_run"
    N, D = _rows(x, name)
    ops_arr = _pack_ops(ops)
    for t, n in ((z, N * D), (logdet, N), (logprob, N), (gauss_loc, D), (gauss_log_scale, D)):
        if t is not None and t.numel() != n:
            raise NativeError(f"{name}: tensor with {t.numel()} elements, expected {n}")
    args = (_f32(x, name), _f32(z, name), _f32(logdet, name), _f32(gauss_loc, name),
            _f32(gauss_log_scale, name), _f32(logprob, name), N, D, ops_arr, _n_ops(ops),
            _f32(params, name), params.numel(), 1 if accumulate else 0)
    with _device_guard(x):
        rc = lib().tfk_flow_run(*args, _stream(x))
    calls += 1
    _check(rc, name)


def _stream_flag() -> int:
    """Flag bit 3 of tfk_flow_run_mfma: stream the operands of an affine / shift chain (D >= 128) even if they fit the LDS
    (tuning: TORCHFLOWS_AMD_DEBUG=stream_chain=force)."""
    from torchflows_amd.utils import debug_switch
    return 8 if debug_switch("stream_chain", "1") == "force" else 0


_sum_ws = {}      # (device index, stream handle) -> zero-initialised workspace of tfk_flow_run_mfma_sum (resets itself)


def flow_sum_workspace(t: torch.Tensor) -> torch.Tensor:
    key = (t.device.index, _stream(t))
    ws = _sum_ws.get(key)
    if ws is None:
        ws = _sum_ws[key] = torch.zeros(int(lib().tfk_flow_sum_workspace_bytes()) // 8, dtype=torch.float64, device=t.device)
    return ws


def flow_run_mfma(x, z, logdet, gauss_loc, gauss_log_scale, logprob, ops, params, accumulate=False,
                  reverse_out=False, base_of_input=False, D=None, context=None, sum_out=None):
    """Fused flow program with the conditioner GEMMs on the matrix cores (tfk_flow_run_mfma).
    ops: list of (kind, src_plane, gemm2_steps, offset); params packed by fused._pack_mfma.
    ``D``: the kernel's row width when ``x`` is narrower (lean programs, tfk_flow_run_mfma_in): x (N, x_width) is read
    half by half into the heads of the two planes."""
    global calls
    name = "tfk_flow_run_mfma"
    N, xw = _rows(x, name)
    if sum_out is not None:               # lean program + the fp64 sum of its log-probabilities (tfk_flow_run_mfma_sum)
        name = "tfk_flow_run_mfma_sum"
        Dk = xw if D is None else D
        ops_arr = _pack_ops(ops)
        if context is not None or logprob is None or sum_out.dtype != torch.float64 or sum_out.numel() != 1:
            raise NativeError(f"{name}: needs logprob, a 1-element float64 sum_out and no context")
        for t, n in ((z, N * Dk), (logdet, N), (logprob, N), (gauss_loc, Dk), (gauss_log_scale, Dk)):
            if t is not None and t.numel() != n:
                raise NativeError(f"{name}: tensor with {t.numel()} elements, expected {n}")
        ws = flow_sum_workspace(x)
        args = (_f32(x, name), xw, _f32(z, name), _f32(logdet, name), _f32(gauss_loc, name),
                _f32(gauss_log_scale, name), _f32(logprob, name), N, Dk, ops_arr, _n_ops(ops),
                _f32(params, name), params.numel(),
                (1 if accumulate else 0) | (2 if reverse_out else 0) | (4 if base_of_input else 0) | _stream_flag(),
                ws.data_ptr(), sum_out.data_ptr())
        with _device_guard(x):
            rc = lib().tfk_flow_run_mfma_sum(*args, _stream(x))
        calls += 1
        _check(rc, name)
        return
    if context is not None:               # context-conditioned program (tfk_flow_run_mfma_ctx); rows at full width
        name = "tfk_flow_run_mfma_ctx"
        Dk = xw
        ops_arr = _pack_ops(ops)
        if context.dim() != 2 or context.shape[0] != N:
            raise NativeError(f"{name}: context must be (N, C), got {tuple(context.shape)}")
        for t, n in ((z, N * Dk), (logdet, N), (logprob, N), (gauss_loc, Dk), (gauss_log_scale, Dk)):
            if t is not None and t.numel() != n:
                raise NativeError(f"{name}: tensor with {t.numel()} elements, expected {n}")
        args = (_f32(x, name), _f32(context, name), int(context.shape[1]), _f32(z, name), _f32(logdet, name),
                _f32(gauss_loc, name), _f32(gauss_log_scale, name), _f32(logprob, name), N, Dk, ops_arr, _n_ops(ops),
                _f32(params, name), params.numel(),
                (1 if accumulate else 0) | (2 if reverse_out else 0) | (4 if base_of_input else 0) | _stream_flag())
        with _device_guard(x):
            rc = lib().tfk_flow_run_mfma_ctx(*args, _stream(x))
        calls += 1
        _check(rc, name)
        return
    if D is not None and D != xw:
        name = "tfk_flow_run_mfma_in"
        ops_arr = _pack_ops(ops)
        for t, n in ((z, N * D), (logdet, N), (logprob, N), (gauss_loc, D), (gauss_log_scale, D)):
            if t is not None and t.numel() != n:
                raise NativeError(f"{name}: tensor with {t.numel()} elements, expected {n}")
        args = (_f32(x, name), xw, _f32(z, name), _f32(logdet, name), _f32(gauss_loc, name),
                _f32(gauss_log_scale, name), _f32(logprob, name), N, D, ops_arr, _n_ops(ops),
                _f32(params, name), params.numel(),
                (1 if accumulate else 0) | (2 if reverse_out else 0) | (4 if base_of_input else 0) | _stream_flag())
        with _device_guard(x):
            rc = lib().tfk_flow_run_mfma_in(*args, _stream(x))
        calls += 1
        _check(rc, name)
        return
    D = xw
    ops_arr = _pack_ops(ops)
    for t, n in ((z, N * D), (logdet, N), (logprob, N), (gauss_loc, D), (gauss_log_scale, D)):
        if t is not None and t.numel() != n:
            raise NativeError(f"{name}: tensor with {t.numel()} elements, expected {n}")
    args = (_f32(x, name), _f32(z, name), _f32(logdet, name), _f32(gauss_loc, name),
            _f32(gauss_log_scale, name), _f32(logprob, name), N, D, ops_arr, _n_ops(ops),
            _f32(params, name), params.numel(),
            (1 if accumulate else 0) | (2 if reverse_out else 0) | (4 if base_of_input else 0) | _stream_flag())
    with _device_guard(x):
        rc = lib().tfk_flow_run_mfma(*args, _stream(x))
    calls += 1
    _check(rc, name)


def glow_plan(layer: GlowLayer, D: int) -> dict:
    """The launch shape libtfk picks for this layer (or the overrides, validated)."""
    vals = [_i32(0) for _ in range(6)]
    rc = lib().tfk_glow_plan(C.byref(layer), D, *[C.byref(v) for v in vals])
    _check(rc, "tfk_glow_plan")
    return dict(zip(("slots", "block", "cg1", "cg2", "lds_bytes", "tile_rows"), (int(v.value) for v in vals)))


def glow_coupling(rows: torch.Tensor, logdet: torch.Tensor, layer: GlowLayer, inverse: bool = False) -> None:
    """One convolutional coupling in place on the rows (conditioner, bounded output, transform, log-det added)."""
    global calls
    N, D = _rows(rows, "tfk_glow_coupling")
    if logdet.numel() != N:
        raise NativeError(f"tfk_glow_coupling: logdet has {logdet.numel()} elements, expected {N}")
    with _device_guard(rows):
        rc = lib().tfk_glow_coupling(_f32(rows, "tfk_glow_coupling"), _f32(logdet, "tfk_glow_coupling"), N, D,
                                     C.byref(layer), 1 if inverse else 0, _stream(rows))
    calls += 1
    _check(rc, "tfk_glow_coupling")


def glow_level_pack(steps, D: int, D_level: int, samples: int = 0, block: int = 0) -> torch.Tensor:
    """The host copy of a level program's blob (uint8 CPU tensor): launch shape, geometry and background cell lists of
    ``steps`` (a ctypes array of GlowLevelStep whose pointers are DEVICE pointers).  Pure host work, no GPU needed."""
    n = len(steps)
    nbytes = int(lib().tfk_glow_level_blob_bytes(steps, n, int(D), int(D_level), int(samples), int(block)))
    if nbytes <= 0:
        _check(-nbytes if nbytes < 0 else 1, "tfk_glow_level_blob_bytes")
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    rc = lib().tfk_glow_level_pack(steps, n, int(D), int(D_level), int(samples), int(block), blob.data_ptr(), nbytes)
    _check(rc, "tfk_glow_level_pack")
    return blob


def glow_level_info(blob_host: torch.Tensor) -> dict:
    vals = [_i32(0) for _ in range(4)]
    rc = lib().tfk_glow_level_info(blob_host.data_ptr(), *[C.byref(v) for v in vals])
    _check(rc, "tfk_glow_level_info")
    return dict(zip(("samples", "block", "lds_bytes", "wgs_per_cu"), (int(v.value) for v in vals)))


def glow_level(rows_in: torch.Tensor, rows_out: torch.Tensor, logdet: torch.Tensor, row_idx: Optional[torch.Tensor],
               blob_host: torch.Tensor, blob_dev: torch.Tensor) -> None:
    """The couplings of one level back to back on rows held in the LDS (in place unless the level covers the whole row)."""
    global calls
    name = "tfk_glow_level"
    N, D = _rows(rows_out, name)
    if rows_in.shape != rows_out.shape or logdet.numel() != N:
        raise NativeError(f"{name}: rows_in {tuple(rows_in.shape)}, rows_out {tuple(rows_out.shape)}, logdet {logdet.numel()}")
    if row_idx is not None and (row_idx.dtype != torch.int32 or row_idx.device != rows_out.device):
        raise NativeError(f"{name}: row_idx must be an int32 tensor on the rows' device")
    if blob_host.device.type != "cpu" or blob_dev.device != rows_out.device:
        raise NativeError(f"{name}: the blob's two copies must live on the host and on the rows' device")
    with _device_guard(rows_out):
        rc = lib().tfk_glow_level(_f32(rows_in, name), _f32(rows_out, name), _f32(logdet, name), N, D,
                                  None if row_idx is None else row_idx.data_ptr(), blob_host.data_ptr(),
                                  blob_dev.data_ptr(), _stream(rows_out))
    calls += 1
    _check(rc, name)


def rows_fma(rows: torch.Tensor, st: torch.Tensor) -> None:
    """rows[n, d] = st[d, 0] * rows[n, d] + st[d, 1] in place."""
    global calls
    N, D = _rows(rows, "tfk_rows_fma")
    if st.numel() != 2 * D:
        raise NativeError(f"tfk_rows_fma: st has {st.numel()} elements, expected 2 * {D}")
    with _device_guard(rows):
        rc = lib().tfk_rows_fma(_f32(rows, "tfk_rows_fma"), _f32(st, "tfk_rows_fma"), N, D, _stream(rows))
    calls += 1
    _check(rc, "tfk_rows_fma")


def rows_fma_gauss_logprob(rows, st, loc, log_scale, logdet_in, out) -> None:
    """out[n] = base log-density of st[d, 0] * rows[n, d] + st[d, 1] (+ logdet_in[n]); the rows are only read."""
    global calls
    name = "tfk_rows_fma_gauss_logprob"
    N, D = _rows(rows, name)
    if st.numel() != 2 * D or loc.numel() != D or log_scale.numel() != D or out.numel() != N:
        raise NativeError(f"{name}: bad parameter / out shape")
    with _device_guard(rows):
        rc = lib().tfk_rows_fma_gauss_logprob(_f32(rows, name), _f32(st, name), _f32(loc, name), _f32(log_scale, name),
                                              _f32(logdet_in, name), _f32(out, name), N, D, _stream(rows))
    calls += 1
    _check(rc, name)


def sum_f32(values: torch.Tensor) -> torch.Tensor:
    """fp64 sum of an fp32 vector as a 1-element float64 device tensor (feeds the all-reduce)."""
    global calls
    v = values.reshape(-1)
    N = v.numel()
    _f32(v, "tfk_sum_f32")
    out = torch.empty(1, dtype=torch.float64, device=v.device)
    if N < (1 << 14):                 # one workgroup, no workspace (SURVEY 8(b)'s four-argument form)
        with _device_guard(v):
            rc = lib().tfk_sum_f32(_f32(v, "tfk_sum_f32"), out.data_ptr(), N, _stream(v))
        calls += 1
        _check(rc, "tfk_sum_f32")
        return out
    ws = torch.empty(int(lib().tfk_sum_workspace_bytes(N)), dtype=torch.uint8, device=v.device)
    with _device_guard(v):
        rc = lib().tfk_sum_f32_ws(_f32(v, "tfk_sum_f32_ws"), out.data_ptr(), ws.data_ptr(), N, _stream(v))
    calls += 1
    _check(rc, "tfk_sum_f32_ws")
    return out


def device_info():
    name = C.create_string_buffer(256)
    cus = _i32(0)
    rc = lib().tfk_device_info(name, 256, C.byref(cus))
    _check(rc, "tfk_device_info")
    return name.value.decode(), int(cus.value)
