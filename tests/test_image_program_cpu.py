"""The host-side compiler of the image flows (torchflows_amd/image_program.py: squeeze / chunk folded into index
tables, deferred ActNorm layers, folded conditioner constants, MFMA tile order) checked WITHOUT a GPU: the compiled
program is decoded by the float64 emulator of the kernel's documented semantics (tests/glow_emulator.py) and compared
with the reference's own outputs (tests/golden/flow_glow_3x32x32.npz, flow_glow_3x8x8.npz) and with this package's
ATen composite path.  The kernel itself is compared with the same emulator and fixtures in tests/test_gpu_image.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import glow_emulator as ge


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.fixture(scope="module")
def libtfk():
    from torchflows_amd import native
    if not native.available():
        pytest.skip("libtfk.so not built")
    return native


def test_glow32_program_matches_reference_fp64(libtfk):
    """Config 5's model: 19 launches + one flush; against the REFERENCE evaluated in float64 the decoded program is
    exact up to the fp32 rounding of the folded constants (measured 3.5e-7 on z, 1.6e-9 on the log-det)."""
    from golden_util import load_glow32
    from torchflows_amd import image_program
    flow, fx = load_glow32()
    prog = image_program.compile_program(flow.bijection, 0, torch.device("cpu"))
    assert prog is not None and len(prog.steps) == 19 and prog.flush is not None
    kinds = [s.info["kind"] for s in prog.steps]
    assert kinds.count("conv1x1") == 3 and kinds[3:9:2] == ["conv1x1"] * 3       # only the top block's
    # (the emulator is a float64 python loop: 8 standard rows, 4 of the x 4 and 4 of the x 0.01 stress rows; z64 is stored
    # for the first 8 rows, the log-det's fp64 value for all of them)
    pick = list(range(8)) + [32, 33, 40, 47] + [48, 49, 56, 63]
    z, ld = ge.run_program(prog, torch.from_numpy(fx["x"][pick]))
    assert rel(z.numpy()[:8], fx["z64"]) < 2e-6 and rel(ld.numpy(), fx["log_det64"][pick]) < 1e-7
    assert rel(z.numpy(), fx["z"][pick]) < 1e-5
    inv = image_program.compile_program(flow.bijection, 1, torch.device("cpu"))
    assert [s.inverse for s in inv.steps] == [True] * 19
    x, ldi = ge.run_program(inv, torch.from_numpy(fx["z_in"][pick]))
    assert rel(x.numpy(), fx["x_inv"][pick]) < 5e-6 and rel(ldi.numpy(), fx["log_det_inv64"][pick]) < 1e-7
    # the index tables of a step partition the positions the layer touches; targets are listed in ascending order
    for s in prog.steps:
        src, tgt = s.keep[0][: s.layer.c_in * s.layer.hi * s.layer.wi], s.keep[2][: s.layer.T]
        assert not set(src.tolist()) & set(tgt.tolist())
        if s.layer.kind == 0:
            assert bool((tgt[1:] > tgt[:-1]).all())


def test_small_glow_program_matches_fixture_and_host(libtfk):
    import torchflows_amd as tfa
    from torchflows_amd import image_program
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    fx = load_golden("flow_glow_3x8x8.npz")
    flow = tfa.Flow(AffineGlow((3, 8, 8), n_layers=2))
    flow.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")})
    flow.eval()
    prog = image_program.compile_program(flow.bijection, 0, torch.device("cpu"))
    assert prog is not None
    z, ld = ge.run_program(prog, torch.from_numpy(fx["x"]))
    assert rel(z.numpy(), fx["z"]) < 2e-5 and rel(ld.numpy(), fx["log_det"]) < 2e-5
    with torch.no_grad():                               # the package's own ATen path in float64: same weights
        z64, ld64 = flow.double().bijection.forward(torch.from_numpy(fx["x"]).double())
    assert rel(z.numpy(), z64.numpy()) < 2e-6 and rel(ld.numpy(), ld64.numpy()) < 1e-6


@pytest.mark.parametrize("event_shape,n_layers", [((3, 16, 16), None), ((2, 32, 32), 2), ((4, 8, 16), 1), ((6, 16, 16), 2),
                                                  ((5, 8, 8), 1), ((1, 28, 28), None), ((3, 14, 30), 1)])
def test_program_vs_host_other_shapes(libtfk, event_shape, n_layers):
    import torchflows_amd as tfa
    from torchflows_amd import image_program
    from torchflows_amd.bijections.finite.multiscale import AffineGlow, MultiscaleNICE, MultiscaleRealNVP, ShiftGlow
    torch.manual_seed(3)
    cls = {2: MultiscaleRealNVP, 6: ShiftGlow, 5: MultiscaleNICE, 1: MultiscaleRealNVP}.get(event_shape[0], AffineGlow)
    flow = tfa.Flow(cls(event_shape, n_layers=n_layers))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(32, *event_shape))                 # ActNorm init, BatchNorm statistics
    flow.eval()
    x = torch.randn(5, *event_shape)
    for d in (0, 1):
        prog = image_program.compile_program(flow.bijection, d, torch.device("cpu"))
        assert prog is not None
        out, ld = ge.run_program(prog, x)
        with torch.no_grad():
            f64 = flow.double()
            ref, ld_ref = (f64.bijection.forward if d == 0 else f64.bijection.inverse)(x.double())
            flow.float()
        assert rel(out.numpy(), ref.numpy()) < 5e-6 and rel(ld.numpy(), ld_ref.numpy()) < 3e-6, (d, event_shape)   # (fp32-rounded packed constants)


def test_program_declines_what_it_does_not_cover(libtfk):
    import torchflows_amd as tfa
    from torchflows_amd import image_program
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    cpu = torch.device("cpu")
    torch.manual_seed(0)
    # (odd paddings -- a ConvModifier with a 2-wide kernel, classic.py:26-27, as in MNIST-shaped models -- ARE covered)
    assert image_program.compile_program(AffineGlow((1, 28, 28)).eval(), 0, cpu) is not None
    # images beyond the 32x32 frame: the modifier is a real convolution
    assert image_program.compile_program(AffineGlow((3, 64, 64)).eval(), 0, cpu) is None
    # a transformer without a fused kernel
    from torchflows_amd.bijections.finite.autoregressive.transformers.spline.rational_quadratic import RationalQuadratic
    from torchflows_amd.bijections.finite.multiscale.base import MultiscaleBijection
    spline = MultiscaleBijection((3, 16, 16), transformer_class=RationalQuadratic, n_blocks=2).eval()
    assert image_program.compile_program(spline, 0, cpu) is None
    # ActNorm before its first batch, BatchNorm in training mode
    fresh = AffineGlow((3, 16, 16))
    assert fresh.training and image_program.compile_program(fresh, 0, cpu) is None
    ok = AffineGlow((3, 16, 16)).eval()
    assert image_program.compile_program(ok, 0, cpu) is not None
    ok.checkerboard_layers[0].invert()
    assert image_program.compile_program(ok, 0, cpu) is None


def test_level_programs_pack_on_the_host(libtfk, monkeypatch):
    """The opt-in level route (tfk_glow_level): config 5's 19 couplings group into three levels (9 on all 3 072 elements,
    6 on 1 536, 4 on 768 -- multiscale/base.py:249-296), the inverse program in the opposite order; the blobs are packed
    by libtfk on the host (launch shape, geometry, background cell lists, a copy of the conv weights), no GPU needed."""
    from golden_util import load_glow32
    from torchflows_amd import image_program, native
    monkeypatch.setenv("TORCHFLOWS_AMD_GLOW_LEVELS", "1")
    flow, _ = load_glow32()
    fwd = image_program.compile_program(flow.bijection, 0, torch.device("cpu"))
    inv = image_program.compile_program(flow.bijection, 1, torch.device("cpu"))
    assert [(lv.count, lv.D_level) for lv in fwd.levels] == [(9, 3072), (6, 1536), (4, 768)]
    assert [(lv.count, lv.D_level) for lv in inv.levels] == [(4, 768), (6, 1536), (9, 3072)]
    assert fwd.levels[0].row_idx is None and fwd.levels[1].row_idx.numel() == 1536
    for lv in fwd.levels:
        info = native.glow_level_info(lv.blob_host)
        assert info["samples"] == 4 and info["lds_bytes"] <= 160 * 1024 and info["wgs_per_cu"] >= 1
        idx = lv.row_idx
        if idx is not None:
            assert bool((idx[1:] > idx[:-1]).all())
    monkeypatch.setenv("TORCHFLOWS_AMD_GLOW_LEVELS", "0")
    assert image_program.compile_program(flow.bijection, 0, torch.device("cpu")).levels is None
