"""Image / multiscale path on the HIP kernels: the 1x1
convolution kernel against the oracle, masked couplings / squeeze / ActNorm on images and a
whole Glow against the reference's golden outputs and against the package's own ATen path."""
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
    assert torch.cuda.is_available()
    from torchflows_amd import native as nat
    nat.lib()
    return nat


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.mark.parametrize("N,C,n,HW", [(64, 12, 6, 256), (33, 4, 2, 16), (5, 24, 12, 64), (7, 3, 2, 9),
                                      (9, 32, 16, 4), (1000, 2, 1, 1)])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_conv1x1_coupling_vs_oracle(native, oracle, N, C, n, HW, masked, inverse):
    """rows = images (C, HW) channel-major; the target is n whole channels."""
    rng = np.random.default_rng(N + C + n)
    D, T = C * HW, n * HW
    x = rng.standard_normal((N, D)).astype(np.float32)
    h = rng.standard_normal((N, n + n * (n - 1))).astype(np.float32)
    if masked:          # the FIRST n channels are the target (channel-wise split, inverted)
        tgt = np.arange(0, T, dtype=np.int32)
    else:               # the last n channels (contiguous tail)
        tgt = np.arange(D - T, D, dtype=np.int32)
    yb, ld = oracle.conv1x1(x[:, tgt].reshape(N, n, HW), h, inverse=inverse)
    expect = x.copy()
    expect[:, tgt] = yb.reshape(N, T)
    out = torch.full((N, D), float("nan"), device="cuda")
    logdet = torch.full((N,), float("nan"), device="cuda")
    tgt_d = dev(tgt, torch.int32) if masked else None
    native.conv1x1_coupling(dev(x), dev(h), out, logdet, tgt_d, T, n, accumulate=False, inverse=inverse)
    tol = 1e-5 if not inverse else 1e-4       # the triangular solves amplify rounding by cond(LU)
    assert rel(out.cpu().numpy(), expect) < tol
    assert rel(logdet.cpu().numpy(), ld) < 1e-5
    untouched = np.setdiff1d(np.arange(D), tgt)
    assert np.array_equal(out.cpu().numpy()[:, untouched], x[:, untouched])
    run = rng.standard_normal(N).astype(np.float32)
    buf, logdet2 = dev(x), dev(run)
    native.conv1x1_coupling(buf, dev(h), buf, logdet2, tgt_d, T, n, accumulate=True, inverse=inverse)
    assert torch.equal(buf, out)
    assert rel(logdet2.cpu().numpy(), run + ld) < 1e-5
    with pytest.raises(native.NativeError):
        native.conv1x1_coupling(dev(x), torch.zeros(N, 17 * 17, device="cuda"), out, logdet, None, T, 17)


def test_image_layers_golden_on_hip(native):
    from torchflows_amd.bijections.finite.autoregressive.transformers.linear.affine import Affine
    from torchflows_amd.bijections.finite.multiscale import (
        ChannelWiseCoupling, CheckerboardCoupling, Invertible1x1ConvolutionalCoupling)
    fx = load_golden("image_layers.npz")
    layers = {
        "ckb": lambda: CheckerboardCoupling((3, 8, 8), Affine),
        "ckb_alt": lambda: CheckerboardCoupling((3, 8, 8), Affine, alternate=True),
        "chw": lambda: ChannelWiseCoupling((4, 4, 4), Affine),
        "chw_alt": lambda: ChannelWiseCoupling((4, 4, 4), Affine, alternate=True),
        "c1x1": lambda: Invertible1x1ConvolutionalCoupling((4, 4, 4)),
    }
    for tag, make in layers.items():
        layer = make().eval()
        pre = f"{tag}_sd/"
        layer.load_state_dict({k[len(pre):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(pre)})
        layer = layer.cuda()
        x = torch.from_numpy(fx[f"{tag}_x"]).cuda()
        before = native.calls
        with torch.no_grad():
            z, ld = layer.forward(x)
            xi, ldi = layer.inverse(x)
        # one coupling kernel per call (+ the ConvNet conditioner's launches each: three fused blocks, up to
        # two 1x1 modifiers, the bounded-output squash)
        assert 2 <= native.calls - before <= 14, tag
        assert rel(z.cpu().numpy(), fx[f"{tag}_z"]) < 1e-5, tag
        assert rel(ld.cpu().numpy(), fx[f"{tag}_ld"]) < 1e-5, tag
        assert rel(xi.cpu().numpy(), fx[f"{tag}_xinv"]) < 1e-4, tag
        assert rel(ldi.cpu().numpy(), fx[f"{tag}_ldinv"]) < 1e-5, tag


def test_affine_glow_golden_on_hip(native):
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    fx = load_golden("flow_glow_3x8x8.npz")
    flow = tfa.Flow(AffineGlow((3, 8, 8), n_layers=2))
    flow.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")})
    flow = flow.cuda().eval()
    before = native.calls
    with torch.no_grad():
        z, ld = flow.bijection.forward(torch.from_numpy(fx["x"]).cuda())
        lp = flow.log_prob(torch.from_numpy(fx["x"]).cuda())
        xr, ldr = flow.bijection.inverse(torch.from_numpy(fx["z_in"]).cuda())
    assert native.calls - before > 30
    assert rel(z.cpu().numpy(), fx["z"]) < 2e-5 and rel(ld.cpu().numpy(), fx["log_det"]) < 2e-5
    assert rel(lp.cpu().numpy(), fx["log_prob"]) < 1e-5
    assert rel(xr.cpu().numpy(), fx["x_inv"]) < 1e-4 and rel(ldr.cpu().numpy(), fx["log_det_inv"]) < 2e-5


def test_affine_glow_config5_golden_on_hip(native):
    """Config 5 as configured -- AffineGlow((3, 32, 32)), 3.2 M parameters -- on the HIP path against the
    REFERENCE's outputs (tests/golden/flow_glow_3x32x32.npz): log_prob within 1e-5, z / log-det / x within 2e-5
    (the reference's own fp32-vs-fp64 distance on these rows is printed beside the error)."""
    from golden_util import load_glow32
    flow, fx = load_glow32()
    flow = flow.cuda()
    x, z_in = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["z_in"]).cuda()
    before = native.calls
    with torch.no_grad():
        lp = flow.log_prob(x)
        z, ld = flow.bijection.forward(x)
        xr, ldr = flow.bijection.inverse(z_in)
    # one launch per coupling (19) + the flush of the deferred ActNorm maps; log_prob: the flush and the base density are ONE
    # read-only pass (tfk_rows_fma_gauss_logprob, round 4), z is never written
    assert native.calls - before == 20 + 20 + 20
    e = dict(log_prob=rel(lp.cpu().numpy(), fx["log_prob"]), z=rel(z.cpu().numpy(), fx["z"]),
             log_det=rel(ld.cpu().numpy(), fx["log_det"]), x_inv=rel(xr.cpu().numpy(), fx["x_inv"]),
             log_det_inv=rel(ldr.cpu().numpy(), fx["log_det_inv"]))
    floor = dict(log_prob=rel(fx["log_prob"], fx["log_prob64"]), z=float(fx["floor_z"]),
                 log_det=rel(fx["log_det"], fx["log_det64"]), x_inv=float(fx["floor_x_inv"]),
                 log_det_inv=rel(fx["log_det_inv"], fx["log_det_inv64"]))
    # per group of rows: 0..31 standard normal, 32..47 scaled x 4 (the conditioner's sigmoid bound), 48..63 scaled x 0.01
    groups = {"std": slice(0, 32), "x4": slice(32, 48), "x0.01": slice(48, 64)}
    per = {g: dict(log_prob=rel(lp.cpu().numpy()[s], fx["log_prob"][s]), z=rel(z.cpu().numpy()[s], fx["z"][s]))
           for g, s in groups.items()}
    print("glow32 vs reference (64 rows):", e, "reference fp32-vs-fp64:", floor, "per row group:", per)
    # < 1e-5 on ALL of them, stress rows included (round 3: 8 standard rows, z / x at 2e-5)
    assert max(e.values()) < 1e-5, e


def test_affine_glow_config5_full_size_properties(native):
    """Config 5 at its full size, N = 2^18 rows of (3, 32, 32) (3 GiB) evaluated in chunks: the first 64 rows ARE
    the fixture's (reference log_prob within 1e-5), chunk-size invariance, round trip, ld_fwd = -ld_inv, and the
    fp64 sum of the log-likelihood against a host fp64 sum."""
    from golden_util import load_glow32
    from torchflows_amd.distributed import sharded_log_likelihood
    flow, fx = load_glow32()
    flow = flow.cuda()
    N = 1 << 18
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn(N, 3, 32, 32, device="cuda", generator=g)
    n_fx = fx["x"].shape[0]                                       # the fixture's 64 rows, stress rows included
    x[:n_fx] = torch.from_numpy(fx["x"]).cuda()
    with torch.no_grad():
        lp, total = sharded_log_likelihood(flow, x, chunk_rows=1 << 13)
        assert rel(lp[:n_fx].cpu().numpy(), fx["log_prob"]) < 1e-5
        assert torch.isfinite(lp).all()
        host_sum = float(lp.cpu().double().sum())
        assert abs(float(total) - host_sum) <= 1e-9 * abs(host_sum)
        # chunk invariance: rows are independent, so a differently chunked pass gives the same values
        sel = torch.arange(0, N, 37, device="cuda")[:5000]
        lp_b = flow.log_prob(x[sel])
        assert rel(lp_b.cpu().numpy(), lp[sel].cpu().numpy()) < 2e-6
        # round trip on a strided subset
        z, ld = flow.bijection.forward(x[sel])
        xr, ldr = flow.bijection.inverse(z)
        assert float((xr - x[sel]).abs().max()) < 1e-3            # the reference's own round-trip bar
        assert rel((-ldr).cpu().numpy(), ld.cpu().numpy()) < 2e-5


def _clone_layer(native, layer, **kw):
    import ctypes as C
    new = native.GlowLayer()
    C.memmove(C.byref(new), C.byref(layer), C.sizeof(layer))
    for k, v in kw.items():
        setattr(new, k, v)
    return new


@pytest.mark.parametrize("direction", [0, 1])
def test_glow_coupling_kernel_vs_emulator(native, direction):
    """tfk_glow_coupling, launch by launch, against the float64 emulator of its documented semantics
    (tests/glow_emulator.py) on config 5's 19 couplings: 37 rows (a ragged last tile), both directions, the default
    launch shape and -- for each distinct geometry -- other slots / block / channel-group choices (every template
    path of the conv stages).  Targets within 1e-5, untouched elements bit-identical, log-det within 1e-5."""
    import glow_emulator as ge
    from golden_util import load_glow32
    from torchflows_amd import image_program
    flow, fx = load_glow32()
    flow = flow.cuda()
    prog = image_program.get_program(flow.bijection, direction, torch.device("cuda", 0))
    assert prog is not None and len(prog.steps) == 19
    torch.manual_seed(11)
    N = 37
    rows = torch.randn(N, prog.D) * 1.5
    shapes = [dict(), dict(slots=1, block=256, cg1=8, cg2=8), dict(slots=3, block=512, cg1=2, cg2=2),
              dict(slots=4, block=1024, cg1=4, cg2=8)]
    seen = set()
    for step in prog.steps:
        key = (step.info["kind"], step.info["image"])
        variants = shapes if key not in seen else shapes[:1]
        seen.add(key)
        ref = rows.double().clone()
        ld_ref = torch.zeros(N, dtype=torch.float64)
        ge.run_step(ref, ld_ref, step)
        tgt = step.keep[2][: step.layer.T].long().cpu()
        untouched = torch.ones(prog.D, dtype=torch.bool)
        untouched[tgt] = False
        for kw in variants:
            layer = _clone_layer(native, step.layer, **kw) if kw else step.layer
            try:
                native.glow_plan(layer, prog.D)
            except native.NativeError:
                continue                                 # this shape does not fit the LDS for this geometry
            out = rows.cuda()
            ld = torch.full((N,), 0.25, device="cuda")
            native.glow_coupling(out, ld, layer, step.inverse)
            out, ld = out.cpu(), ld.cpu()
            assert torch.equal(out[:, untouched], rows[:, untouched]), (key, kw)
            assert rel(out[:, tgt].numpy(), ref[:, tgt].numpy()) < 1e-5, (key, kw)
            assert rel(ld.numpy() - 0.25, ld_ref.numpy()) < 1e-5, (key, kw)
        rows = ref.float()                               # feed the next layer what this one produced


def test_glow_program_rows_and_batch_shapes(native):
    """Ragged row counts around the 16-row tiles, a batch shape of rank 2, inputs left untouched, and the program
    against the package's own layer-by-layer HIP route (TORCHFLOWS_AMD_IMAGE_PROGRAM=0 semantics via the ATen host path)."""
    import glow_emulator as ge
    from golden_util import load_glow32
    from torchflows_amd import image_program
    flow, fx = load_glow32()
    flow = flow.cuda()
    prog = image_program.get_program(flow.bijection, 0, torch.device("cuda", 0))
    torch.manual_seed(5)
    for n in (1, 15, 16, 17, 250):
        x = torch.randn(n, 3, 32, 32)
        xd = x.cuda()
        keep = xd.clone()
        with torch.no_grad():
            z, ld = flow.bijection.forward(xd)
        assert torch.equal(xd, keep)
        z_ref, ld_ref = ge.run_program(prog, x, check_windows=False)
        assert rel(z.cpu().numpy(), z_ref.numpy()) < 1e-5 and rel(ld.cpu().numpy(), ld_ref.numpy()) < 1e-5, n
    x = torch.randn(3, 5, 3, 32, 32).cuda()
    with torch.no_grad():
        z, ld = flow.bijection.forward(x)
        z2, ld2 = flow.bijection.forward(x.reshape(15, 3, 32, 32))
        lp = flow.log_prob(x)
    assert z.shape == x.shape and ld.shape == (3, 5) and lp.shape == (3, 5)
    assert torch.equal(z.reshape(15, 3, 32, 32), z2) and torch.equal(ld.reshape(15), ld2)


@pytest.mark.parametrize("cls_name,event_shape,n_layers", [("MultiscaleRealNVP", (2, 32, 32), 2), ("AffineGlow", (4, 8, 16), 1),
                                                          ("MultiscaleRealNVP", (1, 16, 16), None), ("AffineGlow", (6, 16, 8), 2),
                                                          ("ShiftGlow", (3, 16, 16), None), ("MultiscaleNICE", (2, 8, 8), 1),
                                                          ("MultiscaleRealNVP", (1, 28, 28), None), ("AffineGlow", (3, 14, 30), 1)])
def test_image_programs_other_presets_vs_host(native, cls_name, event_shape, n_layers):
    """The image-program route beyond config 5: the multiscale RealNVP preset (normalised couplings, no 1x1 convolutions),
    non-square images, one-block models, single-channel images, the shift-coupling presets (NICE, ShiftGlow), MNIST-shaped images whose
    conditioner images need a 2-wide ConvModifier kernel (odd padding, classic.py:26-33) -- forward, inverse and log_prob on the HIP path (one launch
    per coupling, asserted) against this package's ATen path on the host."""
    import torchflows_amd as tfa
    from torchflows_amd import image_program
    from torchflows_amd.bijections.finite import multiscale
    torch.manual_seed(7)
    flow = tfa.Flow(getattr(multiscale, cls_name)(event_shape, n_layers=n_layers))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(48, *event_shape))
    flow.eval()
    x = torch.randn(21, *event_shape)
    with torch.no_grad():
        z_h, ld_h = flow.bijection.forward(x)
        lp_h = flow.log_prob(x)
        flow = flow.cuda()
        prog = image_program.get_program(flow.bijection, 0, torch.device("cuda", 0))
        assert prog is not None, "the compiler declined a model it should cover"
        before = native.calls
        z, ld = flow.bijection.forward(x.cuda())
        assert native.calls - before == len(prog.steps) + (1 if prog.flush is not None else 0)
        lp = flow.log_prob(x.cuda())
        xr, ldr = flow.bijection.inverse(z)
    assert rel(z.cpu().numpy(), z_h.numpy()) < 1e-5 and rel(ld.cpu().numpy(), ld_h.numpy()) < 1e-5
    assert rel(lp.cpu().numpy(), lp_h.numpy()) < 1e-5
    assert rel(xr.cpu().numpy(), x.numpy()) < 1e-4 and rel((-ldr).cpu().numpy(), ld.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("cls_name,event_shape,n_layers", [("AffineGlow", (3, 32, 32), None), ("AffineGlow", (3, 16, 16), None),
                                                          ("MultiscaleRealNVP", (1, 28, 28), None), ("ShiftGlow", (3, 16, 16), None),
                                                          ("AffineGlow", (6, 16, 8), 2)])
def test_level_launches_equal_the_coupling_launches(native, monkeypatch, cls_name, event_shape, n_layers):
    """tfk_glow_level (round 4, opt-in: TORCHFLOWS_AMD_GLOW_LEVELS=1): the couplings of one level of the multiscale
    recursion back to back on rows held in the LDS -- same tables, same pending maps, the Linear layer on
    v_mfma_f32_4x4x1 instead of 16x16x4.  Its rows must equal those of the one-launch-per-coupling route BIT FOR BIT
    (every per-element operation is the same fp32 sequence; the MFMA variants add the 16 products in the same order),
    the log-det to one rounding (summed per level before it meets the running value), forward and inverse, affine /
    shift / 1x1-convolution couplings, the 2-wide modifier kernel of 28-pixel images, and batch sizes that leave a partial
    group of four samples."""
    import torchflows_amd as tfa
    from torchflows_amd import image_program
    from torchflows_amd.bijections.finite import multiscale
    torch.manual_seed(11)
    flow = tfa.Flow(getattr(multiscale, cls_name)(event_shape, n_layers=n_layers))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(48, *event_shape))
    flow = flow.eval().cuda()
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("TORCHFLOWS_AMD_GLOW_LEVELS", "1")
    for d in (0, 1):
        prog = image_program.compile_program(flow.bijection, d, dev)
        assert prog is not None and prog.levels is not None
        assert sum(lv.count for lv in prog.levels) == len(prog.steps) and len(prog.levels) < len(prog.steps)
        for n in (1, 5, 64, 131):
            x = torch.randn(n, *event_shape, device=dev) * (1.0 if d == 0 else 0.7)
            before = native.calls
            out_l, ld_l = image_program.run(prog, x, event_shape)
            assert native.calls - before == len(prog.levels) + (1 if prog.flush is not None else 0)
            levels, prog.levels = prog.levels, None
            out_s, ld_s = image_program.run(prog, x, event_shape)
            prog.levels = levels
            assert torch.equal(out_l, out_s), (cls_name, d, n, float((out_l - out_s).abs().max()))
            assert rel(ld_l.cpu().numpy(), ld_s.cpu().numpy()) < 5e-6     # (partial sums of ~1e2 cancel to ~1e-1: a few ulps of those)


@pytest.mark.parametrize("event_shape,n", [((3, 32, 32), 64), ((1, 28, 28), 16), ((3, 16, 16), 33)])
def test_glow_hip_vs_host_config5(native, event_shape, n):
    """Config 5 model (AffineGlow on 32x32x3, 3.2 M parameters), HIP path vs this package's
    ATen path on the host, plus the reference's round-trip property."""
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    torch.manual_seed(0)
    flow = tfa.Flow(AffineGlow(event_shape))
    x = torch.randn(n, *event_shape)
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(64, *event_shape))       # ActNorm init, BatchNorm statistics
    flow.eval()
    with torch.no_grad():
        lp_h = flow.log_prob(x)
        flow = flow.cuda()
        lp_d = flow.log_prob(x.cuda())
        z, ld = flow.bijection.forward(x.cuda())
        xr, ldr = flow.bijection.inverse(z)
    assert lp_d.shape == (n,)
    assert rel(lp_d.cpu().numpy(), lp_h.numpy()) < 2e-5      # MIOpen vs host convolutions inside
    assert torch.allclose(xr.cpu(), x, atol=1e-3) and torch.allclose(ld, -ldr, atol=1e-3)


@pytest.mark.parametrize("c_in,c_out,H,W,N", [(4, 8, 32, 32, 300), (8, 8, 16, 16, 513), (8, 4, 8, 8, 1000),
                                                (4, 4, 2, 6, 7), (8, 8, 10, 4, 33)])
def test_conv_block_kernel_vs_torch(c_in, c_out, H, W, N):
    """tfk_conv3x3_relu_pool_affine = conv3x3(pad 1) -> ReLU -> MaxPool2d(2) -> inference BatchNorm of the
    Glow ConvNet conditioner (classic.py: ConvNetBlock.forward), against the same ops in fp64 on the host."""
    from torchflows_amd import native
    torch.manual_seed(c_in * 100 + c_out + H)
    conv = torch.nn.Conv2d(c_in, c_out, 3, padding=1)
    bn = torch.nn.BatchNorm2d(c_out).eval()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_()
        bn.bias.normal_()
    x = torch.randn(N, c_in, H, W)
    with torch.no_grad():
        ref = bn.double()(torch.nn.functional.max_pool2d(torch.relu(conv.double()(x.double())), 2))
        conv, bn = conv.float().cuda(), bn.float().cuda()
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        before = native.calls
        out = native.conv3x3_relu_pool_affine(x.cuda(), conv.weight, conv.bias, scale, shift)
    assert native.calls == before + 1 and out.shape == ref.shape
    err = float((out.cpu().double() - ref).abs().max() / max(1.0, float(ref.abs().max())))
    assert err < 1e-5, err


@pytest.mark.parametrize("c,h,w,c_t,h_t,w_t,N", [(3, 32, 16, 4, 32, 32, 257), (12, 16, 16, 4, 32, 32, 100),
                                                  (4, 4, 4, 1, 10, 10, 1000), (24, 8, 8, 4, 32, 32, 31),
                                                  (6, 32, 32, 4, 32, 32, 5)])
def test_conv_modifier_kernel_vs_torch(c, h, w, c_t, h_t, w_t, N):
    """tfk_conv1x1_frame = the reference's ConvModifier when its kernel is 1x1 (classic.py:8-42: conv2d with
    padding > kernel - 1), against conv2d in fp64 on the host; also from a view of wider rows, as the
    coupling hands it over."""
    from torchflows_amd import native
    from torchflows_amd.bijections.finite.multiscale.conditioning.classic import ConvModifier
    torch.manual_seed(c + h)
    mod = ConvModifier((c, h, w), c_target=c_t, h_target=h_t, w_target=w_t).eval()
    assert tuple(mod.conv.kernel_size) == (1, 1)
    rows = torch.randn(N, 2 * c * h * w)
    x = rows[:, :c * h * w].view(N, c, h, w)                 # images contiguous, rows twice as wide
    with torch.no_grad():
        ref = mod.conv.double()(x.double())                  # the reference's op: one padded convolution
        mod = mod.float().cuda()
        before = native.calls
        out_view = mod(rows.cuda()[:, :c * h * w].view(N, c, h, w))
        out = mod(x.contiguous().cuda())
    assert native.calls == before + 2 and out.shape == ref.shape == (N, c_t, h_t, w_t)
    assert torch.equal(out, out_view)
    err = float((out.cpu().double() - ref).abs().max() / max(1.0, float(ref.abs().max())))
    assert err < 1e-5, err


@pytest.mark.parametrize("n", [1, 7, 4096, 100003])
def test_bounded_sigmoid_kernel_matches_aten(n):
    """tfk_bounded_sigmoid = ``sigmoid(h) * (hi - lo) + lo`` (transforms.py:107-113) with the same three
    roundings as ATen's three kernels."""
    from torchflows_amd import native
    torch.manual_seed(n)
    h = (torch.randn(n) * 6).cuda()
    ref = torch.sigmoid(h) * 4.0 + (-2.0)
    out = native.bounded_sigmoid(h, -2.0, 2.0)
    assert float((out - ref).abs().max()) <= 2.4e-7          # <= 1 ulp at |value| <= 2
    ref64 = torch.sigmoid(h.double()) * 4.0 - 2.0
    assert float((out.double() - ref64).abs().max()) < 1e-6


@pytest.mark.parametrize("event_shape", [(3, 56, 56), (3, 64, 64), (12289,)])
def test_large_event_elementwise_and_base_density(native, event_shape):
    """events beyond the 64 KiB LDS parameter cache (D > 8064 for ActNorm / ElementwiseAffine, D > 5461 for
    the base density) -- every multiscale block starts with an ActNorm over the whole image -- go through column
    tiles instead of raising.  Against the same layers on the host (ATen), fp32."""
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm, ElementwiseAffine
    torch.manual_seed(4)
    N = 37
    x = torch.randn(N, *event_shape)
    for cls in (ElementwiseAffine, ActNorm):
        layer = cls(event_shape).eval()
        with torch.no_grad():
            layer.value.mul_(0.3)
            z_h, ld_h = layer.forward(x)
            xi_h, ldi_h = layer.inverse(x)
        layer = layer.cuda()
        before = native.calls
        with torch.no_grad():
            z_d, ld_d = layer.forward(x.cuda())
            xi_d, ldi_d = layer.inverse(x.cuda())
        assert native.calls - before == 2
        assert rel(z_d.cpu().numpy(), z_h.numpy()) < 1e-5 and rel(xi_d.cpu().numpy(), xi_h.numpy()) < 1e-5
        assert rel(ld_d.cpu().numpy(), ld_h.numpy()) < 1e-5 and rel(ldi_d.cpu().numpy(), ldi_h.numpy()) < 1e-5
    flow = tfa.Flow(ElementwiseAffine(event_shape)).eval()
    with torch.no_grad():
        flow.bijection.value.mul_(0.3)
        lp_h = flow.log_prob(x)
        lp_d = flow.cuda().log_prob(x.cuda())
        xs, lps = flow.sample((5,), return_log_prob=True)
    assert rel(lp_d.cpu().numpy(), lp_h.numpy()) < 1e-5
    assert xs.shape == (5, *event_shape) and torch.isfinite(lps).all()
