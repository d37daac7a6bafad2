"""The ConvNet conditioner with gradients on libtfk (torchflows_amd/convnet_train.py, csrc/tfk_convtrain.hip) against
the ATen composite path -- the reference's own op chain (multiscale/conditioning/classic.py:45-122) -- evaluated in
float64 on the CPU; the bar for every quantity is the larger of an absolute floor and three times the distance of the
fp32 ATen/MIOpen route from that float64 value."""
import copy

import numpy as np
import pytest
import torch

from conftest import set_debug

pytestmark = pytest.mark.gpu


def normwise(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def native():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from torchflows_amd import native as nat
    nat.lib()
    return nat


def _net(shape, n_out, seed=0):
    from torchflows_amd.bijections.finite.multiscale.conditioning.classic import ConvNet
    torch.manual_seed(seed)
    net = ConvNet(shape, n_out)
    with torch.no_grad():               # non-trivial BatchNorm state
        for blk in list(net.blocks)[1:4]:
            blk.bn.weight.uniform_(0.5, 1.5)
            blk.bn.bias.uniform_(-0.3, 0.3)
            blk.bn.running_mean.uniform_(-0.2, 0.4)
            blk.bn.running_var.uniform_(0.5, 2.0)
    return net


def _evaluate(net, x, g, training):
    net.train(training)
    x = x.clone().requires_grad_(True)
    out = net(x)
    grads = torch.autograd.grad(out, [x] + list(net.parameters()), g)
    state = {k: v.detach().cpu().double().numpy() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
    return (out.detach().cpu().double().numpy(), [t.detach().cpu().double().numpy() for t in grads], state)


_SHAPES = [((1, 14, 28), 37), ((3, 16, 32), 19), ((6, 16, 16), 64), ((12, 8, 8), 5), ((2, 32, 32), 3), ((2, 7, 14), 21),
           ((4, 7, 7), 33), ((24, 3, 3), 9),
           ((3, 16, 32), 640)]      # (several samples per workgroup, two-level sums)
# batch statistics (what Flow.fit runs) on every input frame; the running-statistics variant of the same launches on three
_CASES = [(s, n, True) for s, n in _SHAPES] + [(s, n, False) for s, n in (_SHAPES[0], _SHAPES[5], _SHAPES[7])]


@pytest.mark.parametrize("shape,N,training", _CASES,
                         ids=[f"{'x'.join(map(str, s))}-N{n}-{'batch' if t else 'running'}" for s, n, t in _CASES])
def test_convnet_forward_and_gradients(native, monkeypatch, shape, N, training):
    n_out = 2 * int(np.prod(shape))
    ref = _net(shape, n_out)
    torch.manual_seed(1)
    x = torch.randn(N, *shape) * 1.5 + 0.2
    g = torch.randn(N, n_out)
    want = _evaluate(copy.deepcopy(ref).double(), x.double(), g.double(), training)          # float64, CPU, ATen
    set_debug(monkeypatch, convnet_train="0")
    aten = _evaluate(copy.deepcopy(ref).cuda(), x.cuda(), g.cuda(), training)                 # fp32 ATen / MIOpen
    set_debug(monkeypatch, convnet_train=None)
    before = native.calls
    got = _evaluate(copy.deepcopy(ref).cuda(), x.cuda(), g.cuda(), training)                  # fp32 libtfk
    assert native.calls - before == 7 + 7, "one forward launch per block / modifier, one reverse-mode launch each + linear"
    names = ["x"] + [k for k, _ in ref.named_parameters()]
    assert normwise(got[0], want[0]) < max(2e-6, 3 * normwise(aten[0], want[0]))
    worst = 0.0
    for name, a, b, c in zip(names, got[1], want[1], aten[1]):
        assert a.shape == b.shape, name
        e, bar = normwise(a, b), max(2e-5, 3 * normwise(c, b))
        worst = max(worst, e / bar)
        assert e < bar, (name, e, bar)
    for k in want[2]:
        if "num_batches" in k:
            assert got[2][k] == want[2][k], k
        else:
            assert normwise(got[2][k], want[2][k]) < 1e-5, k
    print(f"convnet {shape} N={N} training={training}: worst gradient error / bar = {worst:.3f}")


def test_convnet_gradients_are_deterministic(native):
    net = _net((3, 16, 32), 96).cuda().train()
    torch.manual_seed(3)
    x = torch.randn(300, 3, 16, 32, device="cuda")
    g = torch.randn(300, 96, device="cuda")
    runs = []
    for _ in range(2):
        m = copy.deepcopy(net)
        xr = x.clone().requires_grad_(True)
        runs.append(torch.autograd.grad(m(xr), [xr] + list(m.parameters()), g))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_coupling_training_step_counts_the_batch_once(native, monkeypatch):
    """A training-mode coupling through the reverse-mode chain (autograd.ChainFunction re-evaluates the conditioner in its
    backward): gradients equal the plain autograd graph's, and the BatchNorm running statistics move ONCE per step."""
    from torchflows_amd.bijections.finite.multiscale.base import NormalizedCheckerboardCoupling
    from torchflows_amd.bijections.finite.autoregressive.transformers.linear.affine import Affine
    torch.manual_seed(0)
    layer = NormalizedCheckerboardCoupling((1, 28, 28), transformer_class=Affine).cuda()
    x = torch.randn(50, 1, 28, 28, device="cuda")
    with torch.no_grad():
        layer.train()
        layer(x)                      # ActNorm takes its statistics
    counted = {k: int(v) for k, v in layer.state_dict().items() if "num_batches" in k}
    results = []
    for route in ("hip", "aten"):
        m = copy.deepcopy(layer).train()
        monkeypatch.setenv("TORCHFLOWS_AMD_TRAIN", "1" if route == "hip" else "0")
        set_debug(monkeypatch, convnet_train=None if route == "hip" else "0")
        before = native.calls
        z, ld = m(x)
        loss = (z ** 2).sum() * 0.5 - ld.sum()
        grads = torch.autograd.grad(loss, [p for p in m.parameters() if p.requires_grad and p.numel()])
        results.append((grads, {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k},
                        native.calls - before))
    (g_hip, s_hip, n_hip), (g_aten, s_aten, _) = results
    for a, b in zip(g_hip, g_aten):
        assert normwise(a.cpu().numpy(), b.cpu().numpy()) < 2e-4
    for k in s_aten:
        if "num_batches" in k:
            assert int(s_hip[k]) == int(s_aten[k]) == counted[k] + 1, k
        else:
            assert normwise(s_hip[k].cpu().numpy(), s_aten[k].cpu().numpy()) < 1e-5, k
    assert n_hip <= 30, f"{n_hip} libtfk launches for one coupling's training step"


def test_image_flow_fit_replays_a_captured_step(native, monkeypatch):
    """Flow.fit on an image flow: after two eager steps the training step (libtfk launches + elementwise / index ATen ops,
    no MIOpen or GEMM-library call) is captured ONCE into a hipGraph and replayed; the fit moves the loss like the eager
    loop does."""
    import torchflows_amd as tfa
    from torchflows_amd.architectures import MultiscaleRealNVP
    torch.manual_seed(0)
    x = torch.randn(256, 1, 28, 28)
    x = (x - x.mean()) / x.std()
    base = tfa.Flow(MultiscaleRealNVP((1, 28, 28)))
    assert base.cuda()._graph_safe()
    after = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TORCHFLOWS_AMD_GRAPH", mode)
        flow = copy.deepcopy(base).cuda()
        flow.fit(x, n_epochs=14, batch_size=256, lr=0.01)
        stats = flow._fit_stats
        if mode == "1":
            assert stats["graph_captures"] == 1 and stats["graph_replays"] == 12 and stats["eager_steps"] == 2, stats
        else:
            assert stats["graph_replays"] == 0 and stats["eager_steps"] == 14, stats
        assert all(bool(torch.isfinite(p).all()) for p in flow.parameters())
        with torch.no_grad():
            after[mode] = float(-flow.log_prob(x.cuda()).mean())
    with torch.no_grad():
        start = float(-copy.deepcopy(base).cuda().eval().log_prob(x.cuda()).mean())
    assert after["1"] < start and after["0"] < start, (start, after)
    assert abs(after["1"] - after["0"]) < 0.05 * abs(after["0"]), after


def test_bounded_output_one_launch_each_way(native):
    """lo + (hi - lo) sigmoid(h) of the bounded conditioners (transforms.py:107-113) with its gradient: one libtfk launch
    forward, one backward, equal to the three-kernel ATen expression; second derivatives still available."""
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import _BoundedSigmoid
    torch.manual_seed(0)
    h = (torch.randn(513, 97, device="cuda") * 4).requires_grad_(True)
    g = torch.randn(513, 97, device="cuda")
    before = native.calls
    out = _BoundedSigmoid.apply(h, -2.0, 2.0)
    (got,) = torch.autograd.grad(out, h, g)
    assert native.calls - before == 2
    h2 = h.detach().clone().requires_grad_(True)
    want_out = torch.sigmoid(h2) * 4.0 - 2.0
    (want,) = torch.autograd.grad(want_out, h2, g)
    assert torch.equal(out, want_out)
    assert float((got - want).abs().max()) < 2e-6 * float(want.abs().max())
    out = _BoundedSigmoid.apply(h, -2.0, 2.0)
    (first,) = torch.autograd.grad(out.sum(), h, create_graph=True)
    (second,) = torch.autograd.grad(first.sum(), h)
    s = torch.sigmoid(h.detach())
    assert float((second - 4.0 * s * (1 - s) * (1 - 2 * s)).abs().max()) < 1e-4


@pytest.mark.parametrize("training", [False, True], ids=["eval", "train"])
def test_image_flow_inverse_direction_gradients(native, monkeypatch, training):
    """The sampling direction with gradients (variational_fit on an image flow): bijection.inverse through the reverse-mode
    chain + the libtfk conditioner against the plain ATen autograd graph of the same modules."""
    from torchflows_amd.bijections.finite.multiscale.architectures import AffineGlow
    torch.manual_seed(0)
    bij = AffineGlow((3, 8, 8), n_layers=2).cuda()
    z = torch.randn(24, 3, 8, 8, device="cuda")
    with torch.no_grad():
        bij.train()
        bij.forward(torch.randn(24, 3, 8, 8, device="cuda"))       # ActNorm statistics
    bij.train(training)
    got = {}
    for route in ("hip", "aten"):
        m = copy.deepcopy(bij)
        monkeypatch.setenv("TORCHFLOWS_AMD_TRAIN", "1" if route == "hip" else "0")
        set_debug(monkeypatch, convnet_train=None if route == "hip" else "0")
        x, ld = m.inverse(z)
        loss = (x ** 2).sum() * 0.5 - ld.sum()
        params = [p for p in m.parameters() if p.requires_grad and p.numel()]
        got[route] = (x.detach(), ld.detach(), torch.autograd.grad(loss, params, allow_unused=True))
    assert normwise(got["hip"][0].cpu().numpy(), got["aten"][0].cpu().numpy()) < 1e-5
    assert normwise(got["hip"][1].cpu().numpy(), got["aten"][1].cpu().numpy()) < 1e-5
    for a, b in zip(got["hip"][2], got["aten"][2]):
        if a is None or b is None:
            assert a is None and b is None
            continue
        assert normwise(a.cpu().numpy(), b.cpu().numpy()) < 5e-4


def test_config5_model_training_step_matches_the_library_route(native, monkeypatch):
    """One maximum-likelihood step of AffineGlow((3, 32, 32)) -- config 5's model, training mode -- loss and every gradient:
    ConvNet conditioners on csrc/tfk_convtrain.hip against the same chain with the conditioners on ATen / MIOpen."""
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.multiscale.architectures import AffineGlow
    torch.manual_seed(0)
    flow = tfa.Flow(AffineGlow((3, 32, 32))).cuda()
    x = torch.randn(48, 3, 32, 32, device="cuda")
    flow.train()
    with torch.no_grad():
        flow.log_prob(x)                  # ActNorm statistics
    out = {}
    for route in ("libtfk", "library"):
        m = copy.deepcopy(flow).train()
        set_debug(monkeypatch, convnet_train=None if route == "libtfk" else "0")
        before = native.calls
        loss = -m.log_prob(x).mean() / 3072
        params = [(k, p) for k, p in m.named_parameters() if p.requires_grad and p.numel()]
        grads = torch.autograd.grad(loss, [p for _, p in params], allow_unused=True)
        out[route] = (float(loss), {k: g for (k, _), g in zip(params, grads)}, native.calls - before)
    assert abs(out["libtfk"][0] - out["library"][0]) < 1e-5 * abs(out["library"][0])
    worst = ("", 0.0)
    for k, g in out["library"][1].items():
        h = out["libtfk"][1][k]
        assert (g is None) == (h is None), k
        if g is None:
            continue
        e = normwise(h.cpu().numpy(), g.cpu().numpy())
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < 2e-3, worst       # (fp32 noise of two different summation orders through 19 normalised couplings)
    print(f"config-5 training step: loss {out['libtfk'][0]:.6f}, worst gradient distance between routes {worst[1]:.2e} "
          f"({worst[0]}); libtfk launches {out['libtfk'][2]} vs {out['library'][2]}")


def test_one_call_and_launch_by_launch_routes_agree_bitwise(native, monkeypatch):
    """tfk_convnet_train_forward / _backward issue the same launches as the per-launch entry points called from Python."""
    net = _net((3, 16, 32), 96).cuda().train()
    torch.manual_seed(5)
    x = torch.randn(70, 3, 16, 32, device="cuda")
    g = torch.randn(70, 96, device="cuda")
    runs = []
    for mode in (None, "each"):
        set_debug(monkeypatch, convnet_calls=mode)
        m = copy.deepcopy(net)
        xr = x.clone().requires_grad_(True)
        out = m(xr)
        runs.append((out.detach(), torch.autograd.grad(out, [xr] + list(m.parameters()), g),
                     {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
    assert torch.equal(runs[0][0], runs[1][0])
    for i, (a, b) in enumerate(zip(runs[0][1], runs[1][1])):
        if i == 16:       # the second modifier's bias: its frame share is a dot product, summed by libtfk here, rocBLAS there
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        else:
            assert torch.equal(a, b), i
    for k in runs[0][2]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k


@pytest.mark.parametrize("preset,shape", [("MultiscaleNICE", (1, 28, 28)), ("ShiftGlow", (3, 16, 16)), ("AffineGlow", (3, 16, 16))])
def test_every_multiscale_preset_fits_on_the_captured_step(native, monkeypatch, preset, shape):
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.multiscale import architectures
    monkeypatch.setenv("TORCHFLOWS_AMD_GRAPH", "1")
    torch.manual_seed(0)
    flow = tfa.Flow(getattr(architectures, preset)(shape)).cuda()
    assert flow._graph_safe()
    x = torch.randn(96, *shape)
    flow.fit(x, n_epochs=8, batch_size=96, lr=0.01)
    stats = flow._fit_stats
    assert stats["graph_captures"] == 1 and stats["graph_replays"] == 6, stats
    assert all(bool(torch.isfinite(p).all()) for p in flow.parameters())
    with torch.no_grad():
        assert bool(torch.isfinite(flow.log_prob(x.cuda())).all())


def test_image_flow_validation_pass_is_replayed_too(native, monkeypatch):
    """Flow.fit with a validation set on an image flow: the validation pass is captured as well; its losses are the eager
    pass's (same launches, same live parameters), so the two fits are the same fit."""
    import torchflows_amd as tfa
    from torchflows_amd.architectures import MultiscaleRealNVP
    monkeypatch.setenv("TORCHFLOWS_AMD_GRAPH", "1")
    torch.manual_seed(0)
    x = torch.randn(200, 1, 28, 28)
    xv = torch.randn(64, 1, 28, 28)
    base = tfa.Flow(MultiscaleRealNVP((1, 28, 28)))
    out = {}
    for mode in ("1", "0"):
        set_debug(monkeypatch, val_graph=None if mode == "1" else "0")
        flow = copy.deepcopy(base).cuda()
        torch.manual_seed(1)              # (the epoch's row order)
        flow.fit(x, x_val=xv, n_epochs=10, batch_size=200, lr=0.01, early_stopping=True)
        out[mode] = (dict(flow._fit_stats), [p.detach().clone() for p in flow.parameters()]
                     + [b.detach().clone() for b in flow.buffers()])           # (BatchNorm's running statistics too)
    assert out["1"][0].get("val_graph_captures") == 1 and out["1"][0]["val_graph_replays"] >= 6, out["1"][0]
    assert "val_graph_replays" not in out["0"][0]
    assert out["1"][0]["val_loss"] == out["0"][0]["val_loss"]           # (every batch sum has a fixed order: same bits)
    for a, b in zip(out["1"][1], out["0"][1]):
        assert torch.equal(a, b)


def test_image_flow_fit_with_a_ragged_last_batch(native, monkeypatch):
    """Several steps per epoch, the last batch smaller: full batches replay the captured step, the ragged one runs eagerly."""
    import torchflows_amd as tfa
    from torchflows_amd.architectures import MultiscaleRealNVP
    monkeypatch.setenv("TORCHFLOWS_AMD_GRAPH", "1")
    torch.manual_seed(0)
    flow = tfa.Flow(MultiscaleRealNVP((1, 28, 28))).cuda()
    x = torch.randn(250, 1, 28, 28)
    flow.fit(x, n_epochs=4, batch_size=100, lr=0.01)
    stats = flow._fit_stats
    assert stats["graph_captures"] == 1 and stats["graph_replays"] == 6 and stats["eager_steps"] == 2 + 4, stats
    assert all(bool(torch.isfinite(p).all()) for p in flow.parameters())


def test_linear_input_gradient_both_launch_shapes(native):
    """g16 = g W16 of the folded Linear layer: the scalar launch (few rows) and the 16-byte-load launch (>= 16 rows per CU,
    M % 4 == 0) against float64."""
    torch.manual_seed(0)
    for N, M in ((300, 392), (5000, 392), (5000, 394), (4100, 3072)):
        g = torch.randn(N, M, device="cuda")
        W16 = torch.randn(M, 16, device="cuda")
        got = native.convnet_train_linear_bwd_input(g, W16)
        want = (g.double() @ W16.double())
        assert float((got.double() - want).abs().max()) < 2e-5 * float(want.abs().max()), (N, M)
