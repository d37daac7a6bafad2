"""CPU-side tests (run with -m "not gpu"): the host mirror of the plugin surface on its
ATen composite path against the reference's golden outputs, the reference's own property
tests re-pointed at this package, the C-ABI library's exports, and the 2-rank gloo path."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, free_port, load_golden, state_dict_of, set_debug

import torchflows_amd as tfa
from torchflows_amd import native
from torchflows_amd.bijections.base import BijectiveComposition, invert
from torchflows_amd.bijections.finite.autoregressive.conditioning.coupling_masks import (
    GraphicalCoupling, HalfSplit)
from torchflows_amd.bijections.finite.autoregressive.layers import (
    ActNorm, AffineCoupling, ElementwiseAffine, RQSCoupling, ShiftCoupling)
from torchflows_amd.bijections.finite.autoregressive.transformers.linear.affine import Affine
from torchflows_amd.bijections.finite.autoregressive.transformers.spline.rational_quadratic import (
    RationalQuadratic)
from torchflows_amd.bijections.finite.matrix.permutation import (
    RandomPermutationMatrix, ReversePermutationMatrix)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


# ------------------------------------------------------------------ C-ABI library
def test_library_exports_every_declared_symbol():
    """include/tfk.h <-> libtfk.so: every declared entry point is exported (no compute here)."""
    header = open(os.path.join(ROOT, "include", "tfk.h")).read()
    declared = set(re.findall(r"\b(tfk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)
    L = ctypes.CDLL(native.LIB_PATH)
    for s in native.SYMBOLS:
        assert hasattr(L, s), s
    assert native.lib().tfk_abi_version() == native.ABI_VERSION


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "torchflows_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_host_tensors_are_rejected_by_the_binding():
    x = torch.zeros(4, 64)
    with pytest.raises(native.NativeError):
        native.affine_coupling(x, torch.zeros(4, 32, 2), x.clone(), torch.zeros(4), None, 32)
    assert not native.eligible(x)                    # host tensors take the ATen path


# ------------------------------------------------------------------ integer rules
def test_masks_and_permutations_bit_exact_with_reference():
    fx = load_golden("masks.npz")
    for tag in fx["shapes"]:
        es = tuple(int(t) for t in str(tag).split("x"))
        c = HalfSplit(es)
        assert np.array_equal(c.source_mask.numpy().astype(np.uint8), fx[f"src_{tag}"])
        assert np.array_equal(c.target_mask.numpy().astype(np.uint8), fx[f"tgt_{tag}"])
        assert c.source_event_size == int(fx[f"S_{tag}"]) and c.target_event_size == int(fx[f"T_{tag}"])
        assert c.constant_shape == (c.source_event_size,) and c.target_shape == (c.target_event_size,)
        assert c.target_is_tail and (c.source_is_head or c.source_event_size == 0)
        assert np.array_equal(c.target_index.numpy(), np.nonzero(fx[f"tgt_{tag}"].reshape(-1))[0])
        assert np.array_equal(c.source_index.numpy(), np.nonzero(fx[f"src_{tag}"].reshape(-1))[0])
        p = ReversePermutationMatrix(es)
        assert np.array_equal(p.forward_permutation.numpy(), fx[f"pfwd_{tag}"])
        assert np.array_equal(p.inverse_permutation.numpy(), fx[f"pinv_{tag}"])
        assert p._is_reversal
    g = GraphicalCoupling((5,), [(0, 3), (1, 3), (0, 4)])
    assert g.source_index.tolist() == [0, 1] and g.target_index.tolist() == [3, 4]
    assert g.ignored_event_size == 1 and g.target_is_tail and g.source_is_head
    g2 = GraphicalCoupling((5,), [(4, 0), (2, 1)])
    assert not g2.target_is_tail and not g2.source_is_head
    with pytest.raises(ValueError):
        GraphicalCoupling((2, 2), [(0, 1)])


# ------------------------------------------------------------------ golden: transformers
@pytest.mark.parametrize("T", [2, 32, 128])
def test_affine_transformer_golden(T):
    fx = load_golden("affine.npz")
    tr = Affine((T,))
    x, h = torch.from_numpy(fx[f"T{T}_x"]), torch.from_numpy(fx[f"T{T}_h"])
    z, ld = tr.forward(x, h)
    xi, ldi = tr.inverse(x, h)
    assert torch.equal(z, torch.from_numpy(fx[f"T{T}_z"]))          # same ATen ops: bit-exact
    assert torch.equal(xi, torch.from_numpy(fx[f"T{T}_xinv"]))
    assert rel(ld.numpy(), fx[f"T{T}_ld"]) < 1e-6 and rel(ldi.numpy(), fx[f"T{T}_ldinv"]) < 1e-6


def test_rqs_transformer_golden():
    fx = load_golden("rqs.npz")
    for tag in fx["cases"]:
        tag = str(tag)
        K, B = int(tag.split("K")[1]), float(tag.split("_")[0][1:])
        x, h = torch.from_numpy(fx[f"{tag}_x"]), torch.from_numpy(fx[f"{tag}_h"])
        tr = RationalQuadratic((x.shape[1],), boundary=B, n_bins=K)
        z, ld = tr.forward(x, h)
        xi, ldi = tr.inverse(x, h)
        assert rel(z.numpy(), fx[f"{tag}_z"]) < 1e-6 and rel(ld.numpy(), fx[f"{tag}_ld"]) < 2e-6
        assert rel(xi.numpy(), fx[f"{tag}_xinv"]) < 1e-6 and rel(ldi.numpy(), fx[f"{tag}_ldinv"]) < 2e-6


# ------------------------------------------------------------------ golden: whole flows
FLOWS = [
    ("flow_realnvp3.npz", tfa.RealNVP, {}, None),
    ("flow_realnvp64.npz", tfa.RealNVP, dict(n_layers=8), None),
    ("flow_nsf64.npz", tfa.CouplingRQNSF, dict(n_layers=8), None),
    ("flow_realnvp256.npz", tfa.RealNVP, dict(n_layers=8), None),
    ("flow_nice7.npz", tfa.NICE, {}, None),
    ("flow_realnvp_7x11.npz", tfa.RealNVP, {}, None),
    ("flow_realnvp5_ctx3.npz", tfa.RealNVP, {}, (3,)),
    ("flow_nsf6_ctx2.npz", tfa.CouplingRQNSF, {}, (2,)),
    ("flow_nsf_3x5x2.npz", tfa.CouplingRQNSF, {}, None),
    ("flow_lrs16.npz", tfa.CouplingLRS, dict(n_layers=3), None),
    ("flow_maf6.npz", tfa.MAF, dict(n_layers=2), None),
    ("flow_iaf6.npz", tfa.IAF, dict(n_layers=2), None),
    ("flow_marqnsf5.npz", tfa.MaskedAutoregressiveRQNSF, dict(n_layers=2), None),
    ("flow_iarqnsf5.npz", tfa.InverseAutoregressiveRQNSF, dict(n_layers=2), None),
    ("flow_malrs5.npz", tfa.MaskedAutoregressiveLRS, dict(n_layers=2), None),
]


@pytest.mark.parametrize("name,ctor,kw,ctx_shape", FLOWS)
def test_flow_golden_aten_path(name, ctor, kw, ctx_shape):
    """Same seed -> same initial weights and state-dict keys as the reference; same
    weights -> same log_prob / inverse (ATen composite path on host tensors)."""
    fx = load_golden(name)
    es = tuple(int(v) for v in fx["event_shape"])
    kw = dict(kw)
    if ctx_shape:
        kw["context_shape"] = ctx_shape
    torch.manual_seed(0)
    flow = tfa.Flow(ctor(es if len(es) > 1 else es[0], **kw))
    assert [type(l).__name__ for l in flow.bijection.layers] == [str(t) for t in fx["layer_types"]]
    fresh = state_dict_of(fx, "fresh")
    sd = flow.state_dict()
    assert list(sd.keys()) == list(fresh.keys())
    for k, v in fresh.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        if k != "device_buffer" and v.ndim > 0:
            assert np.array_equal(sd[k].numpy(), v), f"same-seed init differs at {k}"
    ctx = torch.from_numpy(fx["context"]) if ctx_shape else None
    for variant in ("fresh", "init"):
        flow.load_state_dict({k: torch.from_numpy(v) for k, v in state_dict_of(fx, variant).items()})
        flow.eval()
        with torch.no_grad():
            z, lp = flow.forward_with_log_prob(torch.from_numpy(fx["x"]), context=ctx)
            xr, ldr = flow.bijection.inverse(torch.from_numpy(fx["z_in"]), context=ctx)
        g = lambda k: fx[f"{variant}/{k}"]
        if ctor is tfa.CouplingRQNSF:
            # per layer this path is bit-identical to the reference up to ATen's vector-body /
            # scalar-tail split (the reference evaluates the gathered in-box subset, this
            # package the whole tensor): 1-ulp differences that later layers amplify, i.e.
            # the reference's own fp32 noise.  Bound: norm-wise 2e-5, log-dets 4e-5.
            nw = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
            assert rel(lp.numpy(), g("log_prob")) < 4e-5
            assert nw(z.numpy(), g("z")) < 2e-5 and nw(xr.numpy(), g("x_inv")) < 2e-5
            assert rel(ldr.numpy(), g("log_det_inv")) < 4e-5
        else:
            assert rel(lp.numpy(), g("log_prob")) < 5e-6
            assert rel(z.numpy(), g("z")) < 5e-6
            assert rel(xr.numpy(), g("x_inv")) < 5e-6
            assert rel(ldr.numpy(), g("log_det_inv")) < 5e-6


def test_actnorm_data_dependent_init_golden():
    fx = load_golden("layers.npz")
    for tag in ("n100", "n1"):
        a = ActNorm((7,))
        assert a.training and a.first_training_batch_pass and not a.value.requires_grad
        with torch.no_grad():
            z, ld = a.forward(torch.from_numpy(fx[f"actnorm_{tag}_x"]))
        assert not a.first_training_batch_pass
        assert rel(a.value.numpy(), fx[f"actnorm_{tag}_value"]) < 1e-6
        assert rel(z.numpy(), fx[f"actnorm_{tag}_z"]) < 1e-5
        assert rel(ld.numpy(), fx[f"actnorm_{tag}_ld"]) < 1e-6


def test_single_layers_with_batch_and_event_rank_golden():
    fx = load_golden("layers.npz")
    for cls, tag in ((ElementwiseAffine, "ea"), (AffineCoupling, "ac"), (RQSCoupling, "rc")):
        layer = cls((3, 5, 2)).eval()
        pre = f"{tag}_sd/"
        layer.load_state_dict({k[len(pre):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(pre)})
        x = torch.from_numpy(fx[f"{tag}_x"])
        with torch.no_grad():
            z, ld = layer.forward(x)
            xi, ldi = layer.inverse(x)
        assert ld.shape == (5, 2, 3) and z.shape == x.shape
        assert rel(z.numpy(), fx[f"{tag}_z"]) < 2e-6 and rel(ld.numpy(), fx[f"{tag}_ld"]) < 5e-6
        assert rel(xi.numpy(), fx[f"{tag}_xinv"]) < 2e-6 and rel(ldi.numpy(), fx[f"{tag}_ldinv"]) < 5e-6


# ------------------------------------------------------------------ the reference's property tests
BATCH_SHAPES = [(1,), (2,), (5,), (5, 2, 3)]
EVENT_SHAPES = [(2,), (3,), (3, 5, 2)]
CONTEXT_SHAPES = [None, (2,), (3,), (3, 5, 2)]


@pytest.mark.parametrize("ctor", [tfa.RealNVP, tfa.NICE, tfa.CouplingRQNSF])
@pytest.mark.parametrize("batch_shape", BATCH_SHAPES)
@pytest.mark.parametrize("event_shape", EVENT_SHAPES)
@pytest.mark.parametrize("context_shape", CONTEXT_SHAPES)
def test_coupling_architectures_reconstruct(ctor, batch_shape, event_shape, context_shape):
    """reference test/test_reconstruction_bijections.py:64-93,140-143 (train mode: ActNorm
    initialises on the first forward); tolerances of test/constants.py:11-14."""
    torch.manual_seed(0)
    b = ctor(event_shape, context_shape=context_shape)
    x = torch.randn(*batch_shape, *event_shape)
    ctx = None if context_shape is None else torch.randn(*batch_shape, *context_shape)
    z, ld_f = b.forward(x, context=ctx)
    xr, ld_i = b.inverse(z, context=ctx)
    assert z.shape == x.shape and ld_f.shape == batch_shape and ld_i.shape == batch_shape
    assert torch.isfinite(z).all() and torch.isfinite(ld_f).all()
    assert torch.allclose(x, xr, atol=1e-2)
    assert torch.allclose(ld_f, -ld_i, atol=1e-2)


@pytest.mark.parametrize("cls", [AffineCoupling, RQSCoupling, ShiftCoupling, ElementwiseAffine])
def test_zero_parameters_give_identity(cls):
    """reference test/test_identity_bijections.py:56-68"""
    torch.manual_seed(0)
    layer = cls((5,))
    with torch.no_grad():
        for p in layer.parameters():
            p.zero_()
    x = torch.randn(20, 5)
    z, ld = layer.forward(x)
    xr, ldi = layer.inverse(x)
    assert torch.allclose(z, x, atol=1e-2) and torch.allclose(xr, x, atol=1e-2)
    assert torch.allclose(ld, torch.zeros(20), atol=1e-2) and torch.allclose(ldi, torch.zeros(20), atol=1e-2)


@pytest.mark.parametrize("ctor", [tfa.RealNVP, tfa.CouplingRQNSF])
def test_log_prob_is_differentiable(ctor):
    """reference test/test_autograd_bijections.py:41-54 -- autograd runs on the ATen path."""
    torch.manual_seed(0)
    flow = tfa.Flow(ctor((4,)))
    x = torch.randn(10, 4, requires_grad=True)
    lp = flow.log_prob(x)
    assert lp.shape == (10,) and torch.isfinite(lp).all()
    lp.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    assert any(p.grad is not None for p in flow.parameters())


def test_sample_and_log_prob_conventions():
    """Flow.sample(return_log_prob=True) = log p(z) + log|dx/dz| (reference flows.py:710-712)."""
    torch.manual_seed(0)
    flow = tfa.Flow(tfa.RealNVP(3)).eval()
    with torch.no_grad():
        x = flow.sample(1000)
        assert x.shape == (1000, 3)
        x2, slp = flow.sample((7, 2), return_log_prob=True, no_grad=True)
        assert x2.shape == (7, 2, 3) and slp.shape == (7, 2)
        z2, ld_f = flow.bijection.forward(x2)
        # log p(z) + log|dx/dz| with log|dx/dz| = -log|dz/dx|
        assert torch.allclose(slp, flow.base_log_prob(z2) - ld_f, atol=1e-3)
    cflow = tfa.Flow(tfa.RealNVP(3, context_shape=(2,))).eval()
    with torch.no_grad():
        xs = cflow.sample(5, context=torch.randn(5, 2))
        assert xs.shape == (5, 3)
        xs = cflow.sample(4, context=torch.randn(6, 2))
        assert xs.shape == (4, 6, 3)
        with pytest.raises(AssertionError):
            cflow.log_prob(torch.randn(5, 3), context=torch.randn(4, 2))
    with pytest.raises(ValueError):
        tfa.Flow(tfa.RealNVP(3)).log_prob(torch.randn(5, 3), context=torch.randn(5, 2))


def test_edge_list_drops_permutations_and_1d_fallback():
    """SURVEY quirks Q7 and architectures.py:84-86."""
    b = tfa.RealNVP(5, edge_list=[(0, 1)])
    names = [type(l).__name__ for l in b.layers]
    assert "ReversePermutationMatrix" not in names and len(names) == 3 * 2 + 3 - 2
    one = tfa.RealNVP(1)
    assert all(type(l).__name__ in ("ElementwiseAffine", "ActNorm", "ReversePermutationMatrix")
               for l in one.layers)
    with pytest.raises(ValueError):
        AffineCoupling((1,))


def test_invert_and_composition_api():
    torch.manual_seed(0)
    b = tfa.RealNVP(4).eval()
    x = torch.randn(6, 4)
    with torch.no_grad():
        z, ld = b.forward(x)
        invert(b)
        x2, ld2 = b.forward(z)
        b.invert()
        z2, _ = b.forward(x)
    assert torch.allclose(x2, x, atol=1e-4) and torch.allclose(ld2, -ld, atol=1e-4)
    assert torch.equal(z2, z)
    assert isinstance(b, BijectiveComposition) and float(b.regularization()) >= 0
    with torch.no_grad():
        zs, lds = b.batch_forward(x, batch_size=4)
    assert torch.allclose(zs, z, atol=1e-6) and lds.shape == (6,)
    p = RandomPermutationMatrix((4,))
    y, l0 = p.forward(x)
    xb, _ = p.inverse(y)
    assert torch.equal(xb, x) and torch.all(l0 == 0)


# ------------------------------------------------------------------ 2 ranks over gloo
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_log_likelihood_two_ranks_gloo(world):
    """N > 1 path on CPU: each rank evaluates its shard, one all-reduce of the fp64 sum.  world = 8 is the rank count
    of the driver's scaling run (1001 rows: shards of 126 / 125 rows): every rank must hold the 1-process total."""
    script = os.path.join(ROOT, "tests", "dist_worker.py")
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1" if world > 2 else "2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", port,
                          script], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIST_OK" in out.stdout


def test_sharded_fit_two_ranks_gloo():
    """Data-parallel training on CPU: per-rank shards, one all-reduce of the flat gradient per step,
    ActNorm statistics of the global first batch; must reproduce the one-process run."""
    script = os.path.join(ROOT, "tests", "dist_fit_worker.py")
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", port,
                          script], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIST_FIT_OK" in out.stdout


# ------------------------------------------------------------------ sibling layers
def _sibling(name):
    from torchflows_amd.bijections.finite.autoregressive import layers as L
    from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import ResidualFeedForward
    if name == "AffineCoupling_ResidualFeedForward":
        return L.AffineCoupling((6,), conditioner_transform_class=ResidualFeedForward)
    if name == "RQSCoupling_ResidualFeedForward":
        return L.RQSCoupling((8,), conditioner_transform_class=ResidualFeedForward,
                             conditioner_kwargs=dict(n_layers=4, block_size=3))
    return getattr(L, name)((6,))


@pytest.mark.parametrize("name", ["ElementwiseScale", "ElementwiseRQSpline", "LinearAffineCoupling",
                                  "LinearRQSCoupling", "LinearShiftCoupling",
                                  "AffineCoupling_ResidualFeedForward", "RQSCoupling_ResidualFeedForward"])
def test_sibling_layers_match_reference(name):
    """Same constructor + the reference's state dict => the reference's outputs (tests/golden/siblings.npz)."""
    fx = load_golden("siblings.npz")
    torch.manual_seed(0)
    layer = _sibling(name)
    sd = {k[len(name) + 4:]: torch.tensor(fx[k]) for k in fx.files if k.startswith(name + "/sd/")}
    assert set(sd) == set(layer.state_dict()), (sorted(sd), sorted(layer.state_dict()))
    layer.load_state_dict(sd)
    x = torch.tensor(fx[f"{name}/x"])
    with torch.no_grad():
        z, ld = layer.forward(x)
        xi, ldi = layer.inverse(x)
    tol = 4e-5 if "RQS" in name else 1e-5
    for mine, key in ((z, "z"), (ld, "ld"), (xi, "xinv"), (ldi, "ldinv")):
        ref = fx[f"{name}/{key}"]
        err = np.max(np.abs(mine.numpy() - ref) / np.maximum(1.0, np.abs(ref)))
        assert err < tol, (name, key, err)


def test_flow_mixture():
    """Reference flows.py:716-829: log-sum-exp of the components, one component per sample."""
    from torchflows_amd.flows import FlowMixture
    torch.manual_seed(0)
    flows = [tfa.Flow(tfa.RealNVP(3)).eval(), tfa.Flow(tfa.NICE(3)).eval()]
    mix = FlowMixture(flows, weights=[0.25, 0.75])
    x = torch.randn(11, 3)
    with torch.no_grad():
        lp = mix.log_prob(x)
        parts = torch.stack([f.log_prob(x) for f in flows]) + torch.log(torch.tensor([0.25, 0.75]))[:, None]
        assert torch.allclose(lp, torch.logsumexp(parts, dim=0), atol=1e-6)
        s, slp = mix.sample(500, return_log_prob=True)
    assert s.shape == (500, 3) and slp.shape == (500,) and torch.isfinite(s).all()
    assert mix.n_components == 2 and abs(float(mix.weights.sum()) - 1.0) < 1e-6
    with pytest.raises(AssertionError):
        FlowMixture(flows, weights=[0.5, 0.6])


def test_bench_launcher_starts_ranks_without_touching_torch():
    """`python bench.py --gpus 2` (no launcher around it) must start its ranks as a child process and relay their exit
    code.  Without a GPU the two ranks stop at bench.py's own "needs a GPU" assertion -- which proves that both were
    started through torch.distributed.run and that the parent relayed the failure instead of printing a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert out.stdout.strip() == ""
    assert out.stderr.count("bench.py needs a GPU") >= 2, out.stderr[-2000:]


def test_invalidate_native_caches_clears_every_pack():
    """the packed-weight caches are keyed on version counters / data pointers, which an edit through
    ``.data`` does not move; ``invalidate_native_caches`` (public, also run by train() / eval() / load_state_dict)
    must drop EVERY ``_tfk_*`` cache, not just the compiled flow programs."""
    import pickle
    import torchflows_amd as tfa
    flow = tfa.Flow(tfa.MAF(6, n_layers=2))
    mods = list(flow.modules())
    for i, m in enumerate(mods):
        m.__dict__["_tfk_compiled"] = {"k": i}
        m.__dict__["_tfk_made_pack"] = (0, None)
        m.__dict__["_tfk_ew_block"] = (0, None)
        m.__dict__["_tfk_bn_affine"] = (0, None)
        m.__dict__["_tfk_slots"] = []
    flow.invalidate_native_caches()
    left = {k for m in mods for k in m.__dict__ if k.startswith("_tfk_")}
    assert left <= {"_tfk_slots", "_tfk_declined_warned"}, left          # (structural: tensor identities, no values -- kept on purpose)
    for trigger in (lambda: flow.eval(), lambda: flow.train(), lambda: flow.load_state_dict(flow.state_dict()),
                    lambda: flow.bijection.invalidate_native_caches()):
        for m in mods:
            m.__dict__["_tfk_made_pack"] = (0, None)
        trigger()
        assert not any(k.startswith("_tfk_") and k != "_tfk_slots" for m in flow.bijection.modules() for k in m.__dict__)
    # the direction tags on the methods are not caches and survive; the module still pickles / deep-copies
    from torchflows_amd.bijections.base import method_direction
    assert method_direction(flow.bijection.forward) == 0
    pickle.loads(pickle.dumps(flow))


def test_torchflows_import_alias_resolves_to_the_build():
    """north_star: "drops into Flow.log_prob/.sample unchanged" -- the reference's import paths
    (test/test_cuda.py:4, test/test_fit.py ...) resolve to this build's modules, the same objects as the
    ``torchflows_amd`` spelling."""
    code = (
        "import torch\n"
        "from torchflows.flows import Flow\n"
        "from torchflows.bijections.finite.autoregressive.architectures import RealNVP, CouplingRQNSF\n"
        "from torchflows.bijections.finite.autoregressive.layers import AffineCoupling\n"
        "from torchflows.bijections.base import Bijection, BijectiveComposition, invert\n"
        "from torchflows.bijections.finite.multiscale.architectures import AffineGlow\n"
        "from torchflows.base_distributions.gaussian import DiagonalGaussian\n"
        "from torchflows.utils import get_batch_shape\n"
        "import torchflows, torchflows_amd\n"
        "import torchflows_amd.flows as f2, torchflows_amd.bijections.finite.autoregressive.architectures as a2\n"
        "assert Flow is f2.Flow and RealNVP is a2.RealNVP and torchflows.Flow is f2.Flow\n"
        "import torchflows.bijections.finite.autoregressive.architectures as a1\n"
        "assert a1 is a2\n"
        "torch.manual_seed(0)\n"
        "flow = Flow(RealNVP(3))\n"
        "x = torch.randn(1000, 3)\n"
        "lp = flow.log_prob(x); xs = flow.sample((1000,))\n"
        "assert lp.shape == (1000,) and xs.shape == (1000, 3)\n"
        "assert isinstance(flow.bijection, torchflows_amd.bijections.base.Bijection)\n"
        "try:\n"
        "    import torchflows.bijections.continuous.rnode\n"
        "    raise SystemExit('out-of-scope module resolved')\n"
        "except ModuleNotFoundError:\n"
        "    pass\n"
        "print('ALIAS_OK')\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert out.returncode == 0 and "ALIAS_OK" in out.stdout, out.stdout + out.stderr


def test_reference_spelled_package_imports_and_kl_fit():
    """the reference re-exports its presets from the PACKAGES
    (bijections/finite/autoregressive/__init__.py:1-22, multiscale/__init__.py:1-16, matrix/__init__.py:4); every
    in-scope name resolves by that spelling through the ``torchflows`` alias to this build's class.  And
    ``BaseFlow.fit_kl_p_to_q`` / ``_loss_kl_p_to_q`` (flows.py:79-197) exist and train: forward KL to a Gaussian target
    falls and the flow is left in eval mode."""
    code = (
        "import torch\n"
        "from torchflows.bijections.finite.autoregressive import (NICE, RealNVP, MAF, IAF, CouplingRQNSF,\n"
        "    MaskedAutoregressiveRQNSF, InverseAutoregressiveRQNSF, CouplingLRS, MaskedAutoregressiveLRS,\n"
        "    InverseAutoregressiveLRS)\n"
        "from torchflows.bijections.finite.multiscale import MultiscaleNICE, MultiscaleRealNVP, AffineGlow, ShiftGlow\n"
        "from torchflows.bijections.finite.matrix import ReversePermutationMatrix, RandomPermutationMatrix\n"
        "from torchflows.flows import Flow, BaseFlow, FlowMixture\n"
        "import torchflows_amd as tfa\n"
        "import torchflows_amd.bijections.finite.multiscale.architectures as ms\n"
        "assert RealNVP is tfa.RealNVP and NICE is tfa.NICE and CouplingRQNSF is tfa.CouplingRQNSF\n"
        "assert AffineGlow is ms.AffineGlow and MultiscaleRealNVP is ms.MultiscaleRealNVP\n"
        "try:\n"
        "    from torchflows.bijections.finite.autoregressive import UMNNMAF\n"
        "    raise SystemExit('out-of-scope preset resolved')\n"
        "except ImportError:\n"
        "    pass\n"
        "torch.manual_seed(0)\n"
        "flow = Flow(RealNVP(4, n_layers=2))\n"
        "tgt = torch.distributions.MultivariateNormal(torch.full((4,), 1.5), 0.25 * torch.eye(4))\n"
        "xt, xv = tgt.sample((512,)), tgt.sample((128,))\n"
        "nlp = lambda x: -tgt.log_prob(x)\n"
        "with torch.no_grad():\n"
        "    before = float(flow._loss_kl_p_to_q(xv, -nlp(xv), use_regularization=False))\n"
        "flow.fit_kl_p_to_q(xt, xv, nlp, n_epochs=40, lr=0.02, batch_size=128)\n"
        "assert not flow.training\n"
        "with torch.no_grad():\n"
        "    after = float(flow._loss_kl_p_to_q(xv, -nlp(xv), use_regularization=False))\n"
        "assert after < 0.25 * before, (before, after)\n"
        "print('SPELLING_OK', before, after)\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert out.returncode == 0 and "SPELLING_OK" in out.stdout, out.stdout + out.stderr


def test_kept_optimizer_is_never_captured():
    """``fit(reset_optimizer=False)`` keeps whatever optimiser an earlier fit built.  A FlatAdamW counts its
    steps on the host, so (i) ``Flow.fit`` must not capture a step of it into a hipGraph -- ``_optimizer_capturable`` is
    the gate -- and (ii) torch's own AdamW qualifies only with ``capturable=True`` in every group."""
    from torchflows_amd.flat_optim import FlatAdamW
    from torchflows_amd.flows import _optimizer_capturable
    ps = [torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(3, 2))]
    assert not _optimizer_capturable(FlatAdamW(ps, lr=1e-3))
    assert not _optimizer_capturable(torch.optim.AdamW(ps, lr=1e-3))
    if torch.cuda.is_available():
        assert _optimizer_capturable(torch.optim.AdamW([p.cuda() for p in ps], lr=1e-3, capturable=True))


@pytest.mark.parametrize("arch,D,n_layers,direction", [("RealNVP", 64, 8, 0), ("RealNVP", 64, 3, 1), ("NICE", 64, 4, 0),
                                                       ("RealNVP", 128, 2, 0), ("RealNVP", 22, 3, 0), ("NICE", 8, 3, 1),
                                                       ("CouplingRQNSF", 64, 3, 0), ("CouplingRQNSF", 64, 2, 1),
                                                       ("CouplingRQNSF", 128, 2, 0), ("CouplingRQNSF", 22, 2, 1),
                                                       ("CouplingRQNSF-h24", 64, 2, 0), ("CouplingRQNSF-h31", 64, 2, 1),
                                                       ("CouplingRQNSF", 256, 2, 0),
                                                       ("CouplingLRS", 64, 3, 0), ("CouplingLRS", 64, 2, 1),
                                                       ("CouplingLRS", 22, 2, 0), ("CouplingLRS-h24", 64, 2, 1),
                                                       ("CouplingLRS", 128, 2, 0),
                                                       ("RealNVP", 256, 8, 0),        # one segment, operands streamed
                                                       ("RealNVP@32", 22, 3, 0), ("NICE@32", 8, 3, 1), ("RealNVP@32", 32, 2, 1),
                                                       ("CouplingRQNSF@32", 22, 2, 0), ("CouplingLRS@32", 16, 2, 1),
                                                       ("MAF@32", 22, 2, 0), ("IAF@32", 32, 2, 1),
                                                       ("MAF", 64, 3, 0), ("IAF", 64, 2, 1), ("MAF", 22, 2, 0), ("MAF", 128, 2, 0),
                                                       ("MaskedAutoregressiveRQNSF", 64, 2, 0), ("InverseAutoregressiveRQNSF", 64, 2, 1),
                                                       ("MaskedAutoregressiveLRS", 64, 2, 0), ("MaskedAutoregressiveRQNSF", 22, 2, 0)])
@pytest.mark.parametrize("bf16x3", ["1", "0"])
def test_lean_chain_packer_against_fp64_emulator(arch, D, n_layers, direction, bf16x3, monkeypatch):
    """Host logic of the lean flow programs (fused._compile_lean: elementwise layers deferred and folded into W1 / b1,
    pre-affines, pre-scaled logits, lane-major operands): an fp64 emulator that decodes the packed blocks exactly as
    csrc/tfk_flow_chain.h reads them must reproduce the composition's forward / inverse."""
    from lean_emulator import run_lean
    from torchflows_amd import fused
    import torchflows_amd as tfa
    if "@" in arch and bf16x3 == "0" and ("RQ" in arch or "LRS" in arch):
        pytest.skip("spline chains at row width 32: bf16 x 3 operands only")
    if bf16x3 == "0" and "RQ" not in arch and "LRS" not in arch and D not in (64, 22, 8, 256):
        pytest.skip("the operand format only concerns spline chains and 64-wide affine chains")
    set_debug(monkeypatch, rqs_bf16x3=bf16x3)
    set_debug(monkeypatch, lean_bf16x3=bf16x3)
    torch.manual_seed(3)
    kw = {}
    if "-h" in arch:                                          # wider conditioner: two hidden tiles (bf16 x 3 format only)
        arch, h = arch.split("-h")
        kw = dict(conditioner_kwargs=dict(n_hidden=int(h)))
        if bf16x3 == "0":
            pytest.skip("fp32 operands stop at hidden width 16")
    if D == 256 and bf16x3 == "0" and "RQ" in arch:
        pytest.skip("CouplingRQNSF(256) has hidden width 17: bf16 x 3 operands only")
    if ("LRS" in arch or "Autoregressive" in arch) and bf16x3 == "0":
        pytest.skip("lean linear rational splines and MADE spline layers: bf16 x 3 operands only")
    Dp = D if D in (64, 128, 256) else (64 if D < 64 else 128)
    if "@" in arch:                                           # row width 32 for event sizes <= 32
        arch, w = arch.split("@")
        Dp = int(w)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=n_layers, **kw))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(512, D) * 1.5 + 0.3)
    flow.eval()
    comp = flow.bijection.double()
    order = comp.layers if direction == 0 else list(comp.layers)[::-1]
    plan = fused._flatten(order, "forward" if direction == 0 else "inverse")
    pos = torch.arange(D)
    if Dp != D:
        pos = torch.where(pos < D // 2, pos, pos - D // 2 + Dp // 2)
    chain = fused._compile_lean(comp, plan, torch.device("cpu"), D, Dp, pos.clone(), pos.clone())
    assert chain is not None and chain.D == Dp
    x = torch.randn(64, D, dtype=torch.float64)
    with torch.no_grad():
        want, ld_want = (comp.forward if direction == 0 else comp.inverse)(x)
    rows = torch.zeros(64, Dp, dtype=torch.float64)
    rows[:, pos] = x
    ld = torch.zeros(64, dtype=torch.float64)
    if D == 256 and n_layers == 8:
        assert len(chain.segments) == 1 and chain.segments[0].params.numel() * 4 > 160 * 1024
    for seg in chain.segments:
        assert seg.mfma and all(12 <= op[0] <= 28 and op[0] not in (19, 20) for op in seg.ops)
        rows, l = run_lean(seg.ops, seg.params, rows, Dp)
        ld = ld + l
    got = rows[:, chain.pos]
    # the blocks are stored in fp32: agreement at fp32 resolution of the weights
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5
    assert float((ld - ld_want).abs().max() / max(1.0, float(ld_want.abs().max()))) < 2e-5


def test_log_prob_and_sum_on_host():
    """Flow.log_prob_and_sum off the HIP path: log_prob followed by an fp64 reduction (1-element float64 tensor)."""
    import torchflows_amd as tfa
    torch.manual_seed(0)
    flow = tfa.Flow(tfa.RealNVP(6)).eval()
    x = torch.randn(100, 6)
    with torch.no_grad():
        lp, total = flow.log_prob_and_sum(x)
        assert torch.equal(lp, flow.log_prob(x))
    assert total.dtype == torch.float64 and total.shape == (1,)
    assert abs(float(total) - float(lp.double().sum())) < 1e-9


def test_cached_parameter_slots_follow_module_edits():
    """The L2 term and the autograd node keep (module, name) slots instead of walking named_modules every step: a
    replaced Parameter and a flipped requires_grad are seen at once; structural edits after invalidate_native_caches()."""
    import torchflows_amd as tfa
    from torchflows_amd import autograd as A
    torch.manual_seed(0)
    flow = tfa.Flow(tfa.RealNVP(6))
    comp = flow.bijection
    want = sum(float((p.detach() ** 2).sum()) * 0.0 for p in comp.parameters())      # (touch every parameter once)
    r0 = float(comp.regularization())
    lin = [m for m in comp.modules() if isinstance(m, torch.nn.Linear)][0]
    with torch.no_grad():
        lin.weight = torch.nn.Parameter(lin.weight.detach() * 3.0)                     # replaced, same slot
    r1 = float(comp.regularization())
    assert r1 > r0 and want == 0.0
    ref = sum(float(layer.l2_coef) * sum(float((p.detach() ** 2).sum()) for p in layer.parameters() if p.requires_grad)
              for layer in comp.layers if getattr(layer, "l2_regularization", False) and hasattr(layer, "l2_coef"))
    assert abs(r1 - ref) <= 1e-5 * max(1.0, abs(ref))
    lin.weight.requires_grad_(False)                                                   # leaves the L2 term
    r2 = float(comp.regularization())
    assert r2 < r1
    ct = [layer.conditioner_transform for layer in comp.layers
          if getattr(layer, "conditioner_transform", None) is not None and list(layer.conditioner_transform.parameters())][0]
    assert [id(p) for p in A._module_params(ct)] == [id(p) for p in ct.parameters()]
    ct.extra = torch.nn.Linear(2, 2)                                                   # structural edit
    comp.invalidate_native_caches()
    assert [id(p) for p in A._module_params(ct)] == [id(p) for p in ct.parameters()]


@pytest.mark.parametrize("arch,D,C,direction", [("RealNVP", 64, 8, 0), ("RealNVP", 64, 3, 1), ("NICE", 64, 5, 0),
                                                ("CouplingRQNSF", 64, 8, 0), ("CouplingRQNSF", 64, 2, 1),
                                                ("CouplingLRS", 64, 16, 0), ("RealNVP", 22, 4, 0), ("RealNVP", 128, 6, 0)])
def test_lean_context_programs_against_fp64_emulator(arch, D, C, direction):
    """Conditional flows as ONE lean program (fused._compile_lean(context=True)): the context's columns of W1 as further
    GEMM-1 k-steps (A1c), the context-conditioned elementwise layers (TFK_OP_EWC_*, scale logits pre-scaled for exp2) and
    constant ones (TFK_OP_EW_FMA) in front of / behind the couplings -- decoded by the fp64 emulator exactly as
    csrc/tfk_flow_chain.h / tfk_flow_rqs_chain.h read them, against the composition's forward / inverse with context."""
    from lean_emulator import run_lean
    from torchflows_amd import fused
    import torchflows_amd as tfa
    torch.manual_seed(4)
    flow = tfa.Flow(getattr(tfa, arch)(D, context_shape=(C,), n_layers=3))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(512, D) * 1.5 + 0.3, context=torch.randn(512, C))
    flow.eval()
    comp = flow.bijection.double()
    Dp = D if D in (64, 128, 256) else (64 if D < 64 else 128)
    order = comp.layers if direction == 0 else list(comp.layers)[::-1]
    plan = fused._flatten(order, "forward" if direction == 0 else "inverse")
    pos = torch.arange(D)
    if Dp != D:
        pos = torch.where(pos < D // 2, pos, pos - D // 2 + Dp // 2)
    chain = fused._compile_lean(comp, plan, torch.device("cpu"), D, Dp, pos.clone(), pos.clone(), context=True)
    assert chain is not None and len(chain.segments) == 1 and chain.D == Dp
    x, c = torch.randn(64, D, dtype=torch.float64), torch.randn(64, C, dtype=torch.float64)
    with torch.no_grad():
        want, ld_want = (comp.forward if direction == 0 else comp.inverse)(x, context=c)
    rows = torch.zeros(64, Dp, dtype=torch.float64)
    rows[:, pos] = x
    seg = chain.segments[0]
    kinds = [op[0] for op in seg.ops]
    assert any(k in (19, 20) for k in kinds) and any((op[1] >> 4) == (C + 3) // 4 for op in seg.ops)
    rows, ld = run_lean(seg.ops, seg.params, rows, Dp, context=c)
    got = rows[:, chain.pos]
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5
    assert float((ld - ld_want).abs().max() / max(1.0, float(ld_want.abs().max()))) < 2e-5


def test_compiled_programs_retire_on_replacement_move_and_context_width():
    """(i) a replaced Parameter retires EVERY cached program of the composition (the other direction was
    served stale when its version happened to match), (ii) moving a CHILD retires the parent's programs (a process-wide
    epoch, bumped by every ``_apply``), (iii) a context of another width is rejected like the reference's Linear layer
    does instead of being truncated / zero-padded to the kernel's k-steps, (iv) sharded_fit refuses an empty shard
    up front (that rank would never join the ActNorm all-reduce) and keeps one loss per step."""
    import torchflows_amd as tfa
    from torchflows_amd import fused
    from torchflows_amd.distributed import sharded_fit
    cpu = torch.device("cpu")
    torch.manual_seed(0)
    flow = tfa.Flow(tfa.RealNVP(64, n_layers=2)).eval()
    comp = flow.bijection
    fwd0, inv0 = fused.get_compiled(comp, 0, cpu), fused.get_compiled(comp, 1, cpu)
    assert fwd0 is not None and inv0 is not None
    assert fused.get_compiled(comp, 0, cpu) is fwd0 and fused.get_compiled(comp, 1, cpu) is inv0
    lin = [m for m in comp.modules() if isinstance(m, torch.nn.Linear)][0]
    lin.weight = torch.nn.Parameter(lin.weight.detach() * 2.0)              # new tensor, version counter 0 like the old
    fwd1 = fused.get_compiled(comp, 0, cpu)
    inv1 = fused.get_compiled(comp, 1, cpu)
    assert fwd1 is not fwd0 and inv1 is not inv0
    assert not torch.equal(inv1.segments[0].params, inv0.segments[0].params)
    # (ii) only a child is converted: the parent's entries must not survive
    comp.layers[1].double()
    comp.layers[1].float()
    assert fused.get_compiled(comp, 0, cpu) is not fwd1 and fused.get_compiled(comp, 1, cpu) is not inv1
    # (iii) context width
    torch.manual_seed(1)
    cflow = tfa.Flow(tfa.RealNVP(64, n_layers=2, context_shape=(3,))).eval()
    chain = fused.get_compiled(cflow.bijection, 0, cpu, context=True)
    if chain is not None:                                                      # (declined without the lean kernels: nothing to check)
        assert chain.ctx_width == 3
        with pytest.raises(ValueError):
            fused.run_chain(chain, torch.zeros(4, 64), want_rows=True, context=torch.zeros(4, 4))
    # (iii-b) a hand-built composition whose FIRST layer takes no context (an ActNorm in front): the
    # composition's own context_shape is None (bijections/base.py:203-209) and the width comes from the couplings
    from torchflows_amd.bijections.base import BijectiveComposition
    from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm
    layers = list(cflow.bijection.layers)
    first = next(i for i, l in enumerate(layers) if isinstance(l, ActNorm))
    reordered = BijectiveComposition([layers[first]] + layers[:first] + layers[first + 1:]).eval()
    assert reordered.context_shape is None
    chain = fused.get_compiled(reordered, 0, cpu, context=True)                # (raised TypeError before)
    if chain is not None:
        assert chain.ctx_width == 3
    # (iv)
    small = tfa.Flow(tfa.RealNVP(4, n_layers=1))
    losses = sharded_fit(small, torch.randn(40, 4), n_epochs=3, batch_size=16, lr=1e-3)
    assert len(losses) == 9 and all(np.isfinite(losses))


@pytest.mark.parametrize("arch,D,want_w", [("RealNVP", 3, 16), ("RealNVP", 15, 16), ("NICE", 21, 32), ("RealNVP", 63, 64),
                                           ("RealNVP", 99, 128), ("NICE", 201, 256), ("CouplingRQNSF", 7, 32),
                                           ("CouplingRQNSF", 63, 64)])
def test_odd_event_sizes_compile_to_lean_programs(arch, D, want_w):
    """The packer's side of odd event sizes (no GPU needed: the programs are packed on the host): the narrowest row width
    with (D + 1) / 2 columns per plane, ONE segment, the middle element reserved the last column of both planes -- it starts
    in plane B's, every coupling that finds it in its source plane carries the move bit (bit 2 of src_plane) and the source
    planes still alternate --, and the final layout a permutation of D distinct columns."""
    import torchflows_amd as tfa
    from torchflows_amd import fused as fz
    torch.manual_seed(D)
    flow = tfa.Flow(getattr(tfa, arch)(D, n_layers=4))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(128, D))
    flow.eval()
    for direction in (0, 1):
        chain = fz.compile_chain(flow.bijection, direction, torch.device("cpu"))
        assert chain is not None and chain.D == want_w and len(chain.segments) == 1 and chain.D_log == D
        h, hp = D // 2, want_w // 2
        assert chain.pos_in[:h].tolist() == list(range(h)) and int(chain.pos_in[h]) == want_w - 1
        assert chain.pos_in[h + 1:].tolist() == [hp + i for i in range(h)]
        assert len(set(chain.pos.tolist())) == D and int(chain.pos.max()) < want_w
        couplings = [op for op in chain.segments[0].ops if op[0] != fz.OP_EW_FMA]
        assert len(couplings) == 4
        planes = [op[1] & 1 for op in couplings]
        assert all(a != b for a, b in zip(planes, planes[1:]))                  # the source planes alternate
        moves = [bool(op[1] & 4) for op in couplings]
        # the middle element sits in plane B at first: a first coupling that reads plane B must take it over, and from then
        # on every coupling does (its source plane is the previous one's target plane)
        assert moves[0] == (planes[0] == 1) and all(moves[1:]), (planes, moves)
