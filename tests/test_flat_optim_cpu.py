"""FlatAdamW (torchflows_amd/flat_optim.py) on the host: the parameters of a module re-homed as views of one buffer, the
update bit-identical to torch.optim.AdamW whichever way the gradients arrive."""
import copy

import pytest

from conftest import set_debug
import torch

from torchflows_amd.flat_optim import FlatAdamW, FlatParams, lookup


def _nets():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net.register_parameter("empty", torch.nn.Parameter(torch.zeros(0)))
    net.register_parameter("frozen", torch.nn.Parameter(torch.ones(4), requires_grad=False))
    return net, copy.deepcopy(net)


def test_layout_and_lookup():
    net, _ = _nets()
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    opt = FlatAdamW(net.parameters(), lr=0.05)
    fb = opt.flat
    assert isinstance(fb, FlatParams) and fb.intact() and len(fb.params) == len(names)
    assert all(o % 4 == 0 for o in fb.offset) and fb.n % 4 == 0 and sum(fb.split_sizes) == fb.n
    for n, p in net.named_parameters():                       # same objects, same values, same shapes
        assert torch.equal(p, before[n]) and p.shape == before[n].shape
    assert float(fb.P[fb.zero_slot]) == 0.0
    assert lookup(list(net.parameters())) is fb
    assert lookup([p for p in net.parameters()] + [torch.nn.Parameter(torch.zeros(2))]) is None
    net[0].weight.data = net[0].weight.data.clone()           # re-homed by hand: no longer intact
    assert not fb.intact() and lookup(list(net.parameters())) is None
    sd = net.state_dict()                                     # the module's own view of itself is unchanged
    assert set(sd) == set(before)


@pytest.mark.parametrize("flat_grads", [False, True])
def test_trajectory_is_bit_identical_to_torch_adamw(flat_grads):
    net, ref = _nets()
    o1 = FlatAdamW(net.parameters(), lr=0.05)
    o2 = torch.optim.AdamW(ref.parameters(), lr=0.05)
    fb = o1.flat
    x = torch.randn(16, 7)
    for it in range(7):
        o1.zero_grad()
        o2.zero_grad()
        net(x).square().sum().backward()
        ref(x).square().sum().backward()
        if flat_grads:                                        # the gradients as slices of one buffer (autograd.py's form)
            G = torch.zeros(fb.n)
            for p, o, n in zip(fb.params, fb.offset, fb.numel):
                if n:                                         # (the empty parameter takes no part in the loss)
                    G[o:o + n].copy_(p.grad.reshape(-1))
                p.grad = G[o:o + n].view(p.shape)
            fb.last_grad = G
            assert fb.grads_are_flat() is G
        versions = [p._version for p in fb.params]
        o1.step()
        o2.step()
        assert all(p._version > v for p, v in zip(fb.params, versions) if p.grad is not None and p.numel())
        for (n, a), b in zip(net.named_parameters(), ref.parameters()):
            assert torch.equal(a, b), (it, n)
    assert (o1.fast_steps, o1.general_steps) == ((7, 0) if flat_grads else (0, 7))
    assert float(fb.P[fb.zero_slot]) == 0.0 and fb.intact()


def test_missing_gradients_are_skipped_like_torch_does():
    net, ref = _nets()
    o1 = FlatAdamW(net.parameters(), lr=0.05)
    o2 = torch.optim.AdamW(ref.parameters(), lr=0.05)
    x = torch.randn(16, 7)
    for it in range(5):
        o1.zero_grad()
        o2.zero_grad()
        net(x).square().sum().backward()
        ref(x).square().sum().backward()
        if it % 2:                                            # the last layer sits this step out
            net[2].weight.grad = net[2].bias.grad = None
            ref[2].weight.grad = ref[2].bias.grad = None
        o1.step()
        o2.step()
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.equal(a, b), it


def test_make_adamw_picks_it_only_where_it_pays(monkeypatch):
    from torchflows_amd.utils import make_adamw
    net, _ = _nets()
    set_debug(monkeypatch, flat_adamw=None)
    assert type(make_adamw(net.parameters(), 0.1)) is torch.optim.AdamW          # host parameters: torch's own
    set_debug(monkeypatch, flat_adamw="1")
    assert isinstance(make_adamw(net.parameters(), 0.1), FlatAdamW)
    set_debug(monkeypatch, flat_adamw="0")
    assert type(make_adamw(net.parameters(), 0.1)) is torch.optim.AdamW


def test_fit_on_the_host_with_the_flat_optimiser_matches(monkeypatch):
    """Flow.fit end to end with the buffer-homed parameters (forced on the host): the same weights as with torch's AdamW,
    best-weight snapshots and load_state_dict included (they copy in place, the parameters stay in the buffer)."""
    from torchflows_amd import Flow, RealNVP
    torch.manual_seed(1)
    x = torch.randn(512, 6) * 0.5 + 1.0
    a = Flow(RealNVP(6, n_layers=2))
    b = copy.deepcopy(a)
    set_debug(monkeypatch, flat_adamw="1")
    a.fit(x, n_epochs=4, lr=0.01, batch_size=128, shuffle=False, x_val=x[:64])
    assert isinstance(a._optimizer, FlatAdamW) and a._optimizer.flat.intact()
    set_debug(monkeypatch, flat_adamw="0")
    b.fit(x, n_epochs=4, lr=0.01, batch_size=128, shuffle=False, x_val=x[:64])
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(pa, pb), n


def test_flows_and_their_optimiser_survive_deepcopy_and_pickle(monkeypatch):
    """copy.deepcopy / pickle of a fitted flow: the caches of the HIP path (``_tfk_*``: device tensors, ctypes arrays, weak
    references) stay behind, and the copied FlatAdamW re-homes the COPY's parameters at first use."""
    import io
    import pickle
    import weakref
    from torchflows_amd import Flow, RealNVP
    torch.manual_seed(2)
    x = torch.randn(256, 6)
    flow = Flow(RealNVP(6, n_layers=2))
    set_debug(monkeypatch, flat_adamw="1")
    flow.fit(x, n_epochs=2, lr=0.01, batch_size=128, shuffle=False)
    assert isinstance(flow._optimizer, FlatAdamW)
    anchor = torch.nn.Linear(1, 1)
    flow.bijection.layers[0].__dict__["_tfk_probe"] = weakref.ref(anchor)       # what a device cache may hold
    flow.__dict__["_tfk_probe"] = weakref.ref(anchor)
    twin = copy.deepcopy(flow)
    back = pickle.loads(pickle.dumps(flow))
    buf = io.BytesIO()
    torch.save(flow, buf)
    for other in (twin, back):
        assert "_tfk_probe" not in other.__dict__ and "_tfk_probe" not in other.bijection.layers[0].__dict__
        for (n, a), b in zip(flow.named_parameters(), other.parameters()):
            assert torch.equal(a, b) and (a.numel() == 0 or a.data_ptr() != b.data_ptr()), n
        other.fit(x, n_epochs=1, lr=0.01, batch_size=128, shuffle=False, reset_optimizer=False)   # the copied optimiser works
        assert other._optimizer.flat.intact() and other._optimizer.flat is not flow._optimizer.flat
    assert flow._optimizer.flat.intact()                      # ... and the original was not disturbed
