"""Host logic of the ConvNet training route (torchflows_amd/convnet_train.py): which networks it accepts, that CPU tensors
never take it, and the re-evaluation rule for BatchNorm on the ATen path (no GPU needed)."""
import copy

import torch

from torchflows_amd import convnet_train
from torchflows_amd.bijections.finite.multiscale.conditioning.classic import ConvNet


def test_only_the_reference_network_shape_is_accepted():
    assert convnet_train.structure_ok(ConvNet((3, 16, 32), 64))
    assert convnet_train.structure_ok(ConvNet((2, 7, 14), 64))            # odd sizes: 2-wide modifier kernels
    assert not convnet_train.structure_ok(ConvNet((3, 16, 32), 64, kernels=(8, 4)))
    assert not convnet_train.structure_ok(ConvNet((3, 16, 32), 64, kernels=(4, 8, 4)))
    net = ConvNet((3, 16, 32), 64)
    net.blocks[2].bn.momentum = None                                       # cumulative average: not covered
    net.__dict__.pop("_tfk_ct_structure", None)
    assert not convnet_train.structure_ok(net)


def test_host_tensors_keep_the_aten_path():
    net = ConvNet((1, 14, 28), 32)
    x = torch.randn(5, 1, 14, 28, requires_grad=True)
    assert not convnet_train.usable(net, x)
    assert not convnet_train.static_usable(net, torch.device("cuda", 0))      # parameters live on the host
    out = net(x)
    assert out.shape == (5, 32) and out.grad_fn is not None


def test_a_repeated_evaluation_leaves_the_running_statistics_alone():
    torch.manual_seed(0)
    net = ConvNet((1, 14, 28), 32).train()
    x = torch.randn(7, 1, 14, 28)
    first = net(x)
    state = copy.deepcopy({k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k})
    with convnet_train.recomputing():
        again = net(x)
    assert torch.allclose(first, again, atol=1e-6)
    for k, v in net.state_dict().items():
        if k in state:
            assert torch.equal(v, state[k]), k
    net(x)                                                                  # (outside: the batch counts again)
    assert int(net.blocks[1].bn.num_batches_tracked) == 2
