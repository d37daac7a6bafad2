#!/usr/bin/env python3
"""Host cost of one libtfk call from Python (enqueue only, tiny batches so that the GPU never is the bound): the
wrappers of torchflows_amd/native.py one by one, and the bare ctypes call underneath for comparison."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from torchflows_amd import native  # noqa: E402
from torchflows_amd import autograd as ag  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
N, D = 1024, 64
x = torch.randn(N, D, device=dev)
g = torch.randn(N, D, device=dev)
out = torch.empty_like(x)
gld = torch.randn(N, device=dev)
ld = torch.zeros(N, device=dev)
flow = bench.make_flow("RealNVP", 64, 8).cuda()
lay = [l for l in flow.bijection.layers if type(l).__name__ == "AffineCoupling"][0]
mlp = ag._fused_bwd_layer(lay, D)
pack = ag._TrainPack.get(D, mlp[0].out_features, dev)
pieces = torch.cat([mlp[0].weight.detach().reshape(-1), mlp[0].bias.detach(), mlp[1].weight.detach().reshape(-1),
                    mlp[1].bias.detach(), pack.zero])
packed = pieces[pack.param_index].contiguous()
acc = torch.empty(pack.n_out, device=dev)
value = torch.randn(D, 2, device=dev)
ops = [(2, 0, pack.steps2, 0)]


def timeit(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:42s} {1e6 * (t1 - t0) / n:7.2f} us per call (enqueue), {1e6 * (t2 - t0) / n:7.2f} with the drain")


timeit("flow_run_mfma (one coupling)", lambda: native.flow_run_mfma(x, out, ld, None, None, None, ops, packed))
timeit("affine_coupling_train_bwd", lambda: native.affine_coupling_train_bwd(x, g, gld, packed, pack.steps2, acc, pack.workspace))
timeit("elementwise_affine", lambda: native.elementwise_affine(x, value, out, ld, False))
timeit("elementwise_affine_bwd (with dvalue)", lambda: native.elementwise_affine_bwd(x, value, g, gld, True))
timeit("permute (reversal)", lambda: native.permute(x, None, out))
timeit("torch.empty_like", lambda: torch.empty_like(x))
timeit("aten add_ (for scale)", lambda: out.add_(1.0))
L = native.lib()
s = native._stream(x)
arr = native._pack_ops(ops)
a = (x.data_ptr(), out.data_ptr(), ld.data_ptr(), None, None, None, N, D, arr, 1, packed.data_ptr(), packed.numel(), 0, s)
timeit("bare ctypes tfk_flow_run_mfma", lambda: L.tfk_flow_run_mfma(*a))
b = (x.data_ptr(), g.data_ptr(), gld.data_ptr(), packed.data_ptr(), packed.numel(), pack.steps2, acc.data_ptr(),
     pack.workspace.data_ptr(), N, D, 0, None, 0, s)
timeit("bare ctypes tfk_affine_coupling_train_bwd", lambda: L.tfk_affine_coupling_train_bwd(*b))
