#!/usr/bin/env python3
"""Launch times of the ConvNet training kernels (csrc/tfk_convtrain.hip) on their own, HIP events over 20 launches:
   python tools/convtrain_bench.py [N] [c h w]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from torchflows_amd import native
from torchflows_amd.bijections.finite.multiscale.conditioning.classic import ConvNet

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (3, 16, 32)
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ConvNet(shape, 2 * shape[0] * shape[1] * shape[2]).to(dev).train()
M = net.linear.out_features
x = torch.randn(N, *shape, device=dev)
b = net.blocks
P = lambda t: t.detach()


def timed(label, fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:28s} {1e3 * e0.elapsed_time(e1) / reps:9.1f} us", flush=True)
    return out


a0 = timed("frame_fwd (modifier 1)", lambda: native.convnet_train_frame_fwd(x, None, P(b[0].conv.weight), P(b[0].conv.bias), 32, 32))
y1, i1, s1 = timed("block_fwd 4->8 @32", lambda: native.convnet_train_block_fwd(a0, None, P(b[1].conv.weight), P(b[1].conv.bias), b[1].bn, True, False))
y2, i2, s2 = timed("block_fwd 8->8 @16", lambda: native.convnet_train_block_fwd(y1, s1, P(b[2].conv.weight), P(b[2].conv.bias), b[2].bn, True, False))
y3, i3, s3 = timed("block_fwd 8->4 @8", lambda: native.convnet_train_block_fwd(y2, s2, P(b[3].conv.weight), P(b[3].conv.bias), b[3].bn, True, False))
a16 = timed("frame_fwd (modifier 2)", lambda: native.convnet_train_frame_fwd(y3, s3, P(b[4].conv.weight), P(b[4].conv.bias), 4, 4)).view(N, 16)
W16, b_eff, w_frame = timed("linear_prep", lambda: native.convnet_train_linear_prep(P(net.linear.weight), P(net.linear.bias), P(b[4].conv.bias), 10, 10))
theta = timed("linear_fwd", lambda: native.convnet_train_linear_fwd(a16, W16, b_eff))
g = torch.randn_like(theta)
g16 = timed("linear_bwd_input", lambda: native.convnet_train_linear_bwd_input(g, W16))
timed("linear_wgrad", lambda: native.convnet_train_linear_wgrad(g, a16, P(b[4].conv.bias), 10, 10))
gz3, _, _, bn3 = timed("frame_bwd (modifier 2)", lambda: native.convnet_train_frame_bwd(g16.view(N, 1, 4, 4), y3, s3, P(b[4].conv.weight), s3, True))
gz2, _, _, bn2 = timed("block_bwd 8->4 @8", lambda: native.convnet_train_block_bwd(gz3, bn3[0], y3, i3, y2, s2, P(b[3].conv.weight), s2, True))
gz1, _, _, bn1 = timed("block_bwd 8->8 @16", lambda: native.convnet_train_block_bwd(gz2, bn2[0], y2, i2, y1, s1, P(b[2].conv.weight), s1, True))
g_a0, _, _, _ = timed("block_bwd 4->8 @32", lambda: native.convnet_train_block_bwd(gz1, bn1[0], y1, i1, a0, None, P(b[1].conv.weight), None, True))
timed("frame_bwd (modifier 1)", lambda: native.convnet_train_frame_bwd(g_a0, x, None, P(b[0].conv.weight), None, True))
