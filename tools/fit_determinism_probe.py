#!/usr/bin/env python3
"""Are repeated Flow.fit calls on an image flow the same fit, bit for bit?  (captured / eager steps, captured / eager
validation pass; row order fixed)    python tools/fit_determinism_probe.py"""
import copy, os, sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
from torchflows_amd.architectures import MultiscaleRealNVP
torch.manual_seed(0)
x = torch.randn(200, 1, 28, 28); xv = torch.randn(64, 1, 28, 28)
base = tfa.Flow(MultiscaleRealNVP((1, 28, 28)))
def run(graph, valg, val=True):
    os.environ["TORCHFLOWS_AMD_GRAPH"] = graph
    os.environ["TORCHFLOWS_AMD_DEBUG"] = "" if valg else "val_graph=0"
    torch.manual_seed(1)
    f = copy.deepcopy(base).cuda()
    f.fit(x, x_val=xv if val else None, n_epochs=10, batch_size=200, lr=0.01, shuffle=False)
    return f._fit_stats.get("val_loss"), torch.cat([p.detach().flatten() for p in f.parameters()])
for name, a, b in (("graph,noval x2", ("1", False, False), ("1", False, False)),
                   ("graph,eagerval x2", ("1", False), ("1", False)),
                   ("graph: valgraph vs eagerval", ("1", True), ("1", False)),
                   ("eager x2", ("0", False), ("0", False))):
    ra, rb = run(*a), run(*b)
    print(name, ra[0], rb[0], float((ra[1] - rb[1]).abs().max()), flush=True)
