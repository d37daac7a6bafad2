"""Child process of tests/test_gpu_train.py::test_fit_with_hipgraph_replay_in_a_subprocess."""
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchflows_amd.flows import Flow  # noqa: E402
from torchflows_amd.bijections.finite.autoregressive.architectures import RealNVP  # noqa: E402

torch.manual_seed(0)
D = 64
mix = torch.randn(8192, D)
x = torch.cat([mix[:, :32] * 0.3 + 2.0, torch.tanh(mix[:, 32:]) + 0.1 * mix[:, :32]], dim=1).cuda()
flow = Flow(RealNVP(D, n_layers=2)).cuda()
flow.train()
with torch.no_grad():
    flow.log_prob(x)
flow.eval()
with torch.no_grad():
    before = float(flow.log_prob(x).mean())
graph, eager = copy.deepcopy(flow), copy.deepcopy(flow)
os.environ["TORCHFLOWS_AMD_GRAPH"] = "1"
graph.fit(x, n_epochs=5, lr=0.01, x_val=x[:1024], shuffle=False)
os.environ["TORCHFLOWS_AMD_GRAPH"] = "0"
eager.fit(x, n_epochs=5, lr=0.01, x_val=x[:1024], shuffle=False)
with torch.no_grad():
    out = {"before": before, "after_graph": float(graph.log_prob(x).mean()),
           "after_eager": float(eager.log_prob(x).mean()),
           "graph_stats": graph._fit_stats, "eager_stats": eager._fit_stats}
print(json.dumps(out))
