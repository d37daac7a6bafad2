import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TORCHFLOWS_AMD_GRAPH"] = "0"
from torchflows.flows import Flow
from torchflows.bijections.finite.multiscale.architectures import AffineGlow
torch.manual_seed(0)
flow = Flow(AffineGlow((3, 32, 32))).cuda()
x = torch.randn(1024, 3, 32, 32, device="cuda")
w = torch.ones(1024, device="cuda")
flow.fit(x.cpu(), n_epochs=3, batch_size=1024)
flow.train()
opt = flow._optimizer
def step():
    opt.zero_grad()
    loss = flow._base_batch_loss((x, w), reduction=torch.mean, use_regularization=True)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
