import sys, time, torch, os
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
from torchflows_amd import native, fused
N = 1 << 20
for D in (62, 60, 58, 46, 34):
    torch.manual_seed(0)
    flow = tfa.Flow(tfa.RealNVP(D, n_layers=8))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(4096, D))
    flow = flow.eval().cuda()
    x = torch.randn(N, D, device="cuda")
    with torch.no_grad():
        before = native.calls
        lp = flow.log_prob(x)
        launches = native.calls - before
        ch = fused.get_compiled(flow.bijection, 0, x.device)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): lp = flow.log_prob(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    H = [m.out_features for m in flow.modules() if isinstance(m, torch.nn.Linear)][0]
    print(f"RealNVP({D}) H={H}: {N / dt:.3e} evals/s, {dt * 1e3:.3f} ms, {launches} launches, Dp={ch.D}, segs={[len(s.ops) for s in ch.segments]}", flush=True)
