import sys, time, torch, os
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torchflows_amd as tfa
from torchflows_amd import native
N = 1 << 19
for arch, D, C in (("RealNVP", 256, 8), ("RealNVP", 128, 8), ("NICE", 64, 8)):
    torch.manual_seed(0)
    flow = tfa.Flow(getattr(tfa, arch)(D, context_shape=(C,), n_layers=8))
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(2048, D), context=torch.randn(2048, C))
    flow = flow.eval().cuda()
    x, c = torch.randn(N, D, device="cuda"), torch.randn(N, C, device="cuda")
    for lean in ("1", "0"):
        os.environ["TORCHFLOWS_AMD_DEBUG"] = "lean=" + str(lean)
        flow.invalidate_native_caches()
        with torch.no_grad():
            before = native.calls
            lp = flow.log_prob(x, context=c)
            launches = native.calls - before
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): lp = flow.log_prob(x, context=c)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{arch}({D}) context {C} lean={lean}: {N / dt:.3e} evals/s, {dt * 1e3:.3f} ms, {launches} launches", flush=True)
