import sys, time, torch
sys.path.insert(0, "/root/repo")
from torchflows_amd import native
N = 1 << 18
A = torch.randn(N, 768, device="cuda"); B = torch.randn(N, 16, device="cuda"); out = torch.empty(768 * 16, device="cuda")
for _ in range(5): native.rows_outer(A, 768, B, out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): native.rows_outer(A, 768, B, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"rows_outer 2^18 x 768: {1e6 * dt:.1f} us, {4 * N * 784 / dt / 1e12:.2f} TB/s")
ref = (A.double().t() @ B.double())
t, q, j, r = torch.meshgrid(torch.arange(48), torch.arange(4), torch.arange(16), torch.arange(4), indexing="ij")
col = 64 * (t >> 2) + 4 * (4 * q + r) + (t & 3)
print("max err", float((out.double().cpu() - ref.cpu()[col, j].reshape(-1)).abs().max()), "scale", float(ref.abs().max()))
