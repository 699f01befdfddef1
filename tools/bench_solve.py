import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cumf_als_amd import als
f = int(sys.argv[1]) if len(sys.argv) > 1 else 100
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
g = torch.Generator(device="cuda"); g.manual_seed(0)
T = torch.rand((batch, 128, f), device="cuda", generator=g) * 0.2
A = torch.bmm(T.transpose(1, 2), T) + 0.048 * 128 * torch.eye(f, device="cuda")
b = torch.rand((batch, f), device="cuda", generator=g)
x = torch.zeros_like(b)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
tl = t(lambda: als.lu_solve(A, b, x))
xr = torch.linalg.solve(A.double(), b.double().unsqueeze(-1)).squeeze(-1)
print(f"f={f} batch={batch} LU   {tl*1e3:8.2f} ms  {tl/batch*1e9:8.1f} ns/system  err={float((x-xr).abs().max()):.2e}")
if len(sys.argv) > 3 and sys.argv[3] == "lu":
    sys.exit(0)
os.environ["CUMF_ALS_LU_EXACT"] = "1"
tl = t(lambda: als.lu_solve(A, b, x))
print(f"f={f} batch={batch} LUex {tl*1e3:8.2f} ms  {tl/batch*1e9:8.1f} ns/system  err={float((x-xr).abs().max()):.2e}")
for it in (0, 6):
    def cg():
        x.zero_(); als.cg_solve(A, x, b, it)
    tc = t(cg)
    print(f"f={f} batch={batch} CG{it}  {tc*1e3:8.2f} ms  {tc/batch*1e9:8.1f} ns/system  err={float((x-xr).abs().max()):.2e}")
