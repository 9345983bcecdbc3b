import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vita_amd import ops
torch.manual_seed(1)
noise = torch.rand(2, 23040, device="cuda")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("argsort 2x23040: %.1f us" % timeit(lambda: ops.argsort_mask(noise, 6912)))
sh, rs, mk = ops.argsort_mask(noise, 6912)
print("match torch stable:", torch.equal(sh, torch.argsort(noise, dim=1, stable=True)))
for n, l in ((4, 2304), (3, 64), (2, 11520), (1, 7), (5, 1001)):
    x = torch.rand(n, l, device="cuda"); x[:, ::3] = x[:, 0:1]   # forced ties
    sh, rs, mk = ops.argsort_mask(x, l // 2)
    print(n, l, torch.equal(sh, torch.argsort(x, dim=1, stable=True)), torch.equal(rs, torch.argsort(sh, dim=1)))
