"""Fixed vs per-K-tile cost of the large-tile GEMM (development aid): time(K) at fixed M, N for a forced tile width."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vita_amd import ops
from gemm_bench import timeit

for m in (13826, 46082):
    for n, v, lb in ((1120, 14, 0), (1120, 14, 1), (4160, 16, 0), (4160, 16, 1), (3360, 14, 0)):
        res = []
        for k in (64, 576, 1088, 2176, 4160):
            a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
            b = torch.randn((n, k) if lb == 0 else (k, n), device="cuda").to(torch.bfloat16)
            c = torch.zeros(m, n, dtype=torch.bfloat16, device="cuda")
            f = lambda: ops.gemm(a, b, c, m, n, k, k, b.shape[1], n, a_layout=0, b_layout=lb, variant=v)
            for _ in range(20): f()
            res.append((k, timeit(f, 20) * 1e3))
        (k0, t0), (k1, t1) = res[2], res[4]
        slope = (t1 - t0) / ((k1 - k0) / 64)
        print(f"M{m} N{n} L0{lb} v{v}: " + " ".join(f"K{k}:{t:6.1f}" for k, t in res) +
              f" | us/K-tile {slope:.2f}, fixed {t0 - slope * k0 / 64:.1f} us", flush=True)
