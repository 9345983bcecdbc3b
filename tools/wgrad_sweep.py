"""Sweep (tile width, split-K) of the weight-gradient GEMMs at the model's shapes against nn._wgrad_plan's pick."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vita_amd import ops
from vita_amd.nn import _wgrad_plan
from gemm_bench import timeit

def sweep(name, m, n, k, row_seg=(0, 0), col_seg=(0, 0), rows_out=None, cols_out=None):
    rows_out = rows_out or m; cols_out = cols_out or n
    a = torch.randn(k, (m + 7) // 8 * 8, device="cuda").to(torch.bfloat16)
    b = torch.randn(k, (n + 7) // 8 * 8, device="cuda").to(torch.bfloat16)
    c = torch.zeros(rows_out, cols_out, device="cuda")
    res = {}
    for nf in (13, 14, 16):
        for sk in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if (k + 63) // 64 // sk < 4: continue
            t = timeit(lambda: ops.gemm(a, b, c, m, n, k, a.shape[1], b.shape[1], cols_out, a_layout=1, b_layout=1, split_k=sk,
                                        variant=nf, row_seg=row_seg, col_seg=col_seg), 5)
            res[(nf, sk)] = t * 1e3
    pick = _wgrad_plan(m, n, k)
    best = min(res, key=res.get)
    top = sorted(res.items(), key=lambda kv: kv[1])[:4]
    print(f"{name:6s} m{m} n{n} k{k}: plan {pick} -> {res.get(pick, float('nan')):.1f} us (forced sk>=2: {res.get((pick[0], max(pick[1], 2)), float('nan')):.1f}) | best "
          + ", ".join(f"{kk}:{v:.1f}" for kk, v in top), flush=True)

if __name__ == "__main__":
    for M in (13826, 46082):
        sweep("fc2w", 1025, 4100, M)
        sweep("fc1w", 4100, 1025, M)
        sweep("projw", 1025, 1120, M, col_seg=(224, 205), cols_out=1025)
        sweep("qkvw", 3360, 1025, M, row_seg=(224, 205), rows_out=3075)
