"""LayerNorm / column-sum micro-benchmark at the Stage-I shapes (development aid; run on the GPU box)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vita_amd import ops


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


D, Dp = 1025, 1088
for rows in (13826, 46082):
    x = torch.randn(rows, Dp, device="cuda"); x[:, D:] = 0
    w = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    t = timeit(lambda: ops.layernorm_fwd(x, Dp, w, b, rows, D, Dp))
    y, mean, rstd = ops.layernorm_fwd(x, Dp, w, b, rows, D, Dp)
    dy = torch.randn(rows, Dp, device="cuda").to(torch.bfloat16)
    din = torch.randn(rows, Dp, device="cuda")
    tb = timeit(lambda: ops.layernorm_bwd(dy, Dp, x, Dp, w, mean, rstd, rows, D, Dp, dx_in=din, want_bf16=True, ld_dxb=Dp))
    print(f"ln rows {rows}: fwd {t:.1f} us ({rows*D*6/t/1e3:.0f} GB/s) bwd {tb:.1f} us ({rows*D*16/tb/1e3:.0f} GB/s)", flush=True)
for rows in (13826, 46082):
    for cols, ld in ((1025, 1088), (3360, 3360), (4100, 4160)):
        xb = torch.randn(rows, ld, device="cuda").to(torch.bfloat16)
        out = torch.zeros(cols, device="cuda")
        t = timeit(lambda: ops.colsum(xb, ld, rows, cols, out=out, accumulate=True))
        ref = xb[:, :cols].float().sum(0)
        got = ops.colsum(xb, ld, rows, cols)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        print(f"colsum {rows}x{cols}: {t:.1f} us ({rows*cols*2/t/1e3:.0f} GB/s) err {err:.1e}", flush=True)
    xf = torch.randn(rows, Dp, device="cuda")
    t = timeit(lambda: ops.colsum(xf, Dp, rows, D))
    print(f"colsum f32 {rows}x{D}: {t:.1f} us ({rows*D*4/t/1e3:.0f} GB/s)", flush=True)
