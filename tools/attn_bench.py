"""Micro-benchmark of the attention kernels at bench shapes (development aid).
usage: python tools/attn_bench.py [attn|ln|gemm|all] [handoff=0|1]"""
import os
import sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vita_amd import ops

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def attn(B, N, H=5, hd=205, handoff=True):
    hdp = ops.rup(hd, 32)
    qkv = torch.zeros(B, N, 3, H, hdp, device="cuda"); qkv[..., :hd] = torch.randn(B, N, 3, H, hd, device="cuda")
    qkv = qkv.to(torch.bfloat16)
    o = torch.empty(B, N, H, hdp, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
    do = torch.zeros(B, N, H, hdp, device="cuda"); do[..., :hd] = torch.randn(B, N, H, hd, device="cuda"); do = do.to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    sc = hd ** -0.5
    tf = timeit(lambda: ops.attn_fwd(qkv, o, lse, B, N, H, hd, hdp, sc))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, N, H, hd, hdp, sc, handoff=handoff))
    fl = 4.0 * B * H * N * N * hd
    print(f"attn B{B} N{N} handoff={handoff}: fwd {tf:.3f} ms {fl/tf/1e9:.0f} TF/s | bwd {tb:.3f} ms {2.5*fl/tb/1e9:.0f} TF/s", flush=True)

def ln(rows, d=1025):
    ld = ops.rup(d, 64)
    x = torch.randn(rows, ld, device="cuda"); g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, ld, g, b, rows, d, ld)
    dy = torch.randn(rows, ld, device="cuda").to(torch.bfloat16)
    tf = timeit(lambda: ops.layernorm_fwd(x, ld, g, b, rows, d, ld), 10)
    tb = timeit(lambda: ops.layernorm_bwd(dy, ld, x, ld, g, mean, rstd, rows, d, ld, dx_in=x, want_bf16=True, ld_dxb=ld), 10)
    print(f"ln rows {rows}: fwd {tf*1e3:.1f} us ({rows*ld*6/tf/1e6:.0f} GB/s) bwd {tb*1e3:.1f} us ({rows*ld*16/tb/1e6:.0f} GB/s)", flush=True)

def colsum(rows, cols):
    x = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: ops.colsum(x, cols, rows, cols), 10)
    print(f"colsum {rows}x{cols}: {t*1e3:.1f} us ({rows*cols*2/t/1e6:.0f} GB/s)", flush=True)

def scatter_mse():
    B, L, keep, D = 2, 23040, 6912, 1025
    ld = ops.rup(D, 64)
    dout = torch.randn(B, L + 1, ld, device="cuda")
    ids = torch.stack([torch.randperm(L, device="cuda") for _ in range(B)])
    ids = torch.cat([torch.zeros(B, 1, dtype=torch.int64, device="cuda"), ids + 1], 1)      # decoder un-shuffle with the cls row
    t = timeit(lambda: ops.row_scatter(dout, ld, L + 1, ids, L + 1, keep + 1, B, D, ld, want_dfill=True), 10)
    print(f"row_scatter {B}x{L+1}x{D}: {t*1e3:.1f} us", flush=True)
    img = torch.rand(B, 9, 50, 128, 128, device="cuda")
    z = torch.randn(B * (L + 1), 328, device="cuda")
    loss = torch.zeros(1, device="cuda"); dz = torch.empty(B * (L + 1), 328, dtype=torch.bfloat16, device="cuda")
    t = timeit(lambda: ops.sigmoid_mse_patch(z, 328, img, (5, 8, 8), None, 1.0 / (B * L * 320), loss, dz=dz, ld_dz=328), 10)
    print(f"sigmoid_mse_patch: {t*1e3:.1f} us", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    hand = [bool(int(sys.argv[2]))] if len(sys.argv) > 2 else [True, False]
    if which in ("all", "ln"):
        ln(13826); ln(46082)
        for r in (13826, 46082):
            for c in (1088, 3360, 4160): colsum(r, c)
    if which in ("all", "row"):
        scatter_mse()
    if which in ("all", "attn"):
        for h in hand:
            attn(2, 6913, handoff=h); attn(2, 23041, handoff=h)
