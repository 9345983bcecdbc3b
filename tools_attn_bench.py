"""Micro-benchmark of the attention / GEMM / LN kernels at bench shapes (development aid)."""
import sys, time
import torch
sys.path.insert(0, ".")
from vita_amd import ops

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def attn(B, N, H=5, hd=205):
    hdp = ops.rup(hd, 32)
    qkv = torch.zeros(B, N, 3, H, hdp, device="cuda"); qkv[..., :hd] = torch.randn(B, N, 3, H, hd, device="cuda")
    qkv = qkv.to(torch.bfloat16)
    o = torch.empty(B, N, H, hdp, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
    do = torch.zeros(B, N, H, hdp, device="cuda"); do[..., :hd] = torch.randn(B, N, H, hd, device="cuda"); do = do.to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    sc = hd ** -0.5
    tf = timeit(lambda: ops.attn_fwd(qkv, o, lse, B, N, H, hd, hdp, sc))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, N, H, hd, hdp, sc))
    fl = 4.0 * B * H * N * N * hd
    print(f"attn B{B} N{N}: fwd {tf:.3f} ms {fl/tf/1e9:.0f} TF/s | bwd {tb:.3f} ms {2.5*fl/tb/1e9:.0f} TF/s", flush=True)

def gemm(m, n, k, la, lb):
    a = torch.randn((m, k) if la == 0 else (k, m), device="cuda").to(torch.bfloat16)
    b = torch.randn((n, k) if lb == 0 else (k, n), device="cuda").to(torch.bfloat16)
    c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    t = timeit(lambda: ops.gemm(a, b, c, m, n, k, a.shape[1], b.shape[1], n, a_layout=la, b_layout=lb), 10)
    print(f"gemm {m}x{n}x{k} L{la}{lb}: {t*1e3:.1f} us {2.0*m*n*k/t/1e9:.0f} TF/s", flush=True)

def ln(rows, d=1025):
    ld = ops.rup(d, 64)
    x = torch.randn(rows, ld, device="cuda"); g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, ld, g, b, rows, d, ld)
    dy = torch.randn(rows, ld, device="cuda").to(torch.bfloat16)
    tf = timeit(lambda: ops.layernorm_fwd(x, ld, g, b, rows, d, ld), 10)
    tb = timeit(lambda: ops.layernorm_bwd(dy, ld, x, ld, g, mean, rstd, rows, d, ld, dx_in=x, want_bf16=True, ld_dxb=ld), 10)
    print(f"ln rows {rows}: fwd {tf*1e3:.1f} us ({rows*ld*6/tf/1e6:.0f} GB/s) bwd {tb*1e3:.1f} us ({rows*ld*16/tb/1e6:.0f} GB/s)", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "ln"):
        ln(6914); ln(46082)
    if which in ("all", "attn"):
        attn(2, 3457); attn(2, 11521)
    if which in ("all", "gemm"):
        for la, lb in ((0, 0), (0, 1), (1, 1)):
            gemm(6912, 4160, 1088, la, lb); gemm(6912, 1088, 4160, la, lb)
        gemm(1088, 4160, 6912, 1, 1); gemm(4096, 4096, 4096, 0, 0); gemm(8192, 8192, 8192, 0, 0)
