"""Yardstick only (development aid, never on the product path): what the vendor BLAS behind torch.matmul reaches on the
model's GEMM shapes, to judge how far gemm256_kernel is from a tuned library.  Run under rocprofv3 --kernel-trace to see
which macro-tile each shape gets."""
import sys
import torch


def timeit(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name, m, n, k, la, lb):
    # la/lb as in vita_gemm_desc: 0 = k-contiguous ([rows][k]), 1 = k-strided ([k][rows])
    a = torch.randn((m, k) if la == 0 else (k, m), device="cuda").to(torch.bfloat16)
    b = torch.randn((n, k) if lb == 0 else (k, n), device="cuda").to(torch.bfloat16)
    A = a if la == 0 else a.t()
    Bt = b.t() if lb == 0 else b
    c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    t = timeit(lambda: torch.matmul(A, Bt, out=c))
    print(f"{name:8s} {m}x{n}x{k} L{la}{lb}: {t*1e3:7.1f} us {2.0*m*n*k/t/1e9:5.0f} TF", flush=True)


if __name__ == "__main__":
    for M in (13826, 46082):
        run("qkv", M, 3360, 1088, 0, 0)
        run("proj", M, 1088, 1120, 0, 0)
        run("fc1", M, 4160, 1088, 0, 0)
        run("fc2", M, 1088, 4160, 0, 0)
        run("du", M, 4160, 1088, 0, 1)
        run("dln2", M, 1088, 4160, 0, 1)
        run("do", M, 1120, 1088, 0, 1)
        run("dln1", M, 1088, 3360, 0, 1)
        run("fc2w", 1088, 4160, M, 1, 1)
        run("fc1w", 4160, 1088, M, 1, 1)
        run("projw", 1088, 1120, M, 1, 1)
        run("qkvw", 3360, 1088, M, 1, 1)
    run("sq4k", 4096, 4096, 4096, 0, 0)
    run("sq8k", 8192, 8192, 8192, 0, 0)
