"""GEMM micro-benchmark at the Stage-I shapes: old 128x128 kernel vs the 256-row LDS-DMA kernel (development aid)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vita_amd import ops


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name, m, n, k, la, lb, variants, out_f32=False, epi=0, split_k=1):
    a = torch.randn((m, k) if la == 0 else (k, m), device="cuda").to(torch.bfloat16)
    b = torch.randn((n, k) if lb == 0 else (k, n), device="cuda").to(torch.bfloat16)
    c = torch.zeros(m, n, dtype=torch.float32 if (out_f32 or split_k > 1) else torch.bfloat16, device="cuda")
    kw = {}
    if epi == ops.EPI_GELU: kw = dict(epilogue=epi, aux_out=torch.empty_like(c), bias=torch.zeros(n, device="cuda"))
    if epi == ops.EPI_RESIDUAL: kw = dict(epilogue=epi, aux_in=torch.randn(m, n, device="cuda"), ld_aux=n, bias=torch.zeros(n, device="cuda"))
    if epi == ops.EPI_GELU_BWD: kw = dict(epilogue=epi, aux_in=torch.randn(m, n, device="cuda").to(torch.bfloat16), ld_aux=n)
    res = []
    # the first measurements on freshly allocated operands read 5-20 % slow (page tables / clocks still warming up), which
    # used to bias the first column (v0): run the first variant untimed until that has passed
    for _ in range(30):
        ops.gemm(a, b, c, m, n, k, a.shape[1], b.shape[1], n, a_layout=la, b_layout=lb, variant=variants[0], split_k=split_k, **kw)
    for v in variants:
        t = timeit(lambda: ops.gemm(a, b, c, m, n, k, a.shape[1], b.shape[1], n, a_layout=la, b_layout=lb, variant=v,
                                    split_k=split_k, **kw))
        res.append(f"v{v}: {t*1e3:7.1f} us {2.0*m*n*k/t/1e9:5.0f} TF")
    print(f"{name:8s} {m}x{n}x{k} L{la}{lb} sk{split_k}: " + " | ".join(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sq":      # main-loop ceiling of each operand layout on large squares
        for n in (4096, 8192):
            for la, lb in ((0, 0), (0, 1), (1, 1)):
                run(f"sq{n}", n, n, n, la, lb, [13, 14, 16])
        sys.exit(0)
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 13826
    V = [0, 8, 10, 12, 13, 14, 16]
    V0 = V + [108, 110, 112]
    if os.environ.get("GEMM_BENCH_V"):                   # e.g. GEMM_BENCH_V=0: the dispatcher's own choice only
        V = V0 = [int(v) for v in os.environ["GEMM_BENCH_V"].split(",")]
    run("qkv", M, 3360, 1088, 0, 0, V0)
    run("proj", M, 1088, 1120, 0, 0, V0, out_f32=True, epi=ops.EPI_RESIDUAL)
    run("fc1", M, 4160, 1088, 0, 0, V0, epi=ops.EPI_GELU)
    run("fc2", M, 1088, 4160, 0, 0, V0, out_f32=True, epi=ops.EPI_RESIDUAL)
    run("du00", M, 4160, 1088, 0, 0, V0, epi=ops.EPI_GELU_BWD)
    run("dln2_00", M, 1088, 4160, 0, 0, V0)
    run("do00", M, 1120, 1088, 0, 0, V0)
    run("dln1_00", M, 1088, 3360, 0, 0, V0)
    run("du", M, 4160, 1088, 0, 1, V, epi=ops.EPI_GELU_BWD)
    run("dln2", M, 1088, 4160, 0, 1, V)
    run("do", M, 1120, 1088, 0, 1, V)
    run("dln1", M, 1088, 3360, 0, 1, V)
    run("fc2w", 1088, 4160, M, 1, 1, V, split_k=3)
    run("fc1w", 4160, 1088, M, 1, 1, V, split_k=3)
    run("projw", 1088, 1120, M, 1, 1, V, split_k=8)
    run("qkvw", 3360, 1088, M, 1, 1, V, split_k=3)
    run("sq4k", 4096, 4096, 4096, 0, 0, V0)
    run("sq8k", 8192, 8192, 8192, 0, 0, V0)
