"""Forward / backward attention timing at the bench shapes (development aid).  usage: python tools/attn_fwd_time.py [fwd|bwd|both]
Select the kernel family with VITA_ATTN_MFMA32=1 (32x32x16) / unset (16x16x32); one process per setting (read once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from attn_bench import timeit
from vita_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for B, N in ((2, 6913), (2, 23041)):
    H, hd = 5, 205
    hdp = ops.rup(hd, 32)
    qkv = torch.zeros(B, N, 3, H, hdp, device="cuda"); qkv[..., :hd] = torch.randn(B, N, 3, H, hd, device="cuda")
    qkv = qkv.to(torch.bfloat16)
    o = torch.empty(B, N, H, hdp, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
    sc = hd ** -0.5
    fl = 4.0 * B * H * N * N * hd
    for rep in range(2):
        msg = f"N{N}"
        if which in ("fwd", "both"):
            tf = timeit(lambda: ops.attn_fwd(qkv, o, lse, B, N, H, hd, hdp, sc), 10)
            msg += f" fwd {tf:.3f} ms {fl/tf/1e9:.0f} TF/s"
        if which in ("bwd", "both"):
            do = torch.zeros(B, N, H, hdp, device="cuda"); do[..., :hd] = torch.randn(B, N, H, hd, device="cuda"); do = do.to(torch.bfloat16)
            dqkv = torch.empty_like(qkv)
            ops.attn_fwd(qkv, o, lse, B, N, H, hd, hdp, sc)
            tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, N, H, hd, hdp, sc, handoff=True), 5)
            msg += f" bwd {tb:.3f} ms {2*fl/tb/1e9:.0f} TF/s (8BHN2d)"
        print(msg, flush=True)
