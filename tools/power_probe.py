import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vita_amd import ops
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N, hd = 23041, 205
hdp = 224
for (B, H) in ((1, 1), (1, 2), (2, 5)):
    for NN in (N, 128 * 64):
        qkv = (torch.randn(B, NN, 3, H, hdp, device="cuda") * 0.5).to(torch.bfloat16)
        o = torch.empty(B, NN, H, hdp, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, NN, device="cuda")
        do = torch.randn(B, NN, H, hdp, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
        sc = hd ** -0.5
        tf = timeit(lambda: ops.attn_fwd(qkv, o, lse, B, NN, H, hd, hdp, sc))
        tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, NN, H, hd, hdp, sc))
        wg_f = B * H * ((NN + 255) // 256); wg_b = B * H * ((NN + 127) // 128); nt = (NN + 31) // 32
        print(f"B{B} H{H} N{NN}: fwd {tf:.3f} ms ({wg_f} wgs, {tf*1e3/nt:.3f} us/tile-step if one round) | bwd {tb:.3f} ms ({wg_b} dkdv wgs)", flush=True)
