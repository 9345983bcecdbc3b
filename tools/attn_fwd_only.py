"""A few forward-attention launches at one shape, for counter passes.  usage: python tools/attn_fwd_only.py [N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 23041
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, H, hd = 2, 5, 205
hdp = ops.rup(hd, 32)
qkv = torch.zeros(B, N, 3, H, hdp, device="cuda"); qkv[..., :hd] = torch.randn(B, N, 3, H, hd, device="cuda")
qkv = qkv.to(torch.bfloat16)
o = torch.empty(B, N, H, hdp, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
for _ in range(reps):
    ops.attn_fwd(qkv, o, lse, B, N, H, hd, hdp, hd ** -0.5)
torch.cuda.synchronize()
