"""Phase-by-phase shader cycles of the forward attention tile loop (diagnostic build libvita_stamp.so, ATT16_STAMP=1).
usage (GPU box): VITA_AMD_LIB=vita_amd/libvita_stamp.so python tools/attn_stamps.py [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vita_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 23041
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
B, H, hd = 2, 5, 205
hdp = ops.rup(hd, 32)
qkv = torch.zeros(B, N, 3, H, hdp, device="cuda"); qkv[..., :hd] = torch.randn(B, N, 3, H, hd, device="cuda")
qkv = qkv.to(torch.bfloat16)
o = torch.empty(B, N, H, hdp, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
for _ in range(3):
    ops.attn_fwd(qkv, o, lse, B, N, H, hd, hdp, hd ** -0.5)
if which == "bwd":
    do = torch.zeros(B, N, H, hdp, device="cuda"); do[..., :hd] = torch.randn(B, N, H, hd, device="cuda"); do = do.to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    for _ in range(2):
        ops.attn_bwd(qkv, o, do, lse, dqkv, B, N, H, hd, hdp, hd ** -0.5, handoff=True)
torch.cuda.synchronize()
nwg = 768
buf = np.zeros((nwg, 8, 8), dtype=np.uint64)
lib = ctypes.CDLL(os.environ["VITA_AMD_LIB"])
rc = lib.vita_attn_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), nwg)
assert rc == 0
names = ["loop-top", "S chain", "softmax", "PV", "dma wait", "barrier"] if which == "fwd" else ["top", "chain (S | dP)", "product+valu (dV | dK)", "-", "dma wait", "barrier"]
ok = buf[:, :, 7] > 0
tiles = buf[:, :, 7][ok].astype(np.float64)
print("workgroups with data:", int(ok.any(1).sum()), "tiles per wg:", tiles.mean())
tot = buf[:, :, 6][ok].astype(np.float64) / tiles
print(f"total cycles per tile: mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f}")
for k, n in enumerate(names):
    v = buf[:, :, k][ok].astype(np.float64) / tiles
    print(f"  {n:28s} mean {v.mean():7.1f}  p10 {np.percentile(v,10):7.1f}  p90 {np.percentile(v,90):7.1f}")
for w in range(8):
    okw = ok[:, w]
    v = [(buf[:, w, k][okw].astype(np.float64) / buf[:, w, 7][okw]).mean() for k in range(6)]
    print(f"  wave {w}: " + " ".join(f"{x:7.1f}" for x in v))
