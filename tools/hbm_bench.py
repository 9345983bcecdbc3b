"""HBM-bound kernels of the path at the Stage-I T=50 shapes (achieved GB/s for K1 / K3 / K4 and the
loss / optimizer kernels).  Development aid; run on the GPU box.  Times are HIP-event times of the ops.* wrapper, so the
output allocation (and the zero fill of the scatter target) is inside them."""
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
    return e0.elapsed_time(e1) / n * 1e3      # us


def show(name, us, nbytes):
    print(f"{name:44s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e3:7.0f} GB/s  ({nbytes / us / 1e3 / 8000:.0%} of 8 TB/s)", flush=True)


B, S, T, HW, D = 2, 9, 50, 128, 1025
L, keep, P = 23040, 6912, 320
Dp, Pp = ops.rup(D, 64), ops.rup(P, 8)
torch.manual_seed(1)
img = torch.rand(B, S, T, HW, HW, device="cuda")
noise = torch.rand(B, L, device="cuda")
# K3: argsort + inverse permutation + mask
t = timeit(lambda: ops.argsort_mask(noise, keep))
show("K3 argsort_mask 2 x 23040", t, B * L * (4 + 8 + 8 + 4))
ids_shuffle, ids_restore, mask = ops.argsort_mask(noise, keep)
ids_keep = ids_shuffle[:, :keep].contiguous()
# K1: kept patches -> bf16 patch matrix (the tokenizer GEMM runs on it), then bias-free assemble (+pos, cls row)
t = timeit(lambda: ops.im2col(img, (5, 8, 8), ids_keep, keep, Pp))
show("K1 im2col of the kept 30 % (fp32 -> bf16)", t, B * keep * P * (4 + 2))
y = torch.randn(B * keep, Dp, device="cuda")
cls = torch.zeros(D, device="cuda"); pos = torch.randn(L + 1, D, device="cuda")
t = timeit(lambda: ops.tokens_assemble(y, Dp, cls, pos, ids_keep, B, keep, D, Dp))
show("K1/K2/K5 tokens_assemble (+pos, cls row)", t, B * keep * D * (4 + 4 + 4))
# K4: Masker gather (embed-then-gather form, used by forward_with_skip / tests) and its scatter backward
x = torch.randn(B, L, Dp, device="cuda")
t = timeit(lambda: ops.row_gather(x, Dp, L, ids_keep, keep, keep, B, D, Dp))
show("K4 row_gather 23040 -> 6912 rows", t, B * keep * (2 * D * 4 + 8))
dout = torch.randn(B, keep, Dp, device="cuda")
t = timeit(lambda: ops.row_scatter(dout, Dp, keep, ids_keep, keep, L, B, D, Dp))
show("K4 row_scatter 6912 -> 23040 rows (+ zero fill)", t, B * keep * (2 * D * 4 + 8) + B * L * Dp * 4)
# decoder un-shuffle: gather with mask-token fill + pos-embed, and its backward
xe = torch.randn(B, keep + 1, Dp, device="cuda")
idr = torch.cat([torch.zeros(B, 1, dtype=torch.int64, device="cuda"), ids_restore + 1], 1).contiguous()
fill = torch.randn(D, device="cuda")
t = timeit(lambda: ops.row_gather(xe, Dp, keep + 1, idr, L + 1, L + 1, B, D, Dp, fill=fill, pos=pos))
show("K4/K15 un-shuffle gather + mask token + pos", t, B * ((keep + 1) * D * 4 + (L + 1) * (D * 4 + D * 4 + 8)))
dd = torch.randn(B, L + 1, Dp, device="cuda")
t = timeit(lambda: ops.row_scatter(dd, Dp, L + 1, idr, L + 1, keep + 1, B, D, Dp, want_dfill=True))
show("K4/K15 un-shuffle scatter + mask-token grad", t, B * ((L + 1) * (D * 4 + 8) + (keep + 1) * D * 4))
# K14: sigmoid + patchified target + MSE + dz
z = torch.randn(B * (L + 1), Pp, device="cuda")
loss = torch.zeros(1, device="cuda"); dz = torch.empty(B * (L + 1), Pp, dtype=torch.bfloat16, device="cuda")
t = timeit(lambda: ops.sigmoid_mse_patch(z, Pp, img, (5, 8, 8), None, 1.0 / (B * L * P), loss, dz=dz, ld_dz=Pp))
show("K14 sigmoid + patchify + MSE + dz", t, B * L * P * (4 + 4 + 2))
# K16: Adam over the flat parameter buffer
n = 102_700_000
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda"); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
t = timeit(lambda: ops.adam_step(p, g, m, v, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1.0, 1), 5)
show("K16 adam_step 102.7 M parameters", t, n * 28)
