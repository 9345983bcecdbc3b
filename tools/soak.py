"""Soak run (development aid): the bench's Stage-I model trained for many steps on one fixed synthetic batch - the loss must
stay finite and fall.  usage: python tools/soak.py [steps] [frames]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vita_amd.train import FlatAdam, ReconMAE, stage1_train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.manual_seed(1)
imgs = torch.rand(2, 9, T, 128, 128, device="cuda")
model = ReconMAE(img_shape=[9, T, 128, 128], use_both_axes=True, **bench.REF_CFG).cuda()
opt = FlatAdam(model.parameters(), lr=1e-4)
losses = []
t0 = time.time()
for i in range(steps):
    loss = stage1_train_step(model, opt, imgs)
    if i % 10 == 0 or i == steps - 1:
        losses.append(float(loss))
        print(f"step {i:4d} loss {losses[-1]:.6f}  ({time.time() - t0:.1f} s)", flush=True)
        assert losses[-1] == losses[-1] and abs(losses[-1]) < 1e3, "loss is not finite"
assert all(torch.isfinite(p).all() for p in model.parameters())
assert losses[-1] < 0.9 * losses[0], (losses[0], losses[-1])
print("soak ok: loss", losses[0], "->", losses[-1])
