"""Soak run (development aid): the bench's Stage-I (or, with a third argument "stage2", Stage-II) model trained for many steps on
one fixed synthetic batch - the loss must stay finite and fall.  usage: python tools/soak.py [steps] [frames] [stage1|stage2]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vita_amd.train import FlatAdam, ReconMAE, stage1_train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
wl = sys.argv[3] if len(sys.argv) > 3 else "stage1"
torch.manual_seed(1)
if wl == "stage1":
    imgs = torch.rand(2, 9, T, 128, 128, device="cuda")
    model = ReconMAE(img_shape=[9, T, 128, 128], use_both_axes=True, **bench.REF_CFG).cuda()
    opt = FlatAdam(model.parameters(), lr=1e-4)
    step = lambda: stage1_train_step(model, opt, imgs)
else:
    from types import SimpleNamespace
    from vita_amd.train import MultimodalSimCLR, stage2_train_step
    B = 32
    imgs = torch.rand(B, 9, T, 128, 128, device="cuda")
    imaging = dict(bench.REF_CFG, mask_ratio=0.5, mask_loss=True, grad_checkpointing=True, projection_dim=128)
    tab = SimpleNamespace(grad_checkpointing=True, embedding_dim=1025, encoder_num_heads=5, encoder_mlp_ratio=4.0,
                          encoder_num_layers=2, projection_dim=128)
    feats = {"numerical": [f"n{i}" for i in range(bench.TAB_NUM)], "single_categorical": [f"s{i}" for i in range(bench.TAB_SINGLE)],
             "multi_categorical": {f"m{i}": [n] for i, n in enumerate(bench.TAB_MULTI)}}
    model = MultimodalSimCLR([9, T, 128, 128], imaging, tab, feats, temperature=0.1, lambda_0=0.5).cuda()
    opt = FlatAdam(model.parameters(), lr=3e-4, weight_decay=1e-4)
    tabs = torch.cat([torch.randn(B, bench.TAB_NUM, device="cuda"), torch.randint(0, 2, (B, bench.TAB_SINGLE), device="cuda").float(),
                      torch.stack([torch.randint(0, n, (B,), device="cuda") for n in bench.TAB_MULTI], 1).float()], 1)
    step = lambda: stage2_train_step(model, opt, imgs, tabs)[0]
losses = []
t0 = time.time()
for i in range(steps):
    loss = step()
    if i % 10 == 0 or i == steps - 1:
        losses.append(float(loss))
        print(f"step {i:4d} loss {losses[-1]:.6f}  ({time.time() - t0:.1f} s)", flush=True)
        assert losses[-1] == losses[-1] and abs(losses[-1]) < 1e3, "loss is not finite"
assert all(torch.isfinite(p).all() for p in model.parameters())
assert losses[-1] < 0.9 * losses[0], (losses[0], losses[-1])
print("soak ok: loss", losses[0], "->", losses[-1])
