import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_model as T
from oracle import vita_oracle as O
from vita_amd.train import FlatAdam, stage2_train_step
model, p = T._s2_build()
imgs, tabs = T._s2_batch()
opt = FlatAdam(model.parameters(), lr=5e-4, weight_decay=1e-4)
got = []
for i in range(5):
    noise = T.draw_noise(300 + i, 8, 144)
    got.append(stage2_train_step(model, opt, imgs.cuda(), tabs.cuda())[0].item())
print(os.environ.get("VITA_ATTN_MFMA32"), os.environ.get("VITA_ATTN_FWD_NOPIPE"), got)
