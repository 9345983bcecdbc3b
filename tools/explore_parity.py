"""Development aid: per-tensor relative errors of the HIP path against the oracle (plain fp32 and with the oracle's bf16
rounding points) at the real model dims.  usage (GPU box): python tools/explore_parity.py [stage1|skip|stage2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import vita_oracle as O


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


CFG = dict(patch_size=(5, 8, 8), enc_embed_dim=1025, enc_depth=6, enc_num_heads=5, dec_embed_dim=1025, dec_depth=2,
           dec_num_heads=5, mlp_ratio=4.0, mask_ratio=0.7, mask_type="random", use_enc_pe=True, circular_pe=False,
           patch_embed_cls="PatchEmbed", patch_in_channels=1, mask_loss=False)
IMG = (9, 5, 128, 128)
L = 2304


def stage1():
    from vita_amd.train import FlatAdam, ReconMAE, stage1_train_step
    p = O.init_stage1_params(img_shape=IMG, patch_size=(5, 8, 8), enc_dim=1025, enc_depth=6, enc_heads=5, dec_dim=1025,
                             dec_depth=2, dec_heads=5, seed=1)
    model = ReconMAE(img_shape=list(IMG), use_both_axes=True, **CFG)
    model.load_state_dict(p, strict=True)
    model.cuda()
    opt = FlatAdam(model.parameters(), lr=1e-4)
    torch.manual_seed(3)
    imgs = torch.rand(1, *IMG)
    torch.manual_seed(11)
    noise = torch.rand(1, L, device="cuda").cpu().numpy()
    torch.manual_seed(11)
    loss = stage1_train_step(model, opt, imgs.cuda())
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    cfg = dict(patch_size=(5, 8, 8), enc_depth=6, enc_heads=5, dec_depth=2, dec_heads=5, mask_ratio=0.7)
    for rnd in (False, True):
        O.ROUND_BF16 = rnd
        t0 = time.time()
        ref, grads = O.stage1_train_step(imgs, {k: v.clone() for k, v in p.items()}, noise, {}, lr=1e-4, **cfg)
        O.ROUND_BF16 = False
        errs = {k: rel(named[k].grad, g) for k, g in grads.items()}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
        print(f"stage1 round={rnd}: loss {loss.item():.6f} ref {ref.item():.6f} rel {abs(loss.item() - ref.item()) / ref.item():.2e} "
              f"oracle {time.time() - t0:.1f}s  worst grads:")
        for k, e in worst:
            print(f"   {e:.4f} {k}")


def skip():
    from vita_amd.networks import ImagingMaskedEncoder
    p = O.init_stage1_params(img_shape=IMG, patch_size=(5, 8, 8), enc_dim=1025, enc_depth=6, enc_heads=5, dec_dim=1025,
                             dec_depth=0, dec_heads=5, seed=2)
    pe = {k[len("encoder_imaging."):]: v for k, v in p.items() if k.startswith("encoder_imaging.")}
    enc = ImagingMaskedEncoder(img_shape=list(IMG), use_both_axes=True, **dict(CFG, mask_ratio=0.0))
    enc.load_state_dict(pe, strict=True)
    enc.cuda()
    torch.manual_seed(4)
    imgs = torch.rand(1, *IMG)
    y, hidden = enc.forward_with_skip_connection(imgs.cuda())
    cy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    c2 = torch.randn(hidden[2].shape, generator=torch.Generator().manual_seed(2))
    c4 = torch.randn(hidden[4].shape, generator=torch.Generator().manual_seed(3))
    ((y * cy.cuda()).sum() + (hidden[2] * c2.cuda()).sum() + (hidden[4] * c4.cuda()).sum()).backward()
    named = dict(enc.named_parameters())
    for rnd in (False, True):
        O.ROUND_BF16 = rnd
        leaves = {k: v.clone().requires_grad_(not k.endswith(O.FROZEN_KEYS)) for k, v in pe.items()}
        yr, hr = O.encoder_forward_with_skip(imgs, leaves, patch_size=(5, 8, 8), num_heads=5, depth=6)
        ((yr * cy).sum() + (hr[2] * c2).sum() + (hr[4] * c4).sum()).backward()
        O.ROUND_BF16 = False
        errs = {k: rel(named[k].grad, v.grad) for k, v in leaves.items() if v.grad is not None}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
        print(f"skip round={rnd}: y {rel(y, yr):.4f} hidden " + " ".join(f"{rel(h, r):.4f}" for h, r in zip(hidden, hr)))
        for k, e in worst:
            print(f"   {e:.4f} {k}")


def stage2():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import test_gpu_model as T
    from vita_amd.train import FlatAdam, stage2_train_step
    model, p = T._s2_build()
    imgs, tabs = T._s2_batch()
    opt = FlatAdam(model.parameters(), lr=3e-3, weight_decay=1e-4)
    noise = T.draw_noise(21, 8, 144)
    loss, logits, labels = stage2_train_step(model, opt, imgs.cuda(), tabs.cuda())
    named = dict(model.named_parameters())
    for rnd in (False, True):
        O.ROUND_BF16 = rnd
        ref, grads = O.stage2_train_step(imgs, tabs, {k: v.clone() for k, v in p.items()}, noise, {}, lr=3e-3,
                                         weight_decay=1e-4, **T.S2_OCFG)
        O.ROUND_BF16 = False
        gmax = max(g.norm().item() for g in grads.values())
        errs = {k: rel(named[k].grad, g) for k, g in grads.items() if g.norm().item() >= 1e-3 * gmax}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
        print(f"stage2 round={rnd}: loss {loss.item():.6f} ref {ref.item():.6f}; worst grads:")
        for k, e in worst:
            print(f"   {e:.4f} {k}")




def small():
    """Worst per-tensor errors of the small-dims tests (which tensor needs more than 2e-2, if any)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import test_gpu_model as T
    from vita_amd.train import FlatAdam, stage1_train_step
    model, p = T.build()
    torch.manual_seed(3)
    imgs = torch.rand(2, *T.IMG)
    opt = FlatAdam(model.parameters(), lr=1e-4)
    noise = T.draw_noise(55, 2, T.L_TOK)
    stage1_train_step(model, opt, imgs.cuda())
    for rnd in (False, True):
        O.ROUND_BF16 = rnd
        _, grads = O.stage1_train_step(imgs, {k: v.clone() for k, v in p.items()}, noise, {}, lr=1e-4, **T.ocfg())
        O.ROUND_BF16 = False
        named = dict(model.named_parameters())
        errs = sorted(((rel(named[k].grad, g), k) for k, g in grads.items()), reverse=True)[:6]
        print(f"small stage1 round={rnd}:", errs)
    model, p = T._s2_build()
    imgs, tabs = T._s2_batch()
    y = model.encoder_tabular(tabs.cuda())
    cot = torch.randn(8, 13, 129, generator=torch.Generator().manual_seed(9))
    (y * cot.cuda()).sum().backward()
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("encoder_tabular.")}
    (O.tabular_vit_forward(tabs, pr, prefix="encoder_tabular.", num_heads=3, depth=2, n_num=5, n_single=3, n_multi=4) * cot).sum().backward()
    named = dict(model.named_parameters())
    errs = sorted(((rel(named[k].grad, v.grad), k) for k, v in pr.items() if v.grad is not None and named[k].requires_grad), reverse=True)[:6]
    print("small tabular:", errs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["stage1", "skip", "stage2"]
    for w in which:
        globals()[w]()
