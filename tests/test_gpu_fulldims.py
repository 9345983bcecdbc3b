"""End-to-end GPU parity at the REAL model dims (BASELINE.json cfg2 / cfg5: D = 1025, encoder 6 / decoder 2 blocks,
5 heads of 205, hidden 4100; configs/imaging_model/pretraining_reconstruction_mae.yaml:39-47) against the CPU oracle.
One subject of (9, 5, 128, 128) keeps the oracle's materialised attention at N = 2305 (about a second of CPU time):
L = 2304 tokens, 691 kept, N_dec = 2305.  Loss and EVERY gradient tensor are held to north_star's rtol 2e-2
(relative L2 per tensor); the measured worst case is 0.8e-2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vita_oracle as O  # noqa: E402

CFG = dict(patch_size=(5, 8, 8), enc_embed_dim=1025, enc_depth=6, enc_num_heads=5, dec_embed_dim=1025, dec_depth=2,
           dec_num_heads=5, mlp_ratio=4.0, mask_ratio=0.7, mask_type="random", use_enc_pe=True, circular_pe=False,
           patch_embed_cls="PatchEmbed", patch_in_channels=1, mask_loss=False)
IMG = (9, 5, 128, 128)
L = 2304


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def test_stage1_step_full_dims_vs_oracle():
    """networks/imaging_encoders.py:79-116 + networks/imaging_decoders.py:273-305 + MSE + Adam, one Stage-I step."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
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
    p_ref = {k: v.clone() for k, v in p.items()}
    ref, grads = O.stage1_train_step(imgs, p_ref, noise, {}, lr=1e-4, patch_size=(5, 8, 8), enc_depth=6, enc_heads=5,
                                     dec_depth=2, dec_heads=5, mask_ratio=0.7)
    assert abs(loss.item() - ref.item()) < 2e-2 * abs(ref.item()), (loss.item(), ref.item())
    named = dict(model.named_parameters())
    assert set(grads) == {k for k, v in named.items() if v.requires_grad}
    for k, g in grads.items():
        e = rel(named[k].grad, g)
        assert e < 2e-2, (k, e)
    # The same step against the oracle with its bf16 rounding points on (O.ROUND_BF16: operands of every GEMM / attention
    # product rounded where the kernels store bf16).  What is left is accumulation order and the few places the kernels
    # keep more precision, so every gradient tensor must now agree to 5e-3 (measured: 0.7-0.8e-3, cls_token 2.4e-3) -
    # an indexing / masking / padding bug that hides inside 2e-2 of fp32-vs-bf16 rounding cannot hide here.
    grads_fp32 = {k: named[k].grad.detach().clone() for k in grads}
    O.ROUND_BF16 = True
    try:
        ref_r, grads_r = O.stage1_train_step(imgs, {k: v.clone() for k, v in p.items()}, noise, {}, lr=1e-4, patch_size=(5, 8, 8),
                                             enc_depth=6, enc_heads=5, dec_depth=2, dec_heads=5, mask_ratio=0.7)
    finally:
        O.ROUND_BF16 = False
    assert abs(loss.item() - ref_r.item()) < 1e-3 * abs(ref_r.item()), (loss.item(), ref_r.item())
    worst = max((rel(grads_fp32[k], g), k) for k, g in grads_r.items())
    assert worst[0] < 5e-3, worst
    # Adam moved every weight by at most lr, in the oracle's direction
    for k in ("encoder_imaging.encoder.0.attn.qkv.weight", "decoder_imaging.decoder.network.1.mlp.fc2.weight",
              "encoder_imaging.patch_embed.proj.weight", "decoder_imaging.mask_token"):
        upd, upd_ref = named[k].detach().cpu() - p[k], p_ref[k] - p[k]
        assert upd.abs().max().item() <= 1.01e-4
        assert torch.nn.functional.cosine_similarity(upd.flatten(), upd_ref.flatten(), dim=0) > 0.95, k


def test_skip_connection_encoder_full_dims_vs_oracle():
    """forward_with_skip_connection (networks/imaging_encoders.py:118-145) at D = 1025 / depth 6, forward and backward with
    cotangents on the output AND on hidden states 2 and 4 (what SegMAE consumes, models/segmentation_models.py:55-59): a
    block input that feeds two consumers - its gradient is accumulated by autograd before the next Block backward sees it."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
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
    leaves = {k: v.clone().requires_grad_(not k.endswith(O.FROZEN_KEYS)) for k, v in pe.items()}
    yr, hr = O.encoder_forward_with_skip(imgs, leaves, patch_size=(5, 8, 8), num_heads=5, depth=6)
    ((yr * cy).sum() + (hr[2] * c2).sum() + (hr[4] * c4).sum()).backward()
    assert y.shape == yr.shape == (1, L + 1, 1025) and len(hidden) == 6
    assert rel(y, yr) < 2e-2
    for h, r in zip(hidden, hr):
        assert h.shape == r.shape == (1, L, 1025) and rel(h, r) < 2e-2
    named = dict(enc.named_parameters())
    for k, v in leaves.items():
        if v.grad is not None:
            assert rel(named[k].grad, v.grad) < 2e-2, (k, rel(named[k].grad, v.grad))


def test_stale_bf16_twin_is_ignored():
    """vita_amd.nn.with_bf16_twin / bf16_twin: a bf16 copy attached to an fp32 gradient is used only while the tensor's
    version is the one recorded with it - an in-place accumulation (autograd's input buffer adding a second consumer's
    gradient) invalidates it."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from vita_amd import nn as vnn
    g = torch.randn(4, 8, 64, device="cuda")
    gb = g.view(32, 64).to(torch.bfloat16)
    vnn.with_bf16_twin(g, gb)
    assert vnn.bf16_twin(g, 32, 64) is gb
    g.add_(1.0)
    assert vnn.bf16_twin(g, 32, 64) is None
