"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host mirror keeps
the reference's state_dict keys, and the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vita_hip.h")).read()
    declared = set(re.findall(r"\b(vita_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    so = os.path.join(ROOT, "vita_amd", "libvita_hip.so")
    assert os.path.exists(so), "build the extension first (__graft_entry__.build())"
    lib = ctypes.CDLL(so)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    from vita_amd import _lib
    assert declared == set(_lib.SIGNATURES) | {"vita_last_error"}
    assert _lib.lib().vita_version() >= 100


def test_state_dict_keys_match_reference_contract():
    from oracle import vita_oracle as O
    from vita_amd.train import ReconMAE
    cfg = dict(patch_size=(5, 8, 8), enc_embed_dim=17, enc_depth=2, enc_num_heads=1, dec_embed_dim=17, dec_depth=1,
               dec_num_heads=1, mlp_ratio=4.0, mask_ratio=0.7, mask_type="random", use_enc_pe=True,
               patch_embed_cls="PatchEmbed", patch_in_channels=1, some_unrelated_hparam=3)
    m = ReconMAE(img_shape=[9, 10, 16, 16], use_both_axes=True, **cfg)
    p = O.init_stage1_params(img_shape=(9, 10, 16, 16), patch_size=(5, 8, 8), enc_dim=17, enc_depth=2, enc_heads=1,
                             dec_dim=17, dec_depth=1, dec_heads=1)
    sd = m.state_dict()
    assert set(sd) == set(p)
    for k in p:
        assert tuple(sd[k].shape) == tuple(p[k].shape), k
    assert not m.encoder_imaging.enc_pos_embed.requires_grad and not m.decoder_imaging.dec_pos_embed.requires_grad
    # the frozen tables equal the reference-pinned oracle tables
    assert torch.equal(sd["encoder_imaging.enc_pos_embed"], p["encoder_imaging.enc_pos_embed"])
    assert m.encoder_imaging.patch_embed.grid_size == (9, 2, 2, 2) and int(m.encoder_imaging.patch_embed.num_patches) == 72
    assert int(m.encoder_imaging.patch_p_num) == 320 and m.encoder_imaging.using_masker


def test_no_cpu_fallback():
    from vita_amd.losses import CLIPLoss
    from vita_amd.nn import Block
    with pytest.raises(RuntimeError):
        Block(16, 2, 4.0, qkv_bias=True)(torch.zeros(1, 4, 16))
    with pytest.raises(RuntimeError):
        CLIPLoss(0.1)(torch.zeros(2, 4), torch.zeros(2, 4))
    with pytest.raises(NotImplementedError):
        Block(16, 2, 4.0, drop=0.1)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vita_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f


def test_stage2_state_dict_keys_and_head_formulation():
    """Stage-II mirror keeps the reference's module tree (models/MultimodalSimCLR.py:32-64, networks/tabular_encoders.py:
    20-50); the oracle's projection head equals the nn.Sequential(Linear, BatchNorm1d, ReLU, Linear) it restates."""
    from types import SimpleNamespace
    from oracle import vita_oracle as O
    from vita_amd.train import MultimodalSimCLR
    feats = {"numerical": ["a", "b"], "single_categorical": ["c"], "multi_categorical": {"m0": [3], "m1": [5]}}
    imaging = dict(patch_size=(5, 8, 8), enc_embed_dim=17, enc_depth=1, enc_num_heads=1, mlp_ratio=4.0, mask_ratio=0.5,
                   mask_type="random", use_enc_pe=True, patch_embed_cls="PatchEmbed", patch_in_channels=1,
                   grad_checkpointing=True, projection_dim=8)
    tab = SimpleNamespace(grad_checkpointing=True, embedding_dim=17, encoder_num_heads=1, encoder_mlp_ratio=4.0,
                          encoder_num_layers=1, projection_dim=8)
    m = MultimodalSimCLR([9, 10, 16, 16], imaging, tab, feats)
    p = O.init_stage2_params(img_shape=(9, 10, 16, 16), patch_size=(5, 8, 8), enc_dim=17, enc_depth=1, enc_heads=1,
                             tab_depth=1, n_num=2, n_single=1, multi_classes=[3, 5], proj_dim=8)
    sd = m.state_dict()
    buffers = {k for k in sd if "running_" in k or "num_batches_tracked" in k}
    assert set(sd) - buffers == set(p)
    for k in p:
        assert tuple(sd[k].shape) == tuple(p[k].shape), k
    assert not m.encoder_tabular.enc_pos_embed.requires_grad
    assert torch.equal(sd["encoder_tabular.enc_pos_embed"], p["encoder_tabular.enc_pos_embed"])
    for k in ("encoder_tabular.numerical_fe.1.weight", "encoder_tabular.single_c_fe.0.weight",
              "encoder_tabular.multi_c_fe.1.weight", "encoder_tabular.cls_token", "encoder_tabular.network.0.attn.qkv.weight",
              "encoder_tabular.norm.weight", "projector_imaging.layers.0.weight", "projector_tabular.layers.3.bias"):
        assert k in sd, k
    assert tuple(sd["encoder_tabular.multi_c_fe.1.weight"].shape) == (5, 17)          # nn.Embedding(n_classes, D)
    head = torch.nn.Sequential(torch.nn.Linear(17, 17, bias=False), torch.nn.BatchNorm1d(17), torch.nn.ReLU(),
                               torch.nn.Linear(17, 8))
    head.load_state_dict({k[len("projector_tabular.layers."):]: v for k, v in p.items()
                          if k.startswith("projector_tabular.layers.")}, strict=False)
    x = torch.randn(6, 17, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(head.train()(x), O.projection_head(x, p, "projector_tabular."), atol=1e-5)


def test_lr_schedules_match_their_libraries():
    """warmup-cosine == transformers.get_cosine_schedule_with_warmup, cosine == torch CosineAnnealingLR (both installed)."""
    from vita_amd.train import cosine_annealing_lr, warmup_cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    try:
        from transformers import get_cosine_schedule_with_warmup
    except Exception:
        get_cosine_schedule_with_warmup = None
    if get_cosine_schedule_with_warmup is not None:
        opt = torch.optim.Adam([p], lr=3e-3)
        sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=10, num_training_steps=200)
        for ep in range(205):
            assert abs(opt.param_groups[0]["lr"] - warmup_cosine_lr(3e-3, ep, 10, 200)) < 1e-12, ep
            opt.step(); sch.step()
    opt = torch.optim.Adam([p], lr=3e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50, eta_min=0)
    for ep in range(50):
        assert abs(opt.param_groups[0]["lr"] - cosine_annealing_lr(3e-3, ep, 50)) < 1e-9, ep
        opt.step(); sch.step()


def test_frame_indices_and_slice_window():
    """Host side of the input-pipeline contract (datasets/datasets.py:162-165, 419-424)."""
    import numpy as np
    from oracle import vita_oracle as O
    from vita_amd.data import frame_indices, sax_slice_window
    assert frame_indices(50, 50, 50) == list(range(50)) and frame_indices(50, 50, None) == list(range(50))
    for keep in (5, 25, 2, 7):
        step = 49 / (keep - 1)
        assert frame_indices(50, 50, keep) == [round(i * step) for i in range(keep)]
    assert frame_indices(50, 50, 25)[:4] == [0, 2, 4, 6]          # 49/24 * 3 = 6.125; round half to even elsewhere
    assert frame_indices(50, 25, 5, t_downsampling_ratio=2) == [0, 12, 24, 36, 48]
    with pytest.raises(IndexError):
        frame_indices(20, 50, 5)                                   # time_frame larger than the stored cine
    rng = np.random.RandomState(0)
    for first in (0, 3, 9):
        seg = np.zeros((8, 8, 12, 4), dtype=np.int32)
        seg[2:4, 2:4, first:, 0] = 1
        seg[..., 1:] = rng.randint(0, 4, seg[..., 1:].shape)
        assert sax_slice_window(torch.from_numpy(seg), 6) == O.sax_slice_window(seg, 6)


def test_load_pretrained_weights_prefix_filter(tmp_path):
    """initializers/initialize_model.py:106-139: key-prefix filter, strict=False, freezing."""
    from vita_amd.data import filter_state_dict, load_pretrained_weights
    from vita_amd.train import ReconMAE
    cfg = dict(patch_size=(5, 8, 8), enc_embed_dim=17, enc_depth=1, enc_num_heads=1, dec_embed_dim=17, dec_depth=1,
               dec_num_heads=1, mlp_ratio=4.0, mask_ratio=0.7, mask_type="random", use_enc_pe=True,
               patch_embed_cls="PatchEmbed", patch_in_channels=1)
    torch.manual_seed(0)
    src = ReconMAE(img_shape=[9, 10, 16, 16], use_both_axes=True, **cfg)
    torch.manual_seed(1)
    dst = ReconMAE(img_shape=[9, 5, 16, 16], use_both_axes=True, **cfg)      # other T: pos-embed shapes differ
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": src.state_dict(), "epoch": 3}, path)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    res = load_pretrained_weights(dst, path, freeze_encoder=True, loading_parts=["encoder_imaging"],
                                  exclude_parts=["encoder_imaging.enc_pos_embed"])
    assert not res.unexpected_keys and any(k.startswith("decoder_imaging") for k in res.missing_keys)
    after, ref = dst.state_dict(), src.state_dict()
    for k in after:
        if k.startswith("encoder_imaging") and k != "encoder_imaging.enc_pos_embed":
            assert torch.equal(after[k], ref[k]), k
        else:
            assert torch.equal(after[k], before[k]), k
    for n, p in dst.named_parameters():
        assert p.requires_grad == (not n.startswith("encoder_imaging") and not n.endswith("pos_embed")), n
    kept = filter_state_dict(ref, ["decoder_imaging"], ["decoder_imaging.dec_pos_embed"])
    assert kept and all(k.startswith("decoder_imaging") and "dec_pos_embed" not in k for k in kept)
    assert filter_state_dict(ref) == dict(ref)


def test_bench_traffic_lookup_knows_the_profiled_kernels():
    """roofline.traffic comes from the newest committed PMC summary: every attention kernel in it must belong to one of
    bench.py's launch groups (a renamed kernel otherwise drops out of the sum silently), and the dominant group's
    traffic must at least cover its algorithmic bytes."""
    import glob
    import importlib.util
    import json
    import subprocess
    spec = importlib.util.spec_from_file_location("vita_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True)
    t50 = [f for f in newest if json.load(open(f)).get("frames") == 50 and json.load(open(f)).get("per_gpu_batch") == 2]
    assert t50, "no T=50 traffic summary under profiles/"
    kernels = json.load(open(t50[0]))["kernels"]
    grouped = [n for names in bench.GROUP_KERNELS.values() for n in names]
    for k in kernels:
        if "attn_" in k:
            assert any(k.startswith(n) or (" " + n) in k for n in grouped), k
    # every grouped kernel name exists in the library (a stale name in GROUP_KERNELS is dead weight, a typo is a bug)
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "vita_amd", "libvita_hip.so")],
                          capture_output=True, text=True).stdout
    for n in grouped:
        assert n in syms, n
    traffic, src = bench.pmc_traffic("attn_bwd_kernels", 50, 2, 8)
    assert traffic is not None and src == os.path.basename(t50[0])
    # Q, K, V, O, dO in and dQ, dK, dV out of the smaller (encoder) launch: B=2, N=6913, H=5, hdp=224, bf16
    assert traffic > 8 * 2 * 6913 * 5 * 224 * 2


def test_product_library_is_not_a_diagnostic_build():
    """Timing-only dissection switches (ATT_DBG, G2_DBG) and phase stamps (ATT_STAMP) compile marker symbols in; the library
    the product loads must carry none (vita_amd/_lib.py warns about a library that does)."""
    lib = ctypes.CDLL(os.path.join(ROOT, "vita_amd", "libvita_hip.so"))
    assert not hasattr(lib, "vita_debug_build_attn") and not hasattr(lib, "vita_debug_build_gemm")
