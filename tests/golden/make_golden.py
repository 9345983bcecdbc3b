"""Generate golden vectors by importing the reference's torch-only leaf modules.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
Writes small .npz fixtures next to this file.  The fixtures are DATA (inputs + the reference's
outputs); no reference source is copied.  Reference entry points exercised:
  networks/tokenizers.py:11-56          PatchEmbed
  utils/imaging_model_related.py:8-71   sincos_pos_embed
  utils/imaging_model_related.py:103-147 patchify / unpatchify
  utils/imaging_model_related.py:230-286 Masker
  utils/clip_loss.py:6-40               CLIPLoss
  utils/general.py:139-147              image_normalization (definition only; the module imports skimage)
"""
import hashlib
import os
import sys

import numpy as np
import torch

REF = os.environ.get("VITA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from networks.tokenizers import PatchEmbed  # noqa: E402
from utils.clip_loss import CLIPLoss  # noqa: E402
from utils.imaging_model_related import Masker, patchify, sincos_pos_embed, unpatchify  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print(name, {k: getattr(v, "shape", v) for k, v in arrs.items()})


def g1_patch_embed():
    out = {}
    for tag, (B, im, D) in {"a": (2, [9, 10, 16, 16], 17), "b": (1, [9, 5, 32, 32], 129)}.items():
        torch.manual_seed(11)
        pe = PatchEmbed(im, in_channels=1, patch_size=[5, 8, 8], out_channels=D)
        x = torch.rand(B, *im)
        y = pe(x)
        cot = torch.randn_like(y)
        (y * cot).sum().backward()
        out.update({f"{tag}_x": x.numpy(), f"{tag}_w": pe.proj.weight.detach().numpy(),
                    f"{tag}_b": pe.proj.bias.detach().numpy(), f"{tag}_y": y.detach().numpy(),
                    f"{tag}_cot": cot.numpy(), f"{tag}_dw": pe.proj.weight.grad.numpy(),
                    f"{tag}_db": pe.proj.bias.grad.numpy(),
                    f"{tag}_grid": np.array(pe.grid_size), f"{tag}_num_patches": np.array(pe.num_patches)})
    save("g1_patch_embed.npz", **out)


def g2_pos_embed():
    out = {}
    for grid in [(9, 2, 2, 2), (9, 1, 4, 4)]:
        for D in (17, 129, 1025):
            t = sincos_pos_embed(D, grid, cls_token=True, use_both_axes=True)
            out[f"full_{'x'.join(map(str, grid))}_{D}"] = t.numpy()
    rng = np.random.RandomState(0)
    for grid, D in [((9, 5, 16, 16), 1025), ((9, 10, 16, 16), 129), ((9, 10, 16, 16), 1025), ((9, 1, 16, 16), 1025)]:
        t = sincos_pos_embed(D, grid, cls_token=True, use_both_axes=True).numpy()
        rows = np.sort(rng.choice(t.shape[0], 64, replace=False))
        rows[0] = 0
        key = f"{'x'.join(map(str, grid))}_{D}"
        out[f"rows_{key}"] = rows
        out[f"samp_{key}"] = t[rows]
        out[f"sha_{key}"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(t).tobytes()).digest(), dtype=np.uint8)
    t = sincos_pos_embed(128, (6, 10, 16, 16), cls_token=True, use_both_axes=False).numpy()
    rows = np.sort(rng.choice(t.shape[0], 64, replace=False))
    out["rows_noaxes_6x10x16x16_128"] = rows
    out["samp_noaxes_6x10x16x16_128"] = t[rows]
    # F7: D=128 with both axes is not constructible
    try:
        sincos_pos_embed(128, (9, 10, 16, 16), cls_token=True, use_both_axes=True)
        out["d128_both_axes_raises"] = np.array(0)
    except AssertionError:
        out["d128_both_axes_raises"] = np.array(1)
    save("g2_pos_embed.npz", **out)


def g3_masker():
    out = {}
    for seed in (0, 1, 2):
        for (N, L) in [(2, 23040), (2, 11520), (4, 2304), (3, 64)]:
            for ratio in (0.7, 0.5):
                if L >= 11520 and not (ratio == 0.7):
                    continue
                torch.manual_seed(seed)
                noise = torch.rand(N, L)
                torch.manual_seed(seed)
                m = Masker("random", ratio, None)
                x = torch.arange(N * L * 2, dtype=torch.float32).reshape(N, L, 2)
                xm, mask, ids_restore = m(x)
                key = f"s{seed}_{N}x{L}_r{int(ratio * 10)}"
                out[f"noise_{key}"] = noise.numpy()
                out[f"mask_{key}"] = mask.numpy().astype(np.uint8)
                out[f"restore_{key}"] = ids_restore.numpy().astype(np.int32)
                # x[...,0] = 2*(n*L + l)  -> recover ids_keep exactly as the reference gathered them
                out[f"keep_{key}"] = ((xm[..., 0].numpy() / 2).astype(np.int64) % L).astype(np.int32)
    # forced ties straddling len_keep (L=16, ratio 0.5 -> keep 8): values chosen so that the
    # 7th..10th smallest are equal.
    noise = torch.tensor([[.9, .5, .1, .5, .2, .5, .3, .5, .05, .95, .15, .85, .25, .75, .35, .65]] * 2)
    noise[1] = noise[1].flip(0)
    import utils.imaging_model_related as imr
    real_rand = torch.rand
    try:
        imr.torch.rand = lambda *a, **k: noise.clone()
        m = Masker("random", 0.5, None)
        x = torch.arange(2 * 16 * 2, dtype=torch.float32).reshape(2, 16, 2)
        xm, mask, ids_restore = m(x)
    finally:
        imr.torch.rand = real_rand
    out["ties_noise"] = noise.numpy()
    out["ties_mask_refcpu"] = mask.numpy().astype(np.uint8)
    out["ties_restore_refcpu"] = ids_restore.numpy().astype(np.int32)
    out["ties_restore_stable"] = torch.argsort(torch.argsort(noise, dim=1, stable=True), dim=1, stable=True).numpy().astype(np.int32)
    save("g3_masker.npz", **out)


def g4_gather():
    torch.manual_seed(5)
    x = torch.randn(2, 64, 17, requires_grad=True)
    torch.manual_seed(7)
    noise = torch.rand(2, 64)
    torch.manual_seed(7)
    xm, mask, ids_restore = Masker("random", 0.7, None)(x)
    cot = torch.randn_like(xm)
    (xm * cot).sum().backward()
    save("g4_gather.npz", x=x.detach().numpy(), noise=noise.numpy(), xm=xm.detach().numpy(),
         mask=mask.numpy(), restore=ids_restore.numpy(), cot=cot.numpy(), dx=x.grad.numpy())


def g5_clip():
    out = {}
    for B in (4, 8, 256):
        for lam in (0.5, 0.3):
            torch.manual_seed(B)
            a = torch.randn(B, 128, requires_grad=True)
            b = torch.randn(B, 128, requires_grad=True)
            loss, logits, labels = CLIPLoss(0.1, lam)(a, b)
            loss.backward()
            key = f"B{B}_l{int(lam * 10)}"
            out.update({f"a_{key}": a.detach().numpy(), f"b_{key}": b.detach().numpy(),
                        f"loss_{key}": loss.detach().numpy(), f"labels_{key}": labels.numpy(),
                        f"da_{key}": a.grad.numpy(), f"db_{key}": b.grad.numpy()})
            if B <= 8:
                out[f"logits_{key}"] = logits.detach().numpy()
            else:
                out[f"logits_diag_{key}"] = logits.detach().diagonal().numpy()
                out[f"logits_row0_{key}"] = logits.detach()[0].numpy()
    # "global" case: 8 rank shards of 4 -> reference loss on the concatenated batch of 32
    torch.manual_seed(99)
    a = torch.randn(32, 128, requires_grad=True)
    b = torch.randn(32, 128, requires_grad=True)
    loss, logits, labels = CLIPLoss(0.1, 0.5)(a, b)
    loss.backward()
    out.update(dict(glob_a=a.detach().numpy(), glob_b=b.detach().numpy(), glob_loss=loss.detach().numpy(),
                    glob_logits=logits.detach().numpy(), glob_da=a.grad.numpy(), glob_db=b.grad.numpy()))
    try:
        CLIPLoss(0.1, 1.5)
        out["lambda_raises"] = np.array(0)
    except ValueError:
        out["lambda_raises"] = np.array(1)
    save("g5_clip.npz", **out)


def g6_patchify():
    torch.manual_seed(3)
    x = torch.rand(2, 9, 10, 16, 16)
    p = patchify(x, [5, 8, 8])
    back = unpatchify(p, x.shape, [5, 8, 8])
    pred = torch.rand_like(p)
    mask = (torch.rand(2, p.shape[1]) > 0.3).float()
    mse_all = torch.nn.MSELoss(reduction="none")(pred, p).mean()
    mse_masked = torch.nn.MSELoss(reduction="none")(pred[mask == 1], p[mask == 1]).mean()
    save("g6_patchify.npz", x=x.numpy(), patches=p.numpy(), roundtrip_equal=np.array(int(torch.equal(back, x))),
         pred=pred.numpy(), mask=mask.numpy(), mse_all=mse_all.numpy(), mse_masked=mse_masked.numpy())


def _reference_function(rel_path, name):
    """Execute ONE function definition of a reference module whose top-level imports are not installed here
    (utils/general.py imports skimage): the definition is parsed out of the file and compiled as is."""
    import ast
    from typing import Union
    path = os.path.join(REF, rel_path)
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np, "torch": torch, "Union": Union}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def g7_cine():
    """image_normalization (utils/general.py:139-147) applied where Cardiac3DplusTAllAX.__getitem__ applies it
    (datasets/datasets.py:414-417): on the (S, T, H, W) volume after the x=y mirror, LAX+SAX concat and transpose.
    The stored arrays are synthetic (UK Biobank data is restricted): fp32 'intensities' [H, W, S, T]."""
    image_normalization = _reference_function("utils/general.py", "image_normalization")
    out = {}
    for tag, (H, W, s_tot, T, z_min, z_num, seed) in {"a": (16, 24, 8, 10, 1, 6, 0), "b": (16, 16, 7, 50, 1, 6, 1)}.items():
        rng = np.random.RandomState(seed)
        lax = (rng.rand(H, W, 3, T) * 900 + rng.randn(H, W, 3, T) * 30).astype(np.float32)
        sax = (rng.rand(H, W, s_tot, T) * 1400 - 100).astype(np.float32)
        sax_im = np.moveaxis(sax[..., z_min:z_min + z_num, :], 1, 0)          # datasets.py:166-167,170
        lax_im = np.moveaxis(lax, 1, 0)                                        # datasets.py:171
        im = np.transpose(np.concatenate([lax_im, sax_im], axis=2), (2, 3, 0, 1))   # datasets.py:414-415
        norm = image_normalization(im)                                         # datasets.py:417
        assert norm.dtype == np.float32
        out.update({f"{tag}_lax": lax, f"{tag}_sax": sax, f"{tag}_z": np.array([z_min, z_num]), f"{tag}_norm": norm})
    save("g7_cine.npz", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1_patch_embed, g2_pos_embed, g3_masker, g4_gather, g5_clip, g6_patchify, g7_cine):
        if not only or fn.__name__.split("_")[0] in only:
            fn()
