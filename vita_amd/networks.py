"""Drop-in mirrors of the reference modules on the pretraining hot path.  Same constructor kwargs,
attributes, return tuples and state_dict keys as the reference; the arithmetic runs in libvita_hip.so.

  PatchEmbed                networks/tokenizers.py:11-56
  Masker                    utils/imaging_model_related.py:230-286
  sincos_pos_embed          utils/imaging_model_related.py:8-100   (init-time table, host side)
  ImagingMaskedEncoder      networks/imaging_encoders.py:11-145
  ViTDecoder / ImagingMaskedDecoder   networks/imaging_decoders.py:40-58, 230-305
  TabularViT                networks/tabular_encoders.py:16-112   (Stage-II other tower)
  SimCLRProjectionHead      lightly.models.modules (call sites models/MultimodalSimCLR.py:51-53, 62-64)
"""
from __future__ import annotations

import os

from typing import Optional

import numpy as np
import torch
from torch import nn
from torch.utils.checkpoint import checkpoint

from . import ops
from .nn import (Block, Dims, _Shadows, _require_cuda, bf16_twin, grad_done, grad_target, layer_norm_padded, linear_padded,
                 pad_tokens, wgrad)

__all__ = ["PatchEmbed", "Masker", "sincos_pos_embed", "ImagingMaskedEncoder", "ViTDecoder", "ImagingMaskedDecoder",
           "TabularViT", "SimCLRProjectionHead", "token_mean"]


def _mark(t: torch.Tensor) -> torch.Tensor:
    t._vita_padded = True
    return t


def as_padded(x: torch.Tensor, D: int, Dp: int) -> torch.Tensor:
    """Return the zero-padded [B,N,Dp] tensor behind a [B,N,D] view produced by this package, or pad."""
    base = x._base
    if (base is not None and getattr(base, "_vita_padded", False) and base.dim() == 3 and base.shape[-1] == Dp
            and x.shape == base.shape[:2] + (D,) and x.stride() == base.stride() and x.storage_offset() == base.storage_offset()):
        return base
    return pad_tokens(x.float(), Dp)


# ----------------------------------------------------------------------------- pos-embed (host, init time)
def _sincos_1d(embed_dim, pos):
    omega = torch.arange(embed_dim // 2, dtype=torch.float)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    out = torch.einsum("m,d->md", pos.reshape(-1), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def sincos_pos_embed(embed_dim, grid_size, cls_token=False, use_both_axes=True, circular_pe=False, shift_patch=(0, 0, 0)):
    """Fixed 3D+T sin/cos table with the LAX/SAX flag channel (same signature as the reference)."""
    if circular_pe or tuple(shift_patch) != (0, 0, 0):
        raise NotImplementedError("circular_pe / shift_patch are not used by the shipped configs")
    grid_size = tuple(int(g) for g in grid_size)

    def multi(dim, sizes):
        g = torch.stack(torch.meshgrid(*[torch.arange(0, e, dtype=torch.float32) for e in sizes], indexing="ij"), 0)
        return torch.cat([_sincos_1d(dim // len(sizes), g[i]) for i in range(len(sizes))], dim=1)

    if use_both_axes:
        assert len(grid_size) >= 3, "Grid_size should be at least 3D for positional embeding with long axis"
        assert (embed_dim - 1) % (len(grid_size) * 2) == 0, "Each dimension has 2 channels (sin, cos)"
        lax = multi(embed_dim - 1, (3, *grid_size[1:]))
        sax = multi(embed_dim - 1, (grid_size[0] - 3, *grid_size[1:]))
        flag = torch.cat([torch.zeros(lax.shape[0], 1), torch.ones(sax.shape[0], 1)], 0)
        pe = torch.cat([flag, torch.cat([lax, sax], 0)], 1)
    else:
        assert len(grid_size) >= 2, "Grid_size should be at least 2D"
        assert embed_dim % (len(grid_size) * 2) == 0, "Each dimension has 2 channels (sin, cos)"
        pe = multi(embed_dim, grid_size)
    if cls_token:
        pe = torch.cat([torch.zeros(1, embed_dim), pe], 0)
    return pe


# ----------------------------------------------------------------------------- tokenizer
class _TokenizeFn(torch.autograd.Function):
    """im2col of the selected tokens -> bf16 MFMA GEMM (+bias) -> (+pos-embed, cls row).
    with_cls: output [B, 1+n, Dp]; else [B, n, Dp]."""

    @staticmethod
    def forward(ctx, img, w, b, cls, pos, ids, holder, dims, with_cls):
        D, Dp, P, Pp, patch = dims
        B = img.shape[0]
        L = (img.shape[1] * (img.shape[2] // patch[0]) * (img.shape[3] // patch[1]) * (img.shape[4] // patch[2]))
        n = ids.shape[1] if ids is not None else L
        ws = holder.get("w", w, lambda p: ops.pack_weight(p.reshape(D, P), Dp, Pp))
        bs = holder.get("b", b, lambda p: ops.pack_vec(p, Dp)) if b is not None else \
            holder.get("b0", w, lambda p: torch.zeros(Dp, dtype=torch.float32, device=p.device))
        cols = ops.im2col(img, patch, ids, n, Pp, with_cls_row=with_cls)
        rows = cols.shape[0]
        y = torch.empty(rows, Dp, dtype=torch.float32, device=img.device)
        ops.gemm(cols, ws, y, rows, Dp, Pp, Pp, Pp, Dp, bias=bs, flops=2.0 * B * n * D * P)
        if with_cls:
            posl = pos.reshape(-1, D) if pos is not None else None
            out = ops.tokens_assemble(y, Dp, cls.reshape(-1), posl, ids, B, n, D, Dp, inplace_with_cls_row=True)
        else:
            out = y.view(B, n, Dp)
        ctx.save_for_backward(cols)
        ctx.meta = (B, n, D, Dp, P, Pp, with_cls, w.shape, b is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        (cols,) = ctx.saved_tensors
        B, n, D, Dp, P, Pp, with_cls, wshape, has_b = ctx.meta
        rows = cols.shape[0]
        dyb = bf16_twin(dout, rows, Dp)
        dout = dout.contiguous().view(rows, Dp)
        if dyb is None:
            dyb = ops.cast_bf16(dout, Dp, rows, D, Dp)
        need = ctx.needs_input_grad            # (img, w, b, cls, pos, ids, holder, dims, with_cls)
        dw = wgrad(dyb, Dp, cols, Pp, D, P, rows).reshape(wshape) if need[1] else None
        db = ops.colsum(dyb, Dp, rows, D) if (has_b and need[2]) else None
        dcls = None
        if with_cls:
            if need[3]:
                dcls = ops.colsum(dout, (n + 1) * Dp, B, D).view(1, 1, D)      # sum over batch of row 0
            if db is not None:
                db = db - ops.colsum(dyb, (n + 1) * Dp, B, D)   # the zero-patch cls rows carry no bias grad
        return None, dw, db, dcls, None, None, None, None, None


class PatchEmbed(nn.Module):
    def __init__(self, im_shape, in_channels: int = 1, patch_size=[1, 16, 16], out_channels: int = 256,
                 flatten: bool = True, bias: bool = True, norm_layer: Optional[nn.Module] = None, **kwargs):
        super().__init__()
        assert len(patch_size) == 3, "Patch size should be 3D"
        assert in_channels == 1, "Only supporting input channel size as 1"
        if not flatten or norm_layer is not None:
            raise NotImplementedError("vita_amd.PatchEmbed: flatten=True, norm_layer=None (the reference configuration)")
        self.im_shape = im_shape
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.flatten = flatten
        self.patch_size = tuple(int(p) for p in patch_size)
        self.proj = nn.Conv3d(in_channels, out_channels, kernel_size=self.patch_size, stride=self.patch_size, bias=bias)
        self.norm = nn.Identity()
        self.grid_size = (im_shape[0], im_shape[1] // patch_size[0], im_shape[2] // patch_size[1], im_shape[3] // patch_size[2])
        self.num_patches = np.prod(self.grid_size)
        P = int(np.prod(self.patch_size))
        self._dims = (out_channels, ops.rup(out_channels, 64), P, ops.rup(P, 8), self.patch_size)
        self._sh = _Shadows()

    def tokens_padded(self, x, ids=None, cls=None, pos=None):
        _require_cuda(x)
        x = x.reshape(x.shape[0], -1, *self.im_shape[-3:]).float().contiguous()
        return _mark(_TokenizeFn.apply(x, self.proj.weight, self.proj.bias, cls, pos, ids, self._sh, self._dims, cls is not None))

    def forward(self, x):
        """(B, S, T, H, W) -> (B, num_patches, out_channels)"""
        return self.tokens_padded(x)[..., :self.out_channels]


# ----------------------------------------------------------------------------- masker
class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, ids, D, fill, pos):
        B, L, Dp = xp.shape
        ids = ids.contiguous()
        n = ids.shape[1]
        out = ops.row_gather(xp, Dp, L, ids, n, n, B, D, Dp, fill=fill.reshape(-1) if fill is not None else None,
                             pos=pos.reshape(-1, D) if pos is not None else None)
        ctx.save_for_backward(ids)
        ctx.meta = (B, L, n, D, Dp, fill is not None, fill.shape if fill is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        B, L, n, D, Dp, has_fill, fshape = ctx.meta
        dx, dfill = ops.row_scatter(dout.contiguous(), Dp, n, ids, n, L, B, D, Dp, want_dfill=has_fill)
        return dx, None, None, dfill.view(fshape) if has_fill else None, None


class Masker:
    """MAE random masking.  Consumes exactly one torch.rand(N, L, device) from the global generator per
    call, like the reference; the argsort / inverse permutation / mask run in one HIP kernel
    (ties ordered by index = torch's GPU sort behaviour)."""

    def __init__(self, mask_type, mask_ratio, grid_size, **kwargs):
        self.mask_ratio = mask_ratio
        if mask_type == "random":
            self.masking_strategy = self.random_masking
        else:
            raise NotImplementedError

    def __call__(self, x):
        _require_cuda(x)
        mask, ids_restore, ids_keep = self.masking_strategy(input_size=x.shape, device=x.device)
        D = x.shape[-1]
        xp = as_padded(x, D, ops.rup(D, 64))
        x_masked = _GatherFn.apply(xp, ids_keep, D, None, None)[..., :D]
        return x_masked, mask, ids_restore

    def random_masking(self, input_size, device, **kwargs):
        N, L, D = input_size
        len_keep = int(L * (1 - self.mask_ratio))
        noise = torch.rand(N, L, device=device)
        ids_shuffle, ids_restore, mask = ops.argsort_mask(noise, len_keep)
        return mask, ids_restore, ids_shuffle[:, :len_keep]


# ----------------------------------------------------------------------------- encoder
class ImagingMaskedEncoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.enc_embed_dim = kwargs.get("enc_embed_dim")
        self.patch_size = kwargs.get("patch_size")
        self.circular_pe = kwargs.get("circular_pe")
        self.use_enc_pe = kwargs.get("use_enc_pe")
        self.mask_loss = kwargs.get("mask_loss")
        self.shift_size = kwargs.get("shift_size")
        name = kwargs.get("patch_embed_cls", "PatchEmbed") or "PatchEmbed"
        if name != "PatchEmbed":
            raise NotImplementedError(f"vita_amd: patch_embed_cls={name!r}; only 'PatchEmbed' is on the hot path")
        self.patch_embed_cls = PatchEmbed
        self.img_shape = kwargs.get("img_shape")
        self.patch_in_channels = kwargs.get("patch_in_channels", 1) or 1
        self.use_both_axes = kwargs.get("use_both_axes")
        self.patch_p_num = np.prod(kwargs.get("patch_size")) * self.patch_in_channels
        self.grad_checkpointing = kwargs.get("grad_checkpointing", False)
        self.pixel_unshuffle_scale = kwargs.get("pixel_unshuffle_scale")
        self.patch_embed = PatchEmbed(self.img_shape, in_channels=self.patch_in_channels,
                                      patch_size=kwargs.get("patch_size"), out_channels=kwargs.get("enc_embed_dim"))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.patch_embed.out_channels))
        self.enc_pos_embed = nn.Parameter(torch.zeros(1, 1 + self.patch_embed.num_patches, self.patch_embed.out_channels),
                                          requires_grad=False)
        self.encoder = nn.ModuleList([Block(dim=self.patch_embed.out_channels, num_heads=kwargs.get("enc_num_heads"),
                                            mlp_ratio=kwargs.get("mlp_ratio"), qkv_bias=True)
                                      for _ in range(kwargs.get("enc_depth"))])
        self.encoder_norm = nn.LayerNorm(self.patch_embed.out_channels)
        self.using_masker = False if kwargs.get("mask_ratio") == 0.0 else True
        if self.using_masker:
            self.masker = Masker(mask_type=kwargs.get("mask_type"), mask_ratio=kwargs.get("mask_ratio"),
                                 grid_size=self.patch_embed.grid_size)
        self.initialize_parameters()

    def initialize_parameters(self):
        pe = sincos_pos_embed(self.enc_embed_dim, self.patch_embed.grid_size, cls_token=True,
                              use_both_axes=self.use_both_axes, circular_pe=bool(self.circular_pe))
        if not self.use_enc_pe:
            pe[:, 0] = 0
        self.enc_pos_embed.data.copy_(pe.unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _run_blocks(self, xp, collect=None):
        D = self.enc_embed_dim
        for blk in self.encoder:
            if collect is not None:
                collect.append(xp[:, 1:, :D])
            if self.grad_checkpointing and torch.is_grad_enabled():
                xp = checkpoint(blk.forward_padded, xp, use_reentrant=False)
            else:
                xp = blk.forward_padded(xp)
        return xp

    def forward_padded(self, x):
        """Returns (tokens fp32 [B, 1+keep, Dp] zero padded, mask, ids_restore)."""
        _require_cuda(x)
        pos = self.enc_pos_embed if self.use_enc_pe else None
        if self.using_masker:
            L = int(self.patch_embed.num_patches)
            # one torch.rand(N, L) per forward, as utils/imaging_model_related.py:270
            mask, ids_restore, ids_keep = self.masker.masking_strategy(input_size=(x.shape[0], L, self.enc_embed_dim), device=x.device)
            ids = ids_keep.contiguous()
        else:
            mask, ids_restore, ids = None, None, None
        # masked rows are never embedded: the GEMM runs on the kept tokens only (row-wise identical result)
        xp = self.patch_embed.tokens_padded(x, ids=ids, cls=self.cls_token, pos=pos)
        xp = self._run_blocks(xp)
        return _mark(layer_norm_padded(xp, self.encoder_norm)), mask, ids_restore

    def forward(self, x):
        xp, mask, ids_restore = self.forward_padded(x)
        return xp[..., :self.enc_embed_dim], mask, ids_restore

    def forward_with_skip_connection(self, x):
        _require_cuda(x)
        pos = self.enc_pos_embed if self.use_enc_pe else None
        xp = self.patch_embed.tokens_padded(x, ids=None, cls=self.cls_token, pos=pos)
        hidden = []
        xp = self._run_blocks(xp, collect=hidden)
        out = _mark(layer_norm_padded(xp, self.encoder_norm))
        return out[..., :self.enc_embed_dim], hidden


# ----------------------------------------------------------------------------- decoder (adjacent)
class ViTDecoder(nn.Module):
    def __init__(self, dim, num_heads, depth, mlp_ratio, norm_layer=nn.LayerNorm, grad_checkpointing: bool = False):
        super().__init__()
        self.network = nn.ModuleList([Block(dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                      for _ in range(depth)])
        self.norm = norm_layer(dim)
        self.grad_checkpointing = grad_checkpointing
        self.dim = dim

    def forward_padded(self, xp):
        for blk in self.network:
            if self.grad_checkpointing and torch.is_grad_enabled():
                xp = checkpoint(blk.forward_padded, xp, use_reentrant=False)
            else:
                xp = blk.forward_padded(xp)
        return _mark(layer_norm_padded(xp, self.norm))

    def forward(self, x):
        return self.forward_padded(as_padded(x, self.dim, ops.rup(self.dim, 64)))[..., :self.dim]


class ImagingMaskedDecoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.grid_size = kwargs.get("grid_size")
        self.dec_embed_dim = kwargs.get("dec_embed_dim")
        self.enc_embed_dim = kwargs.get("enc_embed_dim")
        self.use_both_axes = kwargs.get("use_both_axes")
        self.decoder_num_patches = kwargs.get("num_patches")
        self.use_enc_pe = kwargs.get("use_enc_pe")
        self.decoder_embed = nn.Linear(kwargs.get("enc_embed_dim"), kwargs.get("dec_embed_dim"))
        self.dec_pos_embed = nn.Parameter(torch.zeros(1, 1 + self.decoder_num_patches, kwargs.get("dec_embed_dim")),
                                          requires_grad=False)
        self.decoder = ViTDecoder(dim=kwargs.get("dec_embed_dim"), num_heads=kwargs.get("dec_num_heads"),
                                  depth=kwargs.get("dec_depth"), mlp_ratio=kwargs.get("mlp_ratio"))
        self.recon_head = nn.Linear(in_features=kwargs.get("dec_embed_dim"), out_features=kwargs.get("head_out_dim"))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.dec_embed_dim), requires_grad=True)
        self.head_out_dim = int(kwargs.get("head_out_dim"))
        self._sh_embed, self._sh_head = _Shadows(), _Shadows()
        self.initialize_parameters()

    def initialize_parameters(self):
        pe = sincos_pos_embed(self.dec_embed_dim, self.grid_size, cls_token=True, use_both_axes=self.use_both_axes)
        if not self.use_enc_pe:
            pe[:, 0] = 0
        self.dec_pos_embed.data.copy_(pe.unsqueeze(0))
        torch.nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    _init_weights = ImagingMaskedEncoder._init_weights

    def forward_logits_padded(self, x, ids_restore):
        """-> pre-sigmoid head output fp32 [B, 1+L, Pp] (row 0 = cls), zero padded."""
        _require_cuda(x)
        De, Dd = self.enc_embed_dim, self.dec_embed_dim
        Dpe, Dpd = ops.rup(De, 64), ops.rup(Dd, 64)
        xp = as_padded(x, De, Dpe)
        yp = linear_padded(xp, self.decoder_embed, self._sh_embed, Dpd)
        B, n1, _ = yp.shape
        if ids_restore is not None:
            L = ids_restore.shape[1]
            # row 0 keeps the cls token; token l takes shuffled row 1+ids_restore[l]; indices beyond the
            # encoder's rows select the mask token (cat + gather of networks/imaging_decoders.py:288-291)
            ids = torch.cat([torch.zeros(B, 1, dtype=torch.int64, device=x.device), ids_restore + 1], dim=1)
            xp = _GatherFn.apply(yp, ids, Dd, self.mask_token, self.dec_pos_embed)
        else:
            ids = torch.arange(n1, dtype=torch.int64, device=x.device).expand(B, n1)
            xp = _GatherFn.apply(yp, ids, Dd, None, self.dec_pos_embed)
        xp = self.decoder.forward_padded(xp)
        return linear_padded(xp, self.recon_head, self._sh_head, ops.rup(self.head_out_dim, 8))

    def forward(self, x, ids_restore):
        z = self.forward_logits_padded(x, ids_restore)
        # boundary convenience only; the fused train step uses vita_sigmoid_mse_patch on z directly
        return torch.sigmoid(z[:, 1:, :self.head_out_dim])


# ----------------------------------------------------------------------------- Stage II: tabular tower + heads
CHECK_TABULAR_INPUT = os.environ.get("VITA_CHECK_TABULAR") == "1"


class TabularViT(nn.Module):
    """Drop-in for networks/tabular_encoders.py:16-112: same constructor (args namespace + feature-name dict),
    parameter names (numerical_fe.i, single_c_fe.i, multi_c_fe.i, cls_token, enc_pos_embed, network.i, norm) and
    output [B, 1 + n_features, D].  The 117 per-feature tokenizers are evaluated as two batched torch ops
    (they are a few KB of parameters); the transformer blocks and the LayerNorm run in libvita_hip.so."""

    def __init__(self, args, all_feature_names: dict, **kwargs) -> None:
        super().__init__()
        self.grad_checkpointing = args.grad_checkpointing
        self.embedding_dim = args.embedding_dim
        self.all_feature_names = all_feature_names
        self.num_numerical = len(all_feature_names["numerical"])
        self.num_single_c = len(all_feature_names["single_categorical"])
        self.num_multiple_c = len(all_feature_names["multi_categorical"])
        self.num_features = self.num_numerical + self.num_single_c + self.num_multiple_c
        D = self.embedding_dim
        self.numerical_fe = nn.ModuleList([nn.Linear(1, D) for _ in range(self.num_numerical)])
        self.single_c_fe = nn.ModuleList([nn.Embedding(2, D) for _ in range(self.num_single_c)])
        self.multi_c_fe = nn.ModuleList([nn.Embedding(int(v[0]), D) for _, v in all_feature_names["multi_categorical"].items()])
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.enc_pos_embed = nn.Parameter(torch.zeros(1, 1 + self.num_features, D), requires_grad=False)
        self.network = nn.ModuleList([Block(D, args.encoder_num_heads, args.encoder_mlp_ratio, qkv_bias=True,
                                            norm_layer=nn.LayerNorm) for _ in range(args.encoder_num_layers)])
        self.norm = nn.LayerNorm(D)
        self.initialize_parameters()

    def initialize_parameters(self):
        D = self.embedding_dim
        pe = _sincos_1d(D + 1, torch.arange(self.num_features, dtype=torch.float32))
        pe = torch.cat([torch.zeros(1, D), pe[:, :-1]], dim=0)
        self.enc_pos_embed.data.copy_(pe.unsqueeze(0))
        torch.nn.init.normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    _init_weights = ImagingMaskedEncoder._init_weights

    def _feature_params(self):
        """(weights, biases) in feature order numerical | single | multi; bias None for the embedding tables."""
        ws = [l.weight for l in self.numerical_fe] + [t.weight for t in self.single_c_fe] + [t.weight for t in self.multi_c_fe]
        bs = [l.bias for l in self.numerical_fe] + [None] * (self.num_single_c + self.num_multiple_c)
        return ws, bs

    def _tables(self, dev):
        """Device tables of the per-feature parameter addresses (rebuilt only when a parameter moved, e.g. into
        FlatAdam's flat buffer) and the class counts (0 = numerical)."""
        ws, bs = self._feature_params()
        key = (dev, tuple(w.data_ptr() for w in ws), tuple(b.data_ptr() if b is not None else 0 for b in bs))
        if getattr(self, "_tab_key", None) != key:
            self._tab_key = key
            self._wptr = torch.tensor(key[1], dtype=torch.int64, device=dev)
            self._bptr = torch.tensor(key[2], dtype=torch.int64, device=dev)
            ncls = [0] * self.num_numerical + [t.num_embeddings for t in self.single_c_fe] + [t.num_embeddings for t in self.multi_c_fe]
            self._ncls = torch.tensor(ncls, dtype=torch.int32, device=dev)
        return self._wptr, self._bptr, self._ncls

    def check_categorical_range(self, x):
        """The reference's nn.Embedding raises IndexError on a categorical value outside [0, n_classes); the tokeniser
        kernel clamps instead (memory safety) and would train silently on class 0 / n-1.  This host-side check (one
        device sync; on with VITA_CHECK_TABULAR=1, or call it from a data-validation pass) restores the error, and also
        rejects non-finite values, whose float -> int64 cast is undefined."""
        if not torch.isfinite(x).all():
            raise ValueError("vita_amd.TabularViT: non-finite tabular value")
        ncls = [t.num_embeddings for t in self.single_c_fe] + [t.num_embeddings for t in self.multi_c_fe]
        if ncls:
            cat = x[:, self.num_numerical:self.num_features].long()
            hi = torch.tensor(ncls, device=x.device)
            bad = (cat < 0) | (cat >= hi)
            if bad.any():
                b, f = [int(v) for v in bad.nonzero()[0]]
                raise IndexError(f"vita_amd.TabularViT: categorical feature {self.num_numerical + f} of row {b} has index "
                                 f"{int(cat[b, f])}, out of range [0, {ncls[f]})")

    def forward_embeddings(self, x):
        """(B, n_features) -> (B, n_features, D), feature order numerical | single | multi (no cls, no pos-embed)."""
        _require_cuda(x)
        return self._embed_padded(x, with_pos=False)[:, 1:, :self.embedding_dim]

    def _embed_padded(self, x, with_pos=True):
        ws, bs = self._feature_params()
        if CHECK_TABULAR_INPUT:
            self.check_categorical_range(x)
        return _mark(_TabEmbedFn.apply(x.float().contiguous(), self, with_pos, self.cls_token, *ws,
                                       *[b for b in bs if b is not None]))

    def forward_padded(self, x):
        _require_cuda(x)
        # cls token + 117 per-feature tokenisers + pos-embed: one launch (vita_tabular_embed_fwd)
        xp = self._embed_padded(x)
        for blk in self.network:
            if self.grad_checkpointing and torch.is_grad_enabled():
                xp = checkpoint(blk.forward_padded, xp, use_reentrant=False)
            else:
                xp = blk.forward_padded(xp)
        return _mark(layer_norm_padded(xp, self.norm))

    def forward(self, x: torch.Tensor, num_paddings: int = None) -> torch.Tensor:
        if num_paddings is not None:
            raise NotImplementedError("vita_amd.TabularViT: num_paddings is not used by the shipped configs")
        return self.forward_padded(x)[..., :self.embedding_dim]


class _TabEmbedFn(torch.autograd.Function):
    """cls + per-feature tokens (+ pos-embed) of TabularViT in one launch; the backward adds the batch sums straight
    into the parameters' fp32 .grad buffers where they exist (FlatAdam's flat gradient), one launch for all 185
    tensors, and reports them like every other in-place gradient."""

    @staticmethod
    def forward(ctx, x, mod, with_pos, cls, *params):
        D = mod.embedding_dim
        wptr, bptr, ncls = mod._tables(x.device)
        pos = mod.enc_pos_embed.reshape(-1, D) if with_pos else None
        out = ops.tabular_embed_fwd(x, wptr, bptr, ncls, cls.reshape(-1), pos, D, ops.rup(D, 64))
        ctx.mod = mod
        ctx.save_for_backward(x, ncls)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        x, ncls = ctx.saved_tensors
        D = mod.embedding_dim
        dev = x.device
        dout = dout.contiguous()
        ws, bs = mod._feature_params()
        need = ctx.needs_input_grad                     # (x, mod, with_pos, cls, *weights, *biases)
        nF = len(ws)
        plist = [mod.cls_token] + ws + [b for b in bs if b is not None]
        need_p = list(need[3:])
        targets = [grad_target(p) if n else None for p, n in zip(plist, need_p)]
        fresh_rows = sum(p.numel() // D for p, n, t in zip(plist, need_p, targets) if n and t is None)
        fresh = torch.zeros(max(fresh_rows, 1), D, dtype=torch.float32, device=dev)
        grads, addrs, row = [], [], 0
        for p, n, t in zip(plist, need_p, targets):
            if not n:
                grads.append(None); addrs.append(0)
            elif t is not None:
                grads.append(None); addrs.append(t.data_ptr())
            else:
                r = p.numel() // D
                g = fresh[row:row + r].view(p.shape); row += r
                grads.append(g); addrs.append(g.data_ptr())
        key = tuple(addrs)
        cache = getattr(mod, "_grad_tables", None)
        if cache is None or cache[0] != key:
            gw = torch.tensor(addrs[1:1 + nF], dtype=torch.int64, device=dev)
            it = iter(addrs[1 + nF:])
            gb = torch.tensor([next(it) if b is not None else 0 for b in bs], dtype=torch.int64, device=dev)
            cache = (key, gw, gb)
            if fresh_rows == 0:
                mod._grad_tables = cache              # stable addresses (the flat gradient buffer): keep the tables
        _, gw, gb = cache
        dcls = None
        if addrs[0]:
            dcls = targets[0].view(-1) if targets[0] is not None else grads[0].view(-1)
        ops.tabular_embed_bwd(dout, dout.shape[-1], x, ncls, gw, gb, dcls, D)
        for p, n, t in zip(plist, need_p, targets):
            if n and t is not None:
                grad_done(p)
        return (None, None, None, *grads)


class _TokenMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, D):
        ctx.meta = (xp.shape, D)
        return ops.token_mean_fwd(xp.contiguous(), D)

    @staticmethod
    def backward(ctx, dy):
        (B, N, ld), D = ctx.meta
        return ops.token_mean_bwd(dy.float().contiguous(), B, N, D, ld), None


def token_mean(xp, D):
    """y.mean(dim=1) over ALL tokens (models/MultimodalSimCLR.py:121,133) of a zero-padded [B, N, Dp] tensor -> [B, D]."""
    return _TokenMeanFn.apply(xp, D)


class _HeadFn(torch.autograd.Function):
    """Linear(no bias) -> BatchNorm1d (batch statistics) -> ReLU -> Linear, all fp32 in libvita_hip.so
    (vita_sgemm_f32, vita_batchnorm_fwd / _bwd): no rocBLAS / ATen GEMM on the Stage-II product path."""

    @staticmethod
    def forward(ctx, x, w0, gamma, beta, w3, b3, bn):
        x = x.float().contiguous()
        B, Din = x.shape
        Hd, Dout = w0.shape[0], w3.shape[0]
        dev = x.device
        h = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        ops.sgemm(x, w0, h, B, Hd, Din, Din, Din, Hd)
        if bn.training or bn.running_mean is None:
            if bn.training and B == 1:       # torch's BatchNorm1d raises too: the batch variance of one row is 0
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(h.shape)}")
            track = bn.training and bn.track_running_stats and bn.running_mean is not None
            momentum = bn.momentum
            if track and bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
                if momentum is None:         # torch: momentum=None = cumulative moving average, factor 1 / batches seen
                    momentum = 1.0 / float(bn.num_batches_tracked.item())
            if momentum is None:
                momentum = 0.0
            a, mean, rstd = ops.batchnorm_fwd(h, gamma, beta, bn.eps, True, bn.running_mean if track else None,
                                              bn.running_var if track else None, momentum)
            ctx.train = True
        else:
            a = ops.batchnorm_eval(h, gamma, beta, bn.eps, True, bn.running_mean, bn.running_var)
            mean = rstd = torch.empty(0, device=dev)
            ctx.train = False
        z = torch.empty(B, Dout, dtype=torch.float32, device=dev)
        ops.sgemm(a, w3, z, B, Dout, Hd, Hd, Hd, Dout)
        if b3 is not None:
            ops.add_bias_rows(z, b3)
        ctx.save_for_backward(x, w0, gamma, w3, h, a, mean, rstd)
        ctx.has_b3 = b3 is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        if not ctx.train:
            raise NotImplementedError("vita_amd.SimCLRProjectionHead: backward in eval mode (running statistics) is not implemented")
        x, w0, gamma, w3, h, a, mean, rstd = ctx.saved_tensors
        dz = dz.float().contiguous()
        B, Din = x.shape
        Hd, Dout = w0.shape[0], w3.shape[0]
        dev = x.device
        need = ctx.needs_input_grad                     # (x, w0, gamma, beta, w3, b3, bn)
        db3 = ops.colsum_small(dz) if (ctx.has_b3 and need[5]) else None
        dw3 = None
        if need[4]:
            dw3 = torch.empty(Dout, Hd, dtype=torch.float32, device=dev)
            ops.sgemm(dz, a, dw3, Dout, Hd, B, Dout, Hd, Hd, 1, 1)
        da = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        ops.sgemm(dz, w3, da, B, Hd, Dout, Dout, Hd, Hd, 0, 1)
        dh, dgamma, dbeta = ops.batchnorm_bwd(da, h, a, gamma, mean, rstd, True)
        dw0 = None
        if need[1]:
            dw0 = torch.empty(Hd, Din, dtype=torch.float32, device=dev)
            ops.sgemm(dh, x, dw0, Hd, Din, B, Hd, Din, Din, 1, 1)
        dx = None
        if need[0]:
            dx = torch.empty(B, Din, dtype=torch.float32, device=dev)
            ops.sgemm(dh, w0, dx, B, Din, Hd, Hd, Din, Din, 0, 1)
        return dx, dw0, dgamma if need[2] else None, dbeta if need[3] else None, dw3, db3, None


class SimCLRProjectionHead(nn.Module):
    """lightly's SimCLR projection head as the reference calls it (in, hidden, out): Linear(bias=False) ->
    BatchNorm1d -> ReLU -> Linear.  lightly is not pinned by the reference (absent from requirements.txt) and
    not installed here: the layout follows lightly's published SimCLR head and is an assumption.
    Same sub-module / state_dict names (layers.0 / .1 / .3); the arithmetic runs in libvita_hip.so in fp32 (_HeadFn),
    BatchNorm with local (per-rank) batch statistics as in the reference (no SyncBN)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, output_dim: int = 128):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(input_dim, hidden_dim, bias=False), nn.BatchNorm1d(hidden_dim), nn.ReLU(),
                                    nn.Linear(hidden_dim, output_dim, bias=True))

    def forward(self, x):
        _require_cuda(x)
        l0, bn, _, l3 = self.layers
        return _HeadFn.apply(x, l0.weight, bn.weight, bn.bias, l3.weight, l3.bias, bn)
