"""Loss mirrors: CLIPLoss (utils/clip_loss.py:6-40) with optional cross-GPU negatives, and the fused
sigmoid + MSE reconstruction criterion (utils/losses.py:92-140 'mse' branch on the patchified image,
utils/imaging_model_related.py:103-117)."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from .nn import _require_cuda


def device_collectives(group) -> bool:
    """True when the group's backend moves device memory itself (RCCL, torch's "nccl")."""
    return "nccl" in str(dist.get_backend(group)).lower()


def all_reduce_any(t: torch.Tensor, group):
    """Blocking all-reduce(SUM) in place.  RCCL takes the device tensor.  Any other backend (gloo: CPU tests, and the
    two-ranks-on-one-GPU test) only ever sees HOST memory: device tensors are staged - D2H on the current stream (which
    orders the copy after the kernels that produced them, and blocks the host), the collective on the host copy, H2D.
    gloo's own CUDA path (worker-thread stream syncs and pinned staging of its own) hung 1 run in 4 between two processes
    sharing a GPU in round 2 (gpurun_out/r2_tests12.log: both ranks in finish() -> wait(), gloo worker without a Python
    frame) and is therefore never used."""
    if not t.is_cuda or device_collectives(group):
        dist.all_reduce(t, group=group)
        return t
    h = t.detach().cpu()
    dist.all_reduce(h, group=group)
    t.detach().copy_(h)
    return t


def gather_towers(a: torch.Tensor, b: torch.Tensor, group):
    """All-gather both towers' [B_local, d] embeddings in ONE message -> ([B_global, d], [B_global, d]);
    rank r's rows land at [r*B_local, (r+1)*B_local).  RCCL all_gather_into_tensor on device memory; other backends
    (gloo) get host tensors (see all_reduce_any)."""
    world = dist.get_world_size(group)
    Bl, d = a.shape
    packed = torch.stack([a, b]).contiguous()
    if device_collectives(group):
        gathered = torch.empty(world, 2, Bl, d, dtype=a.dtype, device=a.device)
        dist.all_gather_into_tensor(gathered, packed, group=group)
    else:
        hp = packed.detach().cpu()
        parts = [torch.empty_like(hp) for _ in range(world)]
        dist.all_gather(parts, hp, group=group)
        gathered = torch.stack(parts).to(a.device)
    return (gathered[:, 0].reshape(world * Bl, d).contiguous(), gathered[:, 1].reshape(world * Bl, d).contiguous())


def scatter_tower_grads(cross: torch.Tensor, group):
    """Backward of gather_towers: cross [2, B_global, d] holds this rank's gradient w.r.t. every rank's
    embeddings; returns the sum over ranks of the slice that belongs to this rank, [2, B_local, d]."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    _, Bg, d = cross.shape
    Bl = Bg // world
    send = cross.view(2, world, Bl, d).permute(1, 0, 2, 3).contiguous()
    if device_collectives(group):
        recv = torch.empty(2, Bl, d, dtype=cross.dtype, device=cross.device)
        dist.reduce_scatter_tensor(recv, send, group=group)
        return recv
    all_reduce_any(send, group)            # gloo has no reduce-scatter; host-staged for device tensors
    return send[rank].contiguous()


def clip_rank_forward(a, b, a_all, b_all, rank, tau, lam0):
    """What ONE rank computes in the forward of the (global-negative) CLIP loss, given its own L2-normalised rows
    a, b [Bl, d] and every rank's rows a_all, b_all [Bg, d] (rank r's rows at [r*Bl, (r+1)*Bl)):
    the rank's rows of logits (image->tabular) and of logits.T (tabular->image), its share of the global-batch loss
    (utils/clip_loss.py:33-38 with both CrossEntropy means taken over Bg) and d loss / d logits rows.
    Returns (loss_part [1], l_it [Bl,Bg], l_ti [Bl,Bg], d_it, d_ti, labels = rank*Bl + arange(Bl))."""
    Bl, d = a.shape
    Bg, off = a_all.shape[0], rank * Bl
    dev = a.device
    l_it = torch.empty(Bl, Bg, dtype=torch.float32, device=dev)
    l_ti = torch.empty(Bl, Bg, dtype=torch.float32, device=dev)
    ops.sgemm(a, b_all, l_it, Bl, Bg, d, d, d, Bg, 0, 0, alpha=1.0 / tau)
    ops.sgemm(b, a_all, l_ti, Bl, Bg, d, d, d, Bg, 0, 0, alpha=1.0 / tau)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    d_it = torch.empty_like(l_it); d_ti = torch.empty_like(l_ti)
    ops.ce_rows(l_it, Bg, Bl, Bg, off, lam0 / Bg, loss, dlogits=d_it)
    ops.ce_rows(l_ti, Bg, Bl, Bg, off, (1.0 - lam0) / Bg, loss, dlogits=d_ti)
    labels = off + torch.arange(Bl, device=dev)
    return loss, l_it, l_ti, d_it, d_ti, labels


def clip_rank_backward(a, b, a_all, b_all, d_it, d_ti, tau, cross_out=None):
    """One rank's backward: (da, db) [Bl, d] = the terms through this rank's own rows of the two logit blocks, and
    cross [2, Bg, d] = this rank's gradient with respect to EVERY rank's normalised embeddings through the gathered
    operand (d a_all += d_ti^T b, d b_all += d_it^T a); the all-gather's backward sums cross over ranks and hands
    each rank its slice.  With cross_out = (da, db) (single rank: Bg == Bl) the cross terms are added in place."""
    Bl, d = a.shape
    Bg = a_all.shape[0]
    dev = a.device
    da = torch.empty(Bl, d, dtype=torch.float32, device=dev)
    db = torch.empty(Bl, d, dtype=torch.float32, device=dev)
    ops.sgemm(d_it, b_all, da, Bl, d, Bg, Bg, d, d, 0, 1, alpha=1.0 / tau)
    ops.sgemm(d_ti, a_all, db, Bl, d, Bg, Bg, d, d, 0, 1, alpha=1.0 / tau)
    if cross_out is None:
        cross = torch.empty(2, Bg, d, dtype=torch.float32, device=dev)
        ops.sgemm(d_ti, b, cross[0], Bg, d, Bl, Bg, d, d, 1, 1, alpha=1.0 / tau)
        ops.sgemm(d_it, a, cross[1], Bg, d, Bl, Bg, d, d, 1, 1, alpha=1.0 / tau)
        return da, db, cross
    assert Bg == Bl
    ops.sgemm(d_ti, b, da, Bg, d, Bl, Bg, d, d, 1, 1, alpha=1.0 / tau, beta=1.0)
    ops.sgemm(d_it, a, db, Bg, d, Bl, Bg, d, d, 1, 1, alpha=1.0 / tau, beta=1.0)
    return da, db, None


class _ClipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out0, out1, tau, lam0, group, grad_world_scale):
        out0 = out0.float().contiguous(); out1 = out1.float().contiguous()
        a, ia = ops.l2_normalize_fwd(out0)
        b, ib = ops.l2_normalize_fwd(out1)
        # group is None = no exchange at all.  A group of ONE rank still runs every collective (CLIPLoss keeps such a
        # group only when asked to: it is how the RCCL calls are exercised on a one-GPU box)
        world, rank = (dist.get_world_size(group), dist.get_rank(group)) if group is not None else (1, 0)
        if group is not None:
            a_all, b_all = gather_towers(a, b, group)
        else:
            a_all, b_all = a, b
        loss, l_it, _l_ti, d_it, d_ti, labels = clip_rank_forward(a, b, a_all, b_all, rank, tau, lam0)
        if group is not None:
            all_reduce_any(loss, group)                                  # report the global-batch loss
        ctx.save_for_backward(a, b, ia, ib, a_all, b_all, d_it, d_ti)
        ctx.meta = (tau, group, world, grad_world_scale)
        ctx.mark_non_differentiable(l_it, labels)
        return loss[0], l_it, labels

    @staticmethod
    def backward(ctx, gloss, _gl, _glab):
        a, b, ia, ib, a_all, b_all, d_it, d_ti = ctx.saved_tensors
        tau, group, world, gws = ctx.meta
        if group is not None:
            da, db, cross = clip_rank_backward(a, b, a_all, b_all, d_it, d_ti, tau)
            recv = scatter_tower_grads(cross, group)                      # backward of the all-gather
            da += recv[0]; db += recv[1]
        else:
            da, db, _ = clip_rank_backward(a, b, a_all, b_all, d_it, d_ti, tau, cross_out=True)
        g0 = ops.l2_normalize_bwd(da, a, ia)
        g1 = ops.l2_normalize_bwd(db, b, ib)
        s = gloss * gws
        return g0 * s, g1 * s, None, None, None, None


class CLIPLoss(nn.Module):
    """Same constructor / return tuple as the reference.  With ``group`` set (or
    ``gather_distributed=True`` and torch.distributed initialised) every rank scores its local rows
    against the all-gathered other tower; the returned loss is the loss of the concatenated global
    batch, ``logits`` is [B_local, B_global] and ``labels = rank*B_local + arange(B_local)``.
    ``ddp_average=True`` pre-multiplies gradients by world_size so that DDP's 1/world averaging yields
    exactly the gradient of the global-batch loss.  A group of one rank needs no exchange and is dropped, unless
    ``collectives_at_world1=True`` (or VITA_CLIP_COLLECTIVES_AT_WORLD1=1): then the all-gather, the reduce-scatter
    and the loss all-reduce run over that one rank as well - the same calls, shapes and dtypes as at world > 1."""

    def __init__(self, temperature: float, lambda_0: float = 0.5, gather_distributed: bool = False, group=None,
                 ddp_average: bool = True, collectives_at_world1: bool = False) -> None:
        super().__init__()
        self.temperature = temperature
        if lambda_0 > 1 or lambda_0 < 0:
            raise ValueError('lambda_0 must be a float between 0 and 1.')
        self.lambda_0 = lambda_0
        self.lambda_1 = 1 - lambda_0
        self.gather_distributed = gather_distributed
        self.group = group
        self.ddp_average = ddp_average
        self.collectives_at_world1 = bool(collectives_at_world1)

    def forward(self, out0: torch.Tensor, out1: torch.Tensor) -> Tuple:
        _require_cuda(out0)
        group = self.group
        if group is None and self.gather_distributed and dist.is_available() and dist.is_initialized():
            group = dist.group.WORLD
        world = dist.get_world_size(group) if group is not None else 1
        keep1 = self.collectives_at_world1 or os.environ.get("VITA_CLIP_COLLECTIVES_AT_WORLD1") == "1"
        return _ClipFn.apply(out0, out1, float(self.temperature), float(self.lambda_0),
                             group if (world > 1 or keep1) else None,
                             float(world) if (self.ddp_average and world > 1) else 1.0)


class _ReconMSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, imgs, mask, patch, P, want_pred):
        B, L1, ldz = z.shape
        imgs = imgs.float().contiguous()
        L = L1 - 1
        loss = torch.zeros(1, dtype=torch.float32, device=z.device)
        dz = torch.empty(B * L1, ldz, dtype=torch.bfloat16, device=z.device)
        pred = torch.empty(B, L, P, dtype=torch.float32, device=z.device) if want_pred else None
        if mask is None:
            inv = None
            ops.sigmoid_mse_patch(z.contiguous(), ldz, imgs, patch, None, 1.0 / (B * L * P), loss, pred=pred, dz=dz, ld_dz=ldz)
        else:
            # mask_loss: mean over the masked rows only (utils/losses.py:118-121).  Their number stays on the device -
            # no host sync in the middle of the forward; an all-kept mask gives 0/0 = NaN as the reference's mean of an
            # empty selection does.
            mask = mask.float().contiguous()
            inv = 1.0 / (mask.sum() * P)
            ops.sigmoid_mse_patch(z.contiguous(), ldz, imgs, patch, mask, 1.0, loss, pred=pred, dz=dz, ld_dz=ldz)
            loss = loss * inv
        ctx.save_for_backward(dz, inv)
        ctx.shape = (B, L1, ldz)
        if want_pred:
            ctx.mark_non_differentiable(pred)
            return loss[0], pred
        return loss[0], None

    @staticmethod
    def backward(ctx, gloss, _gp):
        dz, inv = ctx.saved_tensors
        if inv is not None:
            gloss = gloss * inv
        return (dz.float() * gloss).view(ctx.shape), None, None, None, None, None


def recon_mse_from_logits(z, imgs, patch_size, P, mask=None, mask_loss=False, want_pred=False):
    """z: pre-sigmoid head output [B, 1+L, ldz] (row 0 = cls).  Returns (loss, pred|None)."""
    return _ReconMSEFn.apply(z, imgs, mask if mask_loss else None, tuple(int(p) for p in patch_size), int(P), want_pred)
