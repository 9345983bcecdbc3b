"""Step driver for the hot path: Stage-I model (encoder + decoder + fused loss) mirroring
models/reconstruction_models.py:105-134 without the Lightning harness, a flat-buffer Adam
(models/reconstruction_models.py:36-38) and a bucketed RCCL gradient all-reduce overlapped with
backward (the explicit counterpart of Lightning's strategy="ddp", main.py:61)."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import ops
from .losses import CLIPLoss, all_reduce_any, device_collectives, recon_mse_from_logits
from .networks import ImagingMaskedDecoder, ImagingMaskedEncoder, SimCLRProjectionHead, TabularViT, token_mean
from . import nn as _vnn
from .nn import bump_shadow_epoch, ensure_side_stream, prefetch_shadows


class ReconMAE(nn.Module):
    """encoder_imaging + decoder_imaging with the reference's state_dict key prefixes."""

    def __init__(self, img_shape, use_both_axes=True, **hparams):
        super().__init__()
        self.hparams_ = dict(hparams)
        self.patch_size = hparams["patch_size"]
        self.mask_loss = bool(hparams.get("mask_loss", False))
        self.encoder_imaging = ImagingMaskedEncoder(img_shape=img_shape, use_both_axes=use_both_axes, **hparams)
        self.decoder_imaging = ImagingMaskedDecoder(num_patches=self.encoder_imaging.patch_embed.num_patches,
                                                    grid_size=self.encoder_imaging.patch_embed.grid_size,
                                                    use_both_axes=use_both_axes,
                                                    head_out_dim=self.encoder_imaging.patch_p_num, **hparams)

    def forward(self, x):
        x, mask, ids_restore = self.encoder_imaging(x)
        x = self.decoder_imaging(x, ids_restore)
        return x, mask

    def training_loss(self, imgs, want_pred=False):
        """models/reconstruction_models.py:128-134: forward, patchify target, MSE (fused)."""
        xp, mask, ids_restore = self.encoder_imaging.forward_padded(imgs)
        D = self.encoder_imaging.enc_embed_dim
        z = self.decoder_imaging.forward_logits_padded(xp[..., :D], ids_restore)
        loss, pred = recon_mse_from_logits(z, imgs, self.patch_size, int(self.encoder_imaging.patch_p_num), mask,
                                           self.mask_loss, want_pred)
        return loss, pred, mask


class MultimodalSimCLR(nn.Module):
    """Stage-II model (models/MultimodalSimCLR.py:32-64, 117-154) without the Lightning harness: imaging tower
    (masked encoder, activation checkpointing) and tabular tower, mean over all tokens (avg_token=True), SimCLR
    projection heads, CLIPLoss (optionally with cross-GPU negatives).  Same sub-module / state_dict names."""

    def __init__(self, img_shape, imaging_hparams: dict, tabular_hparams, all_feature_names: dict, temperature=0.1,
                 lambda_0=0.5, avg_token=True, use_both_axes=True, gather_distributed=False):
        super().__init__()
        self.avg_token = avg_token
        self.encoder_imaging = ImagingMaskedEncoder(img_shape=img_shape, use_both_axes=use_both_axes, **imaging_hparams)
        D = imaging_hparams["enc_embed_dim"]
        self.projector_imaging = SimCLRProjectionHead(D, D, imaging_hparams["projection_dim"])
        self.encoder_tabular = TabularViT(tabular_hparams, all_feature_names=all_feature_names,
                                          patch_num=self.encoder_imaging.patch_embed.num_patches)
        self.projector_tabular = SimCLRProjectionHead(tabular_hparams.embedding_dim, tabular_hparams.embedding_dim,
                                                      tabular_hparams.projection_dim)
        self.criterion = CLIPLoss(temperature=temperature, lambda_0=lambda_0, gather_distributed=gather_distributed)

    def forward_imaging(self, x):
        yp, _, _ = self.encoder_imaging.forward_padded(x)
        D = self.encoder_imaging.enc_embed_dim
        y = yp[..., :D]
        x_img = token_mean(yp, D) if self.avg_token else y[:, 0, :].contiguous()
        return self.projector_imaging(x_img), x_img, y

    def forward_tabular(self, x):
        yp = self.encoder_tabular.forward_padded(x)
        D = self.encoder_tabular.embedding_dim
        x_tab = token_mean(yp, D) if self.avg_token else yp[:, 0, :D].contiguous()
        return self.projector_tabular(x_tab), x_tab

    def training_loss(self, imgs, tabs):
        """models/MultimodalSimCLR.py:150-154 -> (loss, logits, labels)."""
        z_img, _, _ = self.forward_imaging(imgs)
        z_tab, _ = self.forward_tabular(tabs)
        return self.criterion(z_img, z_tab)


class FlatAdam:
    """torch.optim.Adam semantics on one flat fp32 buffer (params and grads become views of it).

    Restriction (differs from torch.optim.Adam, which skips a parameter whose .grad is None): the kernel runs over the
    whole buffer every step, so a parameter that received no gradient still sees weight decay and moment decay, and
    the step counter is shared.  Build the optimizer from the parameters that train (requires_grad is honoured at
    construction) and do not freeze a parameter afterwards; every shipped Stage-I / Stage-II graph uses every
    trainable parameter in every step."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        sizes = [ops.rup(p.numel(), 4) for p in self.params]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(self.offsets[-1])
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets[:-1]):
            view = self.p[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.g[o:o + p.numel()].view(p.shape)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        bump_shadow_epoch()
        if dev.type == "cuda" and _vnn.WGRAD_SIDE_STREAM:
            ensure_side_stream(dev)        # host-blocking stream probe: here, not in the first backward

    def zero_grad(self):
        self.g.zero_()

    def step(self, grad_scale=1.0, lr=None):
        self.t += 1
        ops.adam_step(self.p, self.g, self.m, self.v, self.lr if lr is None else lr, self.betas[0], self.betas[1],
                      self.eps, self.wd, grad_scale, self.t)
        bump_shadow_epoch()
        prefetch_shadows(self.p.device)


class BucketedAllReduce:
    """Gradient all-reduce over RCCL in ~bucket_mb buckets of the flat gradient buffer, launched from
    post-accumulate hooks as soon as every parameter of a bucket has its gradient (reverse order =
    backward order).  The collective runs on the process group's own stream, ordered after what the
    current stream holds at the launch, so it overlaps the rest of backward; no stream of our own in
    between - HIP streams that share a hardware queue block one another at event waits, and such a
    relay stream on the weight-gradient stream's queue cost that stream its overlap with the attention
    kernels."""

    def __init__(self, opt: FlatAdam, group=None, bucket_mb=48):
        self.opt, self.group = opt, group
        self.world = dist.get_world_size(group)
        self.cuda = opt.g.is_cuda
        # RCCL reduces device memory asynchronously on its own stream.  Every other backend (gloo) gets host tensors:
        # a bucket is staged through the host and reduced synchronously at its launch point (all_reduce_any)
        self.async_device = self.cuda and device_collectives(group)
        if self.cuda:
            if self.async_device:
                # create the communicator and RCCL's stream now (first step timing, and so that the stream-concurrency
                # probe below sees every stream that will exist)
                dist.all_reduce(torch.zeros(1, dtype=torch.float32, device=opt.g.device), group=group)
            # the probe blocks the host: run it here, before any bucket can be in flight - never inside a backward
            if _vnn.WGRAD_SIDE_STREAM:
                ensure_side_stream(opt.g.device)
        cap = bucket_mb * (1 << 20) // 4
        self.buckets = []           # (start, end, n_params)
        self.bucket_of = {}
        end = int(opt.offsets[-1]); cnt = 0; hi = end
        for i in range(len(opt.params) - 1, -1, -1):
            lo = int(opt.offsets[i]); cnt += 1
            self.bucket_of[i] = len(self.buckets)
            if hi - lo >= cap or i == 0:
                self.buckets.append((lo, hi, cnt)); hi = lo; cnt = 0
        self.pending = [0] * len(self.buckets)
        self.fired = {}             # parameter indices that reported since the last finish()
        self.handles = []
        self.launch_order = []
        from . import nn as vnn
        for i, p in enumerate(opt.params):
            hook = self._make_hook(i)
            p.register_post_accumulate_grad_hook(hook)
            vnn.GRAD_READY[id(p)] = hook          # gradients accumulated in place by the kernels (no autograd accumulate)

    def _make_hook(self, i):
        def hook(_p):
            # A parameter reports once per step.  Gradients accumulated in place by the kernels report through
            # vita_amd.nn.GRAD_READY right after their launch; autograd (torch 2.10) then ALSO runs the post-accumulate
            # hook for the None it got back - the second report must not launch the bucket again (a second all-reduce
            # would multiply the gradient by the world size).  No parameter of these models is used twice in one graph.
            if i in self.fired:
                return
            b = self.bucket_of[i]
            self.fired[i] = 1
            self.pending[b] += 1
            if self.pending[b] == self.buckets[b][2]:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self.launch_order.append(b)
        if self.async_device or not self.cuda:
            self.handles.append(dist.all_reduce(self.opt.g[lo:hi], group=self.group, async_op=True))
        else:
            all_reduce_any(self.opt.g[lo:hi], self.group)

    def finish(self):
        """Wait for every bucket (launching any whose hooks did not all fire) and reset."""
        for b, n in enumerate(self.pending):
            if n != self.buckets[b][2]:
                self._launch(b)
        for h in self.handles:
            h.wait()                # RCCL: the current stream waits for the collective; gloo (host tensors): the host does
        self.handles = []
        self.pending = [0] * len(self.buckets)
        self.fired = {}


def warmup_cosine_lr(base_lr, step, num_warmup_steps, num_training_steps, num_cycles=0.5):
    """lr at scheduler step `step` under transformers.get_cosine_schedule_with_warmup (the reference's "anneal"
    scheduler, models/MultimodalSimCLR.py:107-112; Lightning steps it once per epoch, so step = epoch)."""
    import math
    if step < num_warmup_steps:
        return base_lr * float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def cosine_annealing_lr(base_lr, epoch, t_max, eta_min=0.0):
    """Closed form of torch CosineAnnealingLR (the reference's "cosine" scheduler, models/MultimodalSimCLR.py:104-106)."""
    import math
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max)) / 2.0


def stage1_train_step(model: ReconMAE, opt: FlatAdam, imgs, reducer: Optional[BucketedAllReduce] = None):
    opt.zero_grad()
    loss, _, _ = model.training_loss(imgs)
    loss.backward()
    if reducer is not None:
        reducer.finish()
        opt.step(grad_scale=1.0 / reducer.world)
    else:
        opt.step()
    return loss


def stage2_train_step(model: MultimodalSimCLR, opt: FlatAdam, imgs, tabs, reducer: Optional[BucketedAllReduce] = None,
                      lr=None):
    """One Stage-II step (models/MultimodalSimCLR.py:146-178 minus logging): Adam with coupled L2 weight decay,
    lr from the caller's warmup-cosine schedule.  With a reducer (DDP), CLIPLoss(ddp_average=True) pre-scales its
    gradient by world so that the 1/world averaging below yields the gradient of the global-batch loss."""
    opt.zero_grad()
    loss, logits, labels = model.training_loss(imgs, tabs)
    loss.backward()
    if reducer is not None:
        reducer.finish()
        opt.step(grad_scale=1.0 / reducer.world, lr=lr)
    else:
        opt.step(lr=lr)
    return loss, logits, labels
