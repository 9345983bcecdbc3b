"""Input-pipeline contract of the hot path and the stage-to-stage checkpoint adapter.

``prepare_cine_batch`` is the device-side mirror of ``Cardiac3DplusTAllAX.__getitem__``
(datasets/datasets.py:403-427) for a batch of subjects whose stored npz arrays are already in HBM: LAX/SAX concat,
the x=y mirror of ``load_im_seg_arr`` (datasets/datasets.py:169-171), ``image_normalization`` (utils/general.py:139-147),
``[:, ::t_downsampling_ratio]`` and the ``frame_to_keep`` subsampling, in two HBM-bound kernels (csrc/pipeline.hip).
Augmentations (torchvision transforms, datasets/datasets.py:184-196) are out of scope.

``load_pretrained_weights`` follows initializers/initialize_model.py:106-139 (key-prefix filtering, ``strict=False``,
optional freezing).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import check


def frame_indices(n_stored_frames: int, time_frame: int = 50, frame_to_keep: Optional[int] = None,
                  t_downsampling_ratio: int = 1) -> List[int]:
    """Stored-frame index of every output frame: ``im[:, ::ratio]`` followed by the ``round(i * step)`` pick
    (datasets/datasets.py:419-424; Python's round = half to even, on the double ``i * step``)."""
    frames = list(range(0, n_stored_frames, t_downsampling_ratio))
    if frame_to_keep is not None and frame_to_keep != time_frame:
        step = (time_frame - 1) / (frame_to_keep - 1)
        frames = [frames[round(i * step)] for i in range(frame_to_keep)]     # IndexError as numpy/torch indexing would
    return frames


def sax_slice_window(seg_sax: torch.Tensor, z_num: int):
    """(z_min, z_max) of load_im_seg_arr's ``z_num`` branch (datasets/datasets.py:162-165): the window starts two
    slices above the first SAX slice whose frame-0 segmentation holds label 1, clamped to the stack.
    seg_sax: integer tensor [H, W, S, T] (host or device)."""
    has = (seg_sax[..., 0] == 1).any(0).any(0)
    z_start = int(torch.argmax(has.to(torch.int32)).item()) + 2
    z_max = min(z_start + z_num, seg_sax.shape[-2])
    return z_max - z_num, z_max


def prepare_cine_batch(lax: torch.Tensor, sax: torch.Tensor, *, z0: int = 0, sax_slice_num: Optional[int] = None,
                       time_frame: int = 50, frame_to_keep: Optional[int] = None, t_downsampling_ratio: int = 1,
                       mirror: bool = True, return_minmax: bool = False):
    """lax fp32 [B, H, W, 3, T], sax fp32 [B, H, W, S_tot, T] on the GPU (npz layout)  ->  imgs fp32 [B, S, T', H, W]
    in [0, 1], LAX slices first, ready for ``ImagingMaskedEncoder``."""
    if not (lax.is_cuda and sax.is_cuda):
        raise RuntimeError("vita_amd.data.prepare_cine_batch runs only on an MI355X (cuda tensors); there is no CPU fallback")
    assert lax.dtype == torch.float32 and sax.dtype == torch.float32 and lax.is_contiguous() and sax.is_contiguous()
    assert lax.dim() == 5 and sax.dim() == 5 and lax.shape[:3] == sax.shape[:3] and lax.shape[4] == sax.shape[4]
    B, A0, A1, s_lax, T = lax.shape
    s_tot = sax.shape[3]
    s_sax = s_tot - z0 if sax_slice_num is None else sax_slice_num
    if not (0 <= z0 and s_sax > 0 and z0 + s_sax <= s_tot):
        raise ValueError(f"SAX slice window [{z0}, {z0 + s_sax}) is outside the stored stack of {s_tot} slices")
    frames = frame_indices(T, time_frame, frame_to_keep, t_downsampling_ratio)
    if not frames or min(frames) < 0 or max(frames) >= T:
        raise ValueError(f"frame indices {frames} outside the {T} stored frames")
    lib = _lib.lib()
    dev = lax.device
    fr = torch.tensor(frames, dtype=torch.int32, device=dev)
    partial = torch.empty(lib.vita_cine_partial_floats(B), dtype=torch.float32, device=dev)
    minmax = torch.empty(B, 2, dtype=torch.float32, device=dev)
    R, Cc = (A1, A0) if mirror else (A0, A1)
    out = torch.empty(B, s_lax + s_sax, len(frames), R, Cc, dtype=torch.float32, device=dev)
    check(lib.vita_cine_prepare(lax.data_ptr(), sax.data_ptr(), B, A0, A1, s_lax, s_tot, z0, s_sax, T, fr.data_ptr(), len(frames),
                                int(mirror), partial.data_ptr(), minmax.data_ptr(), out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream), "vita_cine_prepare")
    return (out, minmax) if return_minmax else out


def filter_state_dict(state_dict: dict, loading_parts: Optional[Sequence[str]] = None,
                      exclude_parts: Optional[Sequence[str]] = None) -> dict:
    """Key-prefix filter of initializers/initialize_model.py:118-128."""
    if loading_parts is None:
        return dict(state_dict)
    keep = {}
    for k, v in state_dict.items():
        if k.startswith(tuple(loading_parts)) and not (exclude_parts is not None and k.startswith(tuple(exclude_parts))):
            keep[k] = v
    return keep


def load_pretrained_weights(model: torch.nn.Module, ckpt_path, freeze_encoder: bool = False,
                            loading_parts: Optional[Sequence[str]] = None, exclude_parts: Optional[Sequence[str]] = None,
                            weights_only: bool = True):
    """initializers/initialize_model.py:106-139: load the ``state_dict`` of a (Lightning) checkpoint, keep the keys under
    ``loading_parts`` minus ``exclude_parts`` (e.g. ``encoder_imaging`` without ``encoder_imaging.enc_pos_embed``), load
    with ``strict=False`` and optionally freeze the loaded parts.  The reference unpickles with ``weights_only=False``;
    here the safe loader is the default (a Lightning checkpoint whose extra entries it refuses needs weights_only=False,
    which executes code from the file - only for files you trust)."""
    ckpt = torch.load(ckpt_path, weights_only=weights_only, map_location="cpu")
    sd = filter_state_dict(ckpt["state_dict"], loading_parts, exclude_parts)
    result = model.load_state_dict(sd, strict=False)
    from . import nn as vnn
    vnn.bump_shadow_epoch()          # bf16 weight shadows are rebuilt on the next forward
    if freeze_encoder and loading_parts is not None:
        for name, p in model.named_parameters():
            if name.startswith(tuple(loading_parts)):
                p.requires_grad = False
    return result
