"""Host-side mirror of the transformer pieces on the hot path: a timm-0.6.12-compatible ``Block``
(call sites networks/imaging_encoders.py:36-40, networks/imaging_decoders.py:43-45) whose forward and
backward run entirely in libvita_hip.so.

Data layout in HBM:
  * residual stream  : fp32 [B, N, Dp]   Dp = round_up(D, 64), pad columns are always zero
  * GEMM operands    : bf16, leading dims multiples of 8; q/k/v and the attention output are stored
                       head-padded: head h occupies columns [h*hdp, h*hdp+hd), hdp = round_up(hd, 32)
  * weights          : fp32 nn.Parameters with the reference shapes (state_dict compatible) plus
                       padded bf16 "shadows" refreshed whenever the parameter changes.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Optional

import torch
from torch import nn

from . import _lib, ops
from ._lib import check

_SHADOW_EPOCH = 0
DEFAULT_FLASH = True    # fused attention; the materialised path remains for unsupported head sizes and as a cross-check


def bump_shadow_epoch():
    """Call after parameters were modified outside torch's version tracking (our Adam kernel)."""
    global _SHADOW_EPOCH
    _SHADOW_EPOCH += 1


class Dims:
    def __init__(self, dim: int, num_heads: int, hidden: int):
        assert dim % num_heads == 0
        self.D, self.H, self.hd, self.hidden = dim, num_heads, dim // num_heads, hidden
        self.Dp = ops.rup(dim, 64)
        self.hdp = ops.rup(self.hd, 32)
        self.AO = self.H * self.hdp            # attention-output / q / k / v width (head padded)
        self.QKV = 3 * self.AO
        self.Hp = ops.rup(hidden, 64)
        self.scale = self.hd ** -0.5


class _Shadows:
    """bf16 padded copies of fp32 parameters, keyed by (version, epoch, data_ptr).

    After an optimizer step every copy is stale.  Left alone they are rebuilt one by one when the forward first asks for
    them - ~75 small HBM-bound launches on the main stream's critical path (0.76 ms per Stage-I step).  prefetch_shadows()
    (called by FlatAdam.step) rebuilds the copies of every live holder on the second stream instead, in creation (=
    forward) order with one event per holder, and get() makes the consumer's stream wait for its holder's event only: the
    copies of later blocks are made beside the MFMA-bound kernels of the earlier ones."""

    _live = weakref.WeakSet()

    def __init__(self):
        self._c = {}
        self._build = {}          # name -> (param, build): what prefetch re-runs
        self._ready = None        # event of a prefetch no consumer has waited for yet
        _Shadows._live.add(self)
        self._order = _Shadows._count = getattr(_Shadows, "_count", 0) + 1

    def get(self, name, param, build):
        key = (param._version, _SHADOW_EPOCH, param.data_ptr())
        hit = self._c.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, build(param))
            self._c[name] = hit
            self._build[name] = (weakref.ref(param), build)
        elif self._ready is not None:
            torch.cuda.current_stream(param.device).wait_event(self._ready)
            self._ready = None
        return hit[1]

    def refresh(self):
        """Rebuild every known copy on the current stream; False when a parameter has gone."""
        with torch.no_grad():
            for name, (pref, build) in list(self._build.items()):
                param = pref()
                if param is None or not param.is_cuda:
                    return False
                key = (param._version, _SHADOW_EPOCH, param.data_ptr())
                hit = self._c.get(name)
                if hit is None or hit[0] != key:
                    self._c[name] = (key, build(param))
        return True


SHADOW_PREFETCH = os.environ.get("VITA_SHADOW_PREFETCH", "1") != "0"


def prefetch_shadows(device):
    """Rebuild all stale bf16 copies on the second stream (see _Shadows); the caller has just changed the parameters on
    the current stream of `device`."""
    if not (SHADOW_PREFETCH and WGRAD_SIDE_STREAM) or device.type != "cuda":
        return
    holders = sorted((h for h in _Shadows._live if h._build), key=lambda h: h._order)
    if not holders:
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream(idx)
    side = ensure_side_stream(device)
    side.wait_stream(main)          # after the optimizer step - and after every earlier use of the copies being replaced
    with torch.cuda.stream(side):
        for h in holders:
            if h.refresh():
                ev = torch.cuda.Event()
                ev.record(side)
                h._ready = ev


def pad_tokens(x: torch.Tensor, Dp: int) -> torch.Tensor:
    """[B,N,D] fp32 -> contiguous [B,N,Dp] with zero pads (plumbing at the drop-in boundary)."""
    B, N, D = x.shape
    xp = torch.zeros(B, N, Dp, dtype=torch.float32, device=x.device)
    xp[..., :D] = x
    return xp


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop1 = nn.Identity()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Identity()


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Identity()


# ----------------------------------------------------------------------------- attention (bf16 MFMA)
def attention_fwd(qkv, B, N, dm: Dims, use_flash: bool):
    """qkv bf16 [B*N, QKV] -> (o bf16 [B*N, AO], saved)."""
    dev = qkv.device
    H, hdp, AO, QKV = dm.H, dm.hdp, dm.AO, dm.QKV
    o = torch.empty(B * N, AO, dtype=torch.bfloat16, device=dev)
    if use_flash and ops.attn_supported(dm.hd, hdp):
        lse = torch.empty(B, H, N, dtype=torch.float32, device=dev)
        ops.attn_fwd(qkv, o, lse, B, N, H, dm.hd, hdp, dm.scale)
        return o, ("flash", lse)
    Nld = ops.rup(N, 8)
    s = torch.empty(B, H, N, Nld, dtype=torch.float32, device=dev)
    ops.gemm(qkv, qkv, s, N, N, hdp, QKV, QKV, Nld, batch=(B, H), a_strides=(N * QKV, hdp),
             b_strides=(N * QKV, hdp), c_strides=(H * N * Nld, N * Nld), b_off=AO, flops=2.0 * B * H * N * N * dm.hd)
    p = torch.empty(B, H, N, Nld, dtype=torch.bfloat16, device=dev)
    ops.softmax_rows(s, Nld, p, Nld, B * H * N, N, dm.scale)
    del s
    ops.gemm(p, qkv, o, N, hdp, N, Nld, QKV, AO, b_layout=1, batch=(B, H), a_strides=(H * N * Nld, N * Nld),
             b_strides=(N * QKV, hdp), c_strides=(N * AO, hdp), b_off=2 * AO, flops=2.0 * B * H * N * N * dm.hd)
    return o, ("mat", p)


def attention_bwd(do, qkv, o, saved, B, N, dm: Dims):
    """do bf16 [B*N, AO] -> dqkv bf16 [B*N, QKV] (pads zero)."""
    dev = qkv.device
    H, hdp, AO, QKV = dm.H, dm.hdp, dm.AO, dm.QKV
    dqkv = torch.empty(B * N, QKV, dtype=torch.bfloat16, device=dev)
    if saved[0] == "flash":
        ops.attn_bwd(qkv, o, do, saved[1], dqkv, B, N, H, dm.hd, hdp, dm.scale)
        return dqkv
    p = saved[1]
    Nld = p.shape[-1]
    bs_qkv, bs_p, bs_o = (N * QKV, hdp), (H * N * Nld, N * Nld), (N * AO, hdp)
    fl = 2.0 * B * H * N * N * dm.hd
    dp = torch.empty(B, H, N, Nld, dtype=torch.float32, device=dev)
    # dP = dO V^T
    ops.gemm(do, qkv, dp, N, N, hdp, AO, QKV, Nld, batch=(B, H), a_strides=bs_o, b_strides=bs_qkv, c_strides=bs_p,
             b_off=2 * AO, flops=fl)
    # dV = P^T dO
    ops.gemm(p, do, dqkv, N, hdp, N, Nld, AO, QKV, a_layout=1, b_layout=1, batch=(B, H), a_strides=bs_p,
             b_strides=bs_o, c_strides=bs_qkv, c_off=2 * AO, flops=fl)
    ds = torch.empty(B, H, N, Nld, dtype=torch.bfloat16, device=dev)
    ops.softmax_bwd_rows(p, Nld, dp, Nld, ds, Nld, B * H * N, N, dm.scale)
    del dp
    # dQ = dS K ; dK = dS^T Q
    ops.gemm(ds, qkv, dqkv, N, hdp, N, Nld, QKV, QKV, b_layout=1, batch=(B, H), a_strides=bs_p, b_strides=bs_qkv,
             c_strides=bs_qkv, b_off=AO, flops=fl)
    ops.gemm(ds, qkv, dqkv, N, hdp, N, Nld, QKV, QKV, a_layout=1, b_layout=1, batch=(B, H), a_strides=bs_p,
             b_strides=bs_qkv, c_strides=bs_qkv, c_off=AO, flops=fl)
    return dqkv


def _wgrad_plan(m, n, k):
    """(variant, split_k) for a weight gradient dW[m, n] = dY^T X over k tokens.

    Large problems run on the 256-row LDS-DMA kernel: tile width (16 * variant) and split-K factor are chosen
    together by a small cost model - rounds of 256 workgroups x K-tiles per split x tile width, plus the fp32
    atomic traffic of the split (about 1.3 TB/s chip-wide, half of it assumed hidden under the MFMAs)."""
    kt = (k + 63) // 64
    if m < 512 or n < 208 or kt < 8:
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        return 0, max(1, min(8, (2 * 256 + tiles - 1) // tiles, kt // 4 if kt >= 4 else 1))
    tm = (m + 255) // 256
    best = None
    for nf in (13, 14, 16):
        tn = (n + 16 * nf - 1) // (16 * nf)
        for s in range(1, 17):
            per = -(-kt // s)
            if s > 1 and per < 4:
                break
            splits = -(-kt // per)
            rounds = -(-(tm * tn * splits) // 256)
            mfma_us = rounds * (per + 3) * (nf + 2) * 0.134
            atomic_us = 0.5 * splits * m * n * 4 / 1.3e6 if splits > 1 else 0.0
            cost = mfma_us + atomic_us
            if best is None or cost < best[0]:
                best = (cost, nf, splits)
    return best[1], best[2]


INPLACE_PARAM_GRADS = os.environ.get("VITA_INPLACE_GRADS", "1") != "0"   # accumulate weight / bias gradients straight into an existing fp32 .grad (FlatAdam's flat buffer)
GRAD_READY = {}              # id(param) -> callable(param): what a post-accumulate-grad hook would have been told


def grad_target(param):
    """The tensor a parameter gradient may be accumulated into directly (fp32 atomics of the split-K GEMM / colsum),
    instead of materialising it and letting autograd run `grad += it`; None when there is no such buffer."""
    g = param.grad if (INPLACE_PARAM_GRADS and param is not None and param.requires_grad) else None
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != param.shape or not g.is_cuda:
        return None
    return g


_DEFERRED_DONE = None         # list while a Block backward has launches in flight on its side stream


LN_INPLACE_GRADS = os.environ.get("VITA_LN_INPLACE", "1") != "0"


def ln_grad_targets(ln, need_w, need_b):
    """(weight.grad, bias.grad) of a LayerNorm when BOTH can take their gradient in place (vita_layernorm_bwd_acc), else None:
    autograd's `grad += dgamma` is a 5 us launch plus a launch gap per vector on the main stream, four per Block."""
    if not (need_w and need_b and LN_INPLACE_GRADS):
        return None
    gw, gb = grad_target(ln.weight), grad_target(ln.bias)
    return (gw, gb) if gw is not None and gb is not None else None


def grad_done(param):
    """After an in-place accumulation: run the hook autograd would have run (bucketed all-reduce bookkeeping).
    While a side stream holds the accumulating launches the report waits for the join (_SideStream.join)."""
    if _DEFERRED_DONE is not None:
        _DEFERRED_DONE.append(param)
        return
    hook = GRAD_READY.get(id(param))
    if hook is not None:
        hook(param)


WGRAD_SIDE_STREAM = os.environ.get("VITA_WGRAD_STREAM", "1") != "0"
_SIDE_STREAMS = {}            # device index -> the second HIP stream of that device


_STREAM_PROBE = {}            # device index -> (candidates tried, ms of the winning pair of 0.2 ms spins) for diagnostics


def _concurrent_stream(idx, main, tries=12, spin_us=200):
    """A stream whose kernels really run beside `main`'s.  HIP multiplexes its streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) and two streams on one queue serialise: with RCCL's and the reducer's streams
    created first, the next pool stream landed on the main stream's queue and the two-stream backward silently lost its
    overlap (+2.5 ms per step).  So candidates are tested: a 0.2 ms one-wave spin on each of the two streams takes
    0.2 ms when they overlap and 0.4 ms when they do not."""
    cands = []
    best = None
    for _ in range(tries):
        c = torch.cuda.Stream(device=idx)
        if any(c == o for o in cands):
            break                                   # the pool wrapped around
        cands.append(c)
        c.wait_stream(main)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(main)
        check(_lib.lib().vita_spin_us(spin_us, main.cuda_stream), "spin")
        check(_lib.lib().vita_spin_us(spin_us, c.cuda_stream), "spin")
        main.wait_stream(c)
        e1.record(main)
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        if best is None or ms < best[1]:
            best = (c, ms)
        if ms < 1.5e-3 * spin_us:
            break
    _STREAM_PROBE[idx] = (len(cands), best[1])
    return best[0]


def ensure_side_stream(device, probe=True):
    """The device's second stream, chosen (once) by the concurrency probe above.

    The probe blocks the host on an event, so it must run where nothing else can be waiting for this thread: from the
    constructors of FlatAdam / BucketedAllReduce and from the first Block FORWARD (main thread, no collective of the
    step in flight yet) - never from a Block backward.  Round 2 probed lazily inside the first _BlockFn.backward, i.e.
    on autograd's device thread after the reconstruction head's hooks could already have launched the first gradient
    bucket: gpurun_out/r2_share2.log shows one rank blocked there (e1.synchronize) while its peer waited for it inside
    the collective.  With probe=False (what a backward falls back to if no forward has run on this device) a plain
    pool stream is taken without any host wait."""
    idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = _concurrent_stream(idx, torch.cuda.current_stream(idx)) if probe else torch.cuda.Stream(device=idx)
        _SIDE_STREAMS[idx] = s
    return s


class _SideStream:
    """Weight / bias gradient launches of one Block backward on a second HIP stream.

    dW = dY^T X and colsum(dY) depend only on dY, never on the dgrad chain that follows, so they run beside it: the
    256-row GEMM grids are 1-4 rounds of one workgroup per CU, and a second independent grid fills the CUs a
    partial last round leaves idle and takes the workgroups of the two kernels out of lock-step (their epilogue
    store bursts no longer coincide).  fork() orders the side stream after everything issued so far on the main
    stream; join() makes the main stream wait for the side stream and only then reports the accumulated gradients
    (the bucketed all-reduce orders its communication stream after the main stream).  Every tensor the side
    stream touches stays referenced by the caller until join(), so the caching allocator cannot hand its memory
    to a main-stream allocation while the side stream still uses it."""

    def __init__(self, device, enabled=True):
        self.on = bool(enabled and WGRAD_SIDE_STREAM)
        if not self.on:
            return
        global _DEFERRED_DONE
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.main = torch.cuda.current_stream(idx)
        self.side = ensure_side_stream(device, probe=False)       # already chosen by the forward; never probes here
        self.prev = None          # Block backwards never nest; a list left behind by a failed backward is dropped
        _DEFERRED_DONE = []

    def __call__(self, fn, *args, **kw):
        """Run fn(*args) on the side stream, after what the main stream holds now."""
        if not self.on:
            return fn(*args, **kw)
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            return fn(*args, **kw)

    def join(self):
        if not self.on:
            return
        global _DEFERRED_DONE
        self.main.wait_stream(self.side)
        done, _DEFERRED_DONE = _DEFERRED_DONE, self.prev
        for p in done:
            grad_done(p)


def wgrad(dy, lddy, x, ldx, rows_out, cols_out, k_tokens, row_seg=(0, 0), col_seg=(0, 0), m_phys=None, n_phys=None, param=None):
    """dW[logical rows_out, cols_out] fp32 = dy^T x over k_tokens rows (both operands k-strided).
    With `param` whose .grad is a usable fp32 buffer the product is ADDED to it (split-K atomics) and None is returned."""
    m = m_phys if m_phys is not None else rows_out
    n = n_phys if n_phys is not None else cols_out
    variant, sk = _wgrad_plan(m, n, k_tokens)
    tgt = grad_target(param)
    if tgt is not None:
        sk = max(sk, 2)          # the atomic (accumulating) epilogue
        out = tgt.view(rows_out, cols_out)
    else:
        out = (torch.zeros if sk > 1 else torch.empty)(rows_out, cols_out, dtype=torch.float32, device=dy.device)
    ops.gemm(dy, x, out, m, n, k_tokens, lddy, ldx, cols_out, a_layout=1, b_layout=1, split_k=sk, variant=variant,
             row_seg=row_seg, col_seg=col_seg, flops=2.0 * rows_out * cols_out * k_tokens)
    if tgt is not None:
        grad_done(param)
        return None
    return out


def bgrad(dy, ld, rows, cols, param=None):
    """Bias gradient = column sums of dy; accumulated into param.grad when possible (returns None then)."""
    tgt = grad_target(param)
    if tgt is not None:
        ops.colsum(dy, ld, rows, cols, out=tgt, accumulate=True)
        grad_done(param)
        return None
    return ops.colsum(dy, ld, rows, cols)


def with_bf16_twin(g, gb):
    """Attach the bf16 copy a kernel wrote beside an fp32 gradient (LayerNorm backward emits both in one pass).  The
    next backward node gets the very tensor object its producer returned when the two are chained directly, and then
    skips its own fp32 -> bf16 cast pass.  Anything in between either hands over a different object without the
    attribute (a checkpoint boundary, an out-of-place accumulation) or changes the values of this one IN PLACE
    (autograd's input buffer may add a second consumer's gradient with `add_` - a skip-connection latent that also feeds
    a decoder): the tensor's version counter is recorded with the copy, and a copy whose version no longer matches is
    ignored, so the consumer casts the accumulated fp32 values as before."""
    g._vita_bf16 = (gb, g._version)
    return g


def bf16_twin(g, rows, ld):
    twin = getattr(g, "_vita_bf16", None)
    if twin is None:
        return None
    gb, version = twin
    if g._version != version or not g.is_contiguous() or tuple(gb.shape) != (rows, ld) or gb.device != g.device:
        return None
    return gb


# ----------------------------------------------------------------------------- Block
class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, blk, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b):
        dm: Dims = blk.dims
        B, N, Dp = xp.shape
        M = B * N
        if WGRAD_SIDE_STREAM:
            ensure_side_stream(xp.device)           # a dictionary hit after the first call (see its docstring)
        sh = blk.shadows()
        x = xp.reshape(M, Dp)
        ln1, mean1, rstd1 = ops.layernorm_fwd(x, Dp, n1w, n1b, M, dm.D, Dp, eps=blk.norm1.eps)
        qkv = torch.empty(M, dm.QKV, dtype=torch.bfloat16, device=x.device)
        ops.gemm(ln1, sh["qkv_w"], qkv, M, dm.QKV, Dp, Dp, Dp, dm.QKV, bias=sh["qkv_b"], flops=6.0 * M * dm.D * dm.D)
        o, att_saved = attention_fwd(qkv, B, N, dm, DEFAULT_FLASH if blk.use_flash is None else blk.use_flash)
        x1 = torch.empty(M, Dp, dtype=torch.float32, device=x.device)
        ops.gemm(o, sh["proj_w"], x1, M, Dp, dm.AO, dm.AO, dm.AO, Dp, bias=sh["proj_b"], epilogue=ops.EPI_RESIDUAL,
                 aux_in=x, ld_aux=Dp, flops=2.0 * M * dm.D * dm.D)
        ln2, mean2, rstd2 = ops.layernorm_fwd(x1, Dp, n2w, n2b, M, dm.D, Dp, eps=blk.norm2.eps)
        u = torch.empty(M, dm.Hp, dtype=torch.bfloat16, device=x.device)   # GELU'(pre-activation), saved for backward
        g = torch.empty(M, dm.Hp, dtype=torch.bfloat16, device=x.device)
        ops.gemm(ln2, sh["fc1_w"], g, M, dm.Hp, Dp, Dp, Dp, dm.Hp, bias=sh["fc1_b"], epilogue=ops.EPI_GELU, aux_out=u,
                 flops=2.0 * M * dm.D * dm.hidden)
        x2 = torch.empty(M, Dp, dtype=torch.float32, device=x.device)
        ops.gemm(g, sh["fc2_w"], x2, M, Dp, dm.Hp, dm.Hp, dm.Hp, Dp, bias=sh["fc2_b"], epilogue=ops.EPI_RESIDUAL,
                 aux_in=x1, ld_aux=Dp, flops=2.0 * M * dm.D * dm.hidden)
        ctx.blk = blk
        ctx.shape = (B, N)
        # the attention's saved tensor (LSE, or the probabilities of the materialised path) goes through
        # save_for_backward like the rest, so that activation checkpointing releases it with them
        ctx.att_mode = att_saved[0]
        ctx.save_for_backward(x, mean1, rstd1, ln1, qkv, o, x1, mean2, rstd2, ln2, u, g, n1w, n2w, att_saved[1])
        return x2.view(B, N, Dp)

    @staticmethod
    def backward(ctx, dx2):
        blk = ctx.blk
        dm: Dims = blk.dims
        B, N = ctx.shape
        M, Dp, D = B * N, dm.Dp, dm.D
        x, mean1, rstd1, ln1, qkv, o, x1, mean2, rstd2, ln2, u, g, n1w, n2w, att_t = ctx.saved_tensors
        sh = blk.shadows()
        dyb = bf16_twin(dx2, M, Dp)
        dx2 = dx2.contiguous().view(M, Dp)
        if dyb is None:
            dyb = ops.cast_bf16(dx2, Dp, M, D, Dp)
        side = _SideStream(x.device)
        # forward argument order: xp, blk, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b.  A frozen
        # parameter (load_pretrained_weights(freeze_encoder=True)) gets no weight / bias gradient launch at all.
        need = ctx.needs_input_grad
        # ---- MLP (weight / bias gradients beside the dgrad chain, see _SideStream)
        a, mlp = blk.attn, blk.mlp
        d_fc2w = side(wgrad, dyb, Dp, g, dm.Hp, D, dm.hidden, M, param=mlp.fc2.weight) if need[12] else None
        d_fc2b = side(bgrad, dyb, Dp, M, D, param=mlp.fc2.bias) if need[13] else None
        du = torch.empty(M, dm.Hp, dtype=torch.bfloat16, device=x.device)   # GELU'(pre-activation), saved for backward
        ops.gemm(dyb, sh["fc2_w"], du, M, dm.Hp, Dp, Dp, dm.Hp, dm.Hp, b_layout=1, epilogue=ops.EPI_GELU_BWD, aux_in=u,
                 ld_aux=dm.Hp, flops=2.0 * M * D * dm.hidden)
        d_fc1w = side(wgrad, du, dm.Hp, ln2, Dp, dm.hidden, D, M, param=mlp.fc1.weight) if need[10] else None
        d_fc1b = side(bgrad, du, dm.Hp, M, dm.hidden, param=mlp.fc1.bias) if need[11] else None
        dln2 = torch.empty(M, Dp, dtype=torch.bfloat16, device=x.device)
        ops.gemm(du, sh["fc1_w"], dln2, M, Dp, dm.Hp, dm.Hp, Dp, Dp, b_layout=1, flops=2.0 * M * D * dm.hidden)
        if not side.on:
            del du
        acc2 = ln_grad_targets(blk.norm2, need[8], need[9])
        dx1, dx1b, d_n2w, d_n2b = ops.layernorm_bwd(dln2, Dp, x1, Dp, n2w, mean2, rstd2, M, D, Dp, dx_in=dx2,
                                                    want_bf16=True, ld_dxb=Dp, acc=acc2)
        if acc2 is not None:
            grad_done(blk.norm2.weight); grad_done(blk.norm2.bias)
        # ---- attention
        d_projw = side(wgrad, dx1b, Dp, o, dm.AO, D, D, M, col_seg=(dm.hdp, dm.hd), n_phys=dm.AO, param=a.proj.weight) \
            if need[6] else None
        d_projb = side(bgrad, dx1b, Dp, M, D, param=a.proj.bias) if need[7] else None
        do = torch.empty(M, dm.AO, dtype=torch.bfloat16, device=x.device)
        ops.gemm(dx1b, sh["proj_w"], do, M, dm.AO, Dp, Dp, dm.AO, dm.AO, b_layout=1, flops=2.0 * M * D * D)
        dqkv = attention_bwd(do, qkv, o, (ctx.att_mode, att_t), B, N, dm)
        d_qkvw = side(wgrad, dqkv, dm.QKV, ln1, Dp, 3 * D, D, M, row_seg=(dm.hdp, dm.hd), m_phys=dm.QKV, param=a.qkv.weight) \
            if need[4] else None
        d_qkvb = None
        if blk.attn.qkv.bias is not None and need[5]:
            d_qkvb = side(lambda: ops.unpack_vec(ops.colsum(dqkv, dm.QKV, M, dm.QKV), 3 * D, seg=(dm.hdp, dm.hd)))
        dln1 = torch.empty(M, Dp, dtype=torch.bfloat16, device=x.device)
        ops.gemm(dqkv, sh["qkv_w"], dln1, M, Dp, dm.QKV, dm.QKV, Dp, Dp, b_layout=1, flops=6.0 * M * D * D)
        acc1 = ln_grad_targets(blk.norm1, need[2], need[3])
        dx, dxb, d_n1w, d_n1b = ops.layernorm_bwd(dln1, Dp, x, Dp, n1w, mean1, rstd1, M, D, Dp, dx_in=dx1, want_bf16=True,
                                                  ld_dxb=Dp, acc=acc1)
        if acc1 is not None:
            grad_done(blk.norm1.weight); grad_done(blk.norm1.bias)
        side.join()
        dx = with_bf16_twin(dx.view(B, N, Dp), dxb)
        return (dx, None, d_n1w, d_n1b, d_qkvw, d_qkvb, d_projw, d_projb, d_n2w, d_n2b, d_fc1w, d_fc1b,
                d_fc2w, d_fc2b)


class Block(nn.Module):
    """Drop-in for ``timm.models.vision_transformer.Block`` (0.6.12): same constructor, attribute and
    state_dict key names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2).  Only the configuration
    the reference uses is implemented (no dropout / layer-scale / drop-path, exact-erf GELU, LayerNorm)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., init_values=None,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if drop or attn_drop or drop_path or init_values is not None:
            raise NotImplementedError("vita_amd.Block: dropout / layer-scale / drop-path are not used by ViTa")
        if act_layer is not nn.GELU or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("vita_amd.Block: only nn.GELU + nn.LayerNorm (the reference configuration)")
        hidden = int(dim * mlp_ratio)
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.ls1 = nn.Identity(); self.drop_path1 = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, hidden)
        self.ls2 = nn.Identity(); self.drop_path2 = nn.Identity()
        self.dims = Dims(dim, num_heads, hidden)
        self.use_flash = None   # None -> module default (DEFAULT_FLASH)
        self._sh = _Shadows()

    def shadows(self):
        dm, g = self.dims, self._sh.get
        seg = (dm.hdp, dm.hd)
        a, m = self.attn, self.mlp
        dev = a.proj.weight.device
        zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        return {
            "qkv_w": g("qkv_w", a.qkv.weight, lambda p: ops.pack_weight(p, dm.QKV, dm.Dp, row_seg=seg)),
            "qkv_b": g("qkv_b", a.qkv.bias, lambda p: ops.pack_vec(p, dm.QKV, seg=seg)) if a.qkv.bias is not None
            else g("qkv_b0", a.qkv.weight, lambda p: zeros(dm.QKV)),
            "proj_w": g("proj_w", a.proj.weight, lambda p: ops.pack_weight(p, dm.Dp, dm.AO, col_seg=seg)),
            "proj_b": g("proj_b", a.proj.bias, lambda p: ops.pack_vec(p, dm.Dp)),
            "fc1_w": g("fc1_w", m.fc1.weight, lambda p: ops.pack_weight(p, dm.Hp, dm.Dp)),
            "fc1_b": g("fc1_b", m.fc1.bias, lambda p: ops.pack_vec(p, dm.Hp)),
            "fc2_w": g("fc2_w", m.fc2.weight, lambda p: ops.pack_weight(p, dm.Dp, dm.Hp)),
            "fc2_b": g("fc2_b", m.fc2.bias, lambda p: ops.pack_vec(p, dm.Dp)),
        }

    def forward_padded(self, xp: torch.Tensor) -> torch.Tensor:
        """xp fp32 [B,N,Dp] contiguous with zero pads -> same layout."""
        a, m = self.attn, self.mlp
        return _BlockFn.apply(xp, self, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight,
                              a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight,
                              m.fc2.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_cuda(x)
        D = self.dims.D
        return self.forward_padded(pad_tokens(x.float(), self.dims.Dp))[..., :D]


def _require_cuda(x):
    if not x.is_cuda:
        raise RuntimeError("vita_amd modules run only on an MI355X (cuda device); there is no CPU fallback")
    if x.device.index != torch.cuda.current_device():
        # launches go to the CURRENT device's current stream (ops._s): one process per GPU, or set the device first
        raise RuntimeError(f"vita_amd: tensor on cuda:{x.device.index} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device() (one process per GPU)")


# ----------------------------------------------------------------------------- LayerNorm / Linear on the padded stream
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, w, b, eps, D, ln):
        B, N, Dp = xp.shape
        x = xp.reshape(B * N, Dp)
        y, mean, rstd = ops.layernorm_fwd(x, Dp, w, b, B * N, D, Dp, out_dtype=torch.float32, eps=eps)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.D = D
        ctx.ln = ln
        return y.view(B, N, Dp)

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        B, N, Dp = dy.shape
        dy = dy.contiguous().view(B * N, Dp)
        acc = ln_grad_targets(ctx.ln, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        dx, dxb, dw, db = ops.layernorm_bwd(dy, Dp, x, Dp, w, mean, rstd, B * N, ctx.D, Dp, want_bf16=True, ld_dxb=Dp, acc=acc)
        if acc is not None:
            grad_done(ctx.ln.weight); grad_done(ctx.ln.bias)
        return with_bf16_twin(dx.view(B, N, Dp), dxb), dw, db, None, None, None


def layer_norm_padded(xp, ln: nn.LayerNorm):
    return _LayerNormFn.apply(xp, ln.weight, ln.bias, ln.eps, ln.normalized_shape[0], ln)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the padded fp32 stream: [B,N,Kp] -> [B,N,Np] (bf16 MFMA, fp32 out)."""

    @staticmethod
    def forward(ctx, xp, w, b, holder, Np):
        B, N, Kp = xp.shape
        M = B * N
        out_f, in_f = w.shape
        ws = holder.get("w", w, lambda p: ops.pack_weight(p, Np, Kp))
        bs = holder.get("b", b, lambda p: ops.pack_vec(p, Np))
        xb = ops.cast_bf16(xp.reshape(M, Kp), Kp, M, in_f, Kp)
        y = torch.empty(M, Np, dtype=torch.float32, device=xp.device)
        ops.gemm(xb, ws, y, M, Np, Kp, Kp, Kp, Np, bias=bs, flops=2.0 * M * out_f * in_f)
        ctx.save_for_backward(xb, ws)
        ctx.lin_w, ctx.lin_b = w, b
        ctx.dims = (B, N, Kp, Np, out_f, in_f)
        return y.view(B, N, Np)

    @staticmethod
    def backward(ctx, dy):
        xb, ws = ctx.saved_tensors
        B, N, Kp, Np, out_f, in_f = ctx.dims
        M = B * N
        dy = dy.contiguous().view(M, Np)
        dyb = ops.cast_bf16(dy, Np, M, out_f, Np)
        need = ctx.needs_input_grad            # (xp, w, b, holder, Np)
        dw = wgrad(dyb, Np, xb, Kp, out_f, in_f, M, param=ctx.lin_w) if need[1] else None
        db = bgrad(dyb, Np, M, out_f, param=ctx.lin_b) if need[2] else None
        dx = None
        if need[0]:
            dx = torch.empty(M, Kp, dtype=torch.float32, device=dy.device)
            ops.gemm(dyb, ws, dx, M, Kp, Np, Np, Kp, Kp, b_layout=1, flops=2.0 * M * out_f * in_f)
            dx = dx.view(B, N, Kp)
        return dx, dw, db, None, None


def linear_padded(xp, lin: nn.Linear, holder: _Shadows, Np: int):
    return _LinearFn.apply(xp, lin.weight, lin.bias, holder, Np)
