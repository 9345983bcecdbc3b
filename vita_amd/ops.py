"""Thin torch-tensor wrappers over the C ABI (include/vita_hip.h).  PyTorch only provides device
memory and the stream; every computation happens in libvita_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import GemmDesc, check

BF16, F32 = 0, 1
EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_GELU_BWD = 0, 1, 2, 3


def rup(x: int, m: int) -> int:
    return (int(x) + m - 1) // m * m


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


# Optional per-launch timing for bench.py: PROF = [] enables (name, start_event, end_event, algorithmic_work).
PROF = None


class _Timed:
    def __init__(self, name, work):
        self.name, self.work = name, work

    def __enter__(self):
        if PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True); self.e0.record()

    def __exit__(self, *exc):
        if PROF is not None:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            PROF.append((self.name, self.e0, e1, self.work))


def gemm(a, b, c, m, n, k, lda, ldb, ldc, *, a_layout=0, b_layout=0, bias=None, epilogue=EPI_NONE,
         aux_in=None, aux_out=None, ld_aux=0, alpha=1.0, split_k=1, batch=(1, 1),
         a_strides=(0, 0), b_strides=(0, 0), c_strides=(0, 0), row_seg=(0, 0), col_seg=(0, 0),
         a_off=0, b_off=0, c_off=0, flops=None, variant=0):
    """C[m,n] = epi(alpha * A(m,k) B(k,n)); a/b bf16 tensors, c bf16|fp32.  *_off are element offsets.
    flops: algorithmic (un-padded) FLOPs of this launch for the roofline accounting."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    d = GemmDesc()
    d.a = a.data_ptr() + 2 * a_off
    d.b = b.data_ptr() + 2 * b_off
    d.c = c.data_ptr() + c.element_size() * c_off
    d.bias = _p(bias); d.aux_in = _p(aux_in); d.aux_out = _p(aux_out)
    d.m, d.n, d.k = m, n, k
    d.lda, d.ldb, d.ldc, d.ld_aux = lda, ldb, ldc, ld_aux
    d.batch_outer, d.batch_inner = batch
    d.a_stride_outer, d.a_stride_inner = a_strides
    d.b_stride_outer, d.b_stride_inner = b_strides
    d.c_stride_outer, d.c_stride_inner = c_strides
    d.a_layout, d.b_layout = a_layout, b_layout
    d.c_dtype = _dt(c); d.epilogue = epilogue; d.split_k = split_k
    d.row_seg_phys, d.row_seg_log = row_seg
    d.col_seg_phys, d.col_seg_log = col_seg
    d.alpha = alpha
    with _Timed(f"gemm_kernel<{a_layout},{b_layout}>", flops if flops is not None else 2.0 * m * n * k * batch[0] * batch[1]):
        check(_lib.lib().vita_gemm_bf16_ex(C.byref(d), variant, _s()), "vita_gemm_bf16")
    return c


def pack_weight(w: torch.Tensor, rows_phys: int, ld_phys: int, row_seg=(0, 0), col_seg=(0, 0), out=None):
    w = w.detach()
    assert w.dtype == torch.float32 and w.is_contiguous()
    rows_log, cols_log = w.shape[0], w[0].numel()
    if out is None:
        out = torch.empty(rows_phys, ld_phys, dtype=torch.bfloat16, device=w.device)
    check(_lib.lib().vita_pack_weight_bf16(w.data_ptr(), rows_log, cols_log, out.data_ptr(), rows_phys, ld_phys,
                                           row_seg[0], row_seg[1], col_seg[0], col_seg[1], _s()), "pack_weight")
    return out


def pack_vec(v: torch.Tensor, n_phys: int, seg=(0, 0), out=None):
    v = v.detach()
    if out is None:
        out = torch.empty(n_phys, dtype=torch.float32, device=v.device)
    check(_lib.lib().vita_pack_vec_f32(v.data_ptr(), v.numel(), out.data_ptr(), n_phys, seg[0], seg[1], _s()), "pack_vec")
    return out


def unpack_vec(v: torch.Tensor, n_log: int, seg=(0, 0)):
    out = torch.zeros(n_log, dtype=torch.float32, device=v.device)
    check(_lib.lib().vita_unpack_vec_f32(v.data_ptr(), v.numel(), out.data_ptr(), n_log, seg[0], seg[1], _s()), "unpack_vec")
    return out


def layernorm_fwd(x, ldx, gamma, beta, rows, d, ldy, out_dtype=torch.bfloat16, eps=1e-5):
    y = torch.empty(rows, ldy, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.lib().vita_layernorm_fwd(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), ldy, _dt(y),
                                        mean.data_ptr(), rstd.data_ptr(), rows, d, eps, _s()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, lddy, x, ldx, gamma, mean, rstd, rows, d, lddx, dx_in=None, want_bf16=False, ld_dxb=0, acc=None):
    """acc = (dgamma_buf, dbeta_buf): fp32 [d] buffers the weight / bias gradients are ADDED to (None is then returned
    for both) instead of being materialised."""
    dev = x.device
    dx = torch.empty(rows, lddx, dtype=torch.float32, device=dev)
    dxb = torch.empty(rows, ld_dxb, dtype=torch.bfloat16, device=dev) if want_bf16 else None
    nblk = _lib.lib().vita_layernorm_bwd_nblk(rows)
    partial = torch.empty(2 * nblk * d, dtype=torch.float32, device=dev)
    if acc is not None:
        dgamma, dbeta = acc
        assert dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32 and dgamma.numel() == d and dbeta.numel() == d
        assert dgamma.is_contiguous() and dbeta.is_contiguous()
        fn = _lib.lib().vita_layernorm_bwd_acc
    else:
        dgamma = torch.empty(d, dtype=torch.float32, device=dev)
        dbeta = torch.empty(d, dtype=torch.float32, device=dev)
        fn = _lib.lib().vita_layernorm_bwd
    check(fn(dy.data_ptr(), lddy, _dt(dy), x.data_ptr(), ldx, gamma.data_ptr(), mean.data_ptr(),
             rstd.data_ptr(), _p(dx_in), dx.data_ptr(), lddx, _p(dxb), ld_dxb,
             dgamma.data_ptr(), dbeta.data_ptr(), partial.data_ptr(), rows, d, _s()), "layernorm_bwd")
    return (dx, dxb, None, None) if acc is not None else (dx, dxb, dgamma, dbeta)


def argsort_mask(noise: torch.Tensor, len_keep: int):
    n, l = noise.shape
    assert noise.dtype == torch.float32 and noise.is_contiguous()
    ids_shuffle = torch.empty(n, l, dtype=torch.int64, device=noise.device)
    ids_restore = torch.empty(n, l, dtype=torch.int64, device=noise.device)
    mask = torch.empty(n, l, dtype=torch.float32, device=noise.device)
    check(_lib.lib().vita_argsort_mask(noise.data_ptr(), n, l, len_keep, ids_shuffle.data_ptr(), ids_restore.data_ptr(),
                                       mask.data_ptr(), _s()), "argsort_mask")
    return ids_shuffle, ids_restore, mask


def row_gather(x, ldx, l_in, ids, ld_ids, l_out, nb, d, ldo, fill=None, pos=None):
    out = torch.empty(nb, l_out, ldo, dtype=torch.float32, device=x.device)
    check(_lib.lib().vita_row_gather_f32(x.data_ptr(), ldx, l_in, ids.data_ptr(), ld_ids, _p(fill), _p(pos), out.data_ptr(), ldo,
                                         l_out, nb, d, _s()), "row_gather")
    return out


def row_scatter(dout, ldo, l_out, ids, ld_ids, l_in, nb, d, ldx, want_dfill=False):
    dx = torch.zeros(nb, l_in, ldx, dtype=torch.float32, device=dout.device)
    dfill = torch.zeros(d, dtype=torch.float32, device=dout.device) if want_dfill else None
    check(_lib.lib().vita_row_scatter_f32(dout.data_ptr(), ldo, l_out, ids.data_ptr(), ld_ids, dx.data_ptr(), ldx, l_in,
                                          _p(dfill), nb, d, _s()), "row_scatter")
    return dx, dfill


def im2col(img, patch, ids, n_tok, ldo, with_cls_row=False):
    """bf16 patch matrix; with_cls_row: layout [nb][1+n_tok][ldo] with a zero row 0 per batch."""
    nb, S, T, H, W = img.shape
    assert img.dtype == torch.float32 and img.is_contiguous()
    if with_cls_row:
        out = torch.zeros(nb * (n_tok + 1), ldo, dtype=torch.bfloat16, device=img.device)
        bro, r0 = n_tok + 1, 1
    else:
        out = torch.empty(nb * n_tok, ldo, dtype=torch.bfloat16, device=img.device)
        bro, r0 = n_tok, 0
    check(_lib.lib().vita_im2col_bf16(img.data_ptr(), nb, S, T, H, W, patch[0], patch[1], patch[2], _p(ids), n_tok,
                                      out.data_ptr(), ldo, bro, r0, _s()), "im2col")
    return out


def tokens_assemble(y, ldy, cls, pos, ids, nb, n_tok, d, ldo, inplace_with_cls_row=False):
    """inplace_with_cls_row: y is [nb][1+n_tok][ldy] (row 0 = scratch) and is finished in place."""
    if inplace_with_cls_row:
        assert ldy == ldo
        out = y.view(nb, n_tok + 1, ldo)
        check(_lib.lib().vita_tokens_assemble(y.data_ptr() + 4 * ldy, ldy, cls.data_ptr(), _p(pos), _p(ids), nb, n_tok, d,
                                              out.data_ptr(), ldo, n_tok + 1, _s()), "tokens_assemble")
        return out
    out = torch.empty(nb, n_tok + 1, ldo, dtype=torch.float32, device=y.device)
    check(_lib.lib().vita_tokens_assemble(y.data_ptr(), ldy, cls.data_ptr(), _p(pos), _p(ids), nb, n_tok, d,
                                          out.data_ptr(), ldo, n_tok, _s()), "tokens_assemble")
    return out


def cast_bf16(x, ldx, rows, d, ldy):
    y = torch.empty(rows, ldy, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().vita_cast_f32_bf16(x.data_ptr(), ldx, y.data_ptr(), ldy, rows, d, _s()), "cast")
    return y


def colsum(x, ld, rows, cols, out=None, accumulate=False):
    """Column sums; accumulate=True adds them to `out` (fp32 [cols]) instead of overwriting it."""
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    fn = _lib.lib().vita_colsum_acc if accumulate else _lib.lib().vita_colsum
    check(fn(x.data_ptr(), _dt(x), ld, rows, cols, out.data_ptr(), _s()), "colsum")
    return out


def softmax_rows(s, lds, p, ldp, rows, n, scale):
    check(_lib.lib().vita_softmax_rows(s.data_ptr(), lds, p.data_ptr(), ldp, rows, n, scale, _s()), "softmax_rows")


def softmax_bwd_rows(p, ldp, dp, lddp, ds, ldds, rows, n, scale):
    check(_lib.lib().vita_softmax_bwd_rows(p.data_ptr(), ldp, dp.data_ptr(), lddp, ds.data_ptr(), ldds, rows, n, scale, _s()),
          "softmax_bwd_rows")


def l2_normalize_fwd(x, eps=1e-12):
    rows, d = x.shape
    y = torch.empty_like(x); inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.lib().vita_l2_normalize_fwd(x.data_ptr(), rows, d, eps, y.data_ptr(), inv.data_ptr(), _s()), "l2norm_fwd")
    return y, inv


def l2_normalize_bwd(dy, y, inv):
    rows, d = y.shape
    dx = torch.empty_like(y)
    check(_lib.lib().vita_l2_normalize_bwd(dy.data_ptr(), y.data_ptr(), inv.data_ptr(), rows, d, dx.data_ptr(), _s()), "l2norm_bwd")
    return dx


def sgemm(a, b, c, m, n, k, lda, ldb, ldc, a_layout=0, b_layout=0, alpha=1.0, beta=0.0):
    check(_lib.lib().vita_sgemm_f32(a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, lda, ldb, ldc, a_layout, b_layout,
                                    alpha, beta, _s()), "sgemm")
    return c


def ce_rows(logits, ld, rows, cols, label_offset, weight, loss, dlogits=None, row_lse=None):
    check(_lib.lib().vita_ce_rows(logits.data_ptr(), ld, rows, cols, label_offset, weight, loss.data_ptr(), _p(row_lse),
                                  _p(dlogits), ld if dlogits is None else dlogits.stride(0), _s()), "ce_rows")


def sigmoid_mse_patch(z, ldz, img, patch, mask, inv_count, loss, pred=None, dz=None, ld_dz=0):
    nb, S, T, H, W = img.shape
    check(_lib.lib().vita_sigmoid_mse_patch(z.data_ptr(), ldz, img.data_ptr(), nb, S, T, H, W, patch[0], patch[1], patch[2],
                                            _p(mask), inv_count, _p(pred), loss.data_ptr(), _p(dz), ld_dz, _s()), "sigmoid_mse_patch")


def tabular_embed_fwd(x, wptr, bptr, n_class, cls, pos, D, ldo):
    """x fp32 [B, F] -> fp32 [B, 1+F, ldo] (cls row, per-feature tokens, + pos-embed; pads zero)."""
    B, F = x.shape
    out = torch.empty(B, F + 1, ldo, dtype=torch.float32, device=x.device)
    check(_lib.lib().vita_tabular_embed_fwd(x.data_ptr(), B, F, wptr.data_ptr(), bptr.data_ptr(), n_class.data_ptr(),
                                            cls.data_ptr(), _p(pos), out.data_ptr(), ldo, D, _s()), "tabular_embed_fwd")
    return out


def tabular_embed_bwd(dout, ldo, x, n_class, gw, gb, dcls, D):
    B, F = x.shape
    check(_lib.lib().vita_tabular_embed_bwd(dout.data_ptr(), ldo, x.data_ptr(), B, F, n_class.data_ptr(), gw.data_ptr(),
                                            gb.data_ptr(), _p(dcls), D, _s()), "tabular_embed_bwd")


def token_mean_fwd(xp, D):
    B, N, ld = xp.shape
    out = torch.empty(B, D, dtype=torch.float32, device=xp.device)
    check(_lib.lib().vita_token_mean_fwd(xp.data_ptr(), ld, B, N, D, out.data_ptr(), _s()), "token_mean_fwd")
    return out


def token_mean_bwd(dy, B, N, D, ld):
    dx = torch.empty(B, N, ld, dtype=torch.float32, device=dy.device)
    check(_lib.lib().vita_token_mean_bwd(dy.data_ptr(), B, N, D, dx.data_ptr(), ld, _s()), "token_mean_bwd")
    return dx


def batchnorm_fwd(x, gamma, beta, eps, relu, running_mean=None, running_var=None, momentum=0.1):
    B, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(D, dtype=torch.float32, device=x.device); rstd = torch.empty_like(mean)
    check(_lib.lib().vita_batchnorm_fwd(x.data_ptr(), B, D, _p(gamma), _p(beta), eps, int(relu), y.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), _p(running_mean), _p(running_var), momentum, _s()), "batchnorm_fwd")
    return y, mean, rstd


def batchnorm_eval(x, gamma, beta, eps, relu, running_mean, running_var):
    B, D = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().vita_batchnorm_eval(x.data_ptr(), B, D, _p(gamma), _p(beta), eps, int(relu), running_mean.data_ptr(),
                                         running_var.data_ptr(), y.data_ptr(), _s()), "batchnorm_eval")
    return y


def batchnorm_bwd(dy, x, y, gamma, mean, rstd, relu):
    B, D = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.empty(D, dtype=torch.float32, device=x.device); dbeta = torch.empty_like(dgamma)
    check(_lib.lib().vita_batchnorm_bwd(dy.data_ptr(), x.data_ptr(), y.data_ptr(), B, D, _p(gamma), mean.data_ptr(),
                                        rstd.data_ptr(), int(relu), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _s()),
          "batchnorm_bwd")
    return dx, dgamma, dbeta


def add_bias_rows(out, bias):
    rows, D = out.shape
    check(_lib.lib().vita_add_bias_rows(out.data_ptr(), rows, D, bias.data_ptr(), _s()), "add_bias_rows")
    return out


def colsum_small(x):
    rows, D = x.shape
    out = torch.empty(D, dtype=torch.float32, device=x.device)
    check(_lib.lib().vita_colsum_small_f32(x.data_ptr(), rows, D, out.data_ptr(), _s()), "colsum_small")
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    check(_lib.lib().vita_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                                    weight_decay, grad_scale, step, _s()), "adam_step")


def attn_supported(hd, hdp):
    return bool(_lib.lib().vita_attn_supported(hd, hdp))


def attn_fwd(qkv, o, lse, B, N, H, hd, hdp, scale, tail_split=True):
    ws, ws_bytes = None, 0
    if tail_split:          # scratch for the split tail launch (fixed size, reused by the caching allocator)
        ws_bytes = _lib.lib().vita_attn_fwd_ws_bytes(hdp)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device)
    with _Timed("attn_fwd_kernel", 4.0 * B * H * N * N * hd):
        check(_lib.lib().vita_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), _p(ws), ws_bytes, B, N, H, hd, hdp, scale,
                                       _s()), "attn_fwd")


ATTN_DS_HANDOFF = True      # dS tiles through HBM scratch (B*H*N^2*2 B): dQ without the S / dP recompute


def attn_bwd(qkv, o, do, lse, dqkv, B, N, H, hd, hdp, scale, handoff=None):
    delta = torch.empty(B, H, 2, N, dtype=torch.float32, device=qkv.device)   # [lse | rowsum(dO*O)] scratch
    ws, ws_bytes = None, 0
    if ATTN_DS_HANDOFF if handoff is None else handoff:
        ws_bytes = _lib.lib().vita_attn_bwd_ws_bytes(B, N, H, hdp)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device)
    # algorithmic work = the four products the reference's backward performs (dV, dP, dQ, dK: 2x the forward,
    # SURVEY 8(d) "step = 3x fwd"); the S recompute of the fused kernels is overhead, not counted
    with _Timed("attn_bwd_kernels", 8.0 * B * H * N * N * hd):
        check(_lib.lib().vita_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                       dqkv.data_ptr(), _p(ws), ws_bytes, B, N, H, hd, hdp, scale, _s()), "attn_bwd")
