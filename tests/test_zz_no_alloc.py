"""vita_layernorm_bwd_noalloc - the one entry point that allocates nothing (its dgamma / dbeta partials live in a
file-scope device array of the library) - against vita_layernorm_bwd with a caller-provided partial buffer, through the
C ABI: two calls in a row, the second with a smaller input; the same kernels run, so every output is bit-identical."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(lib, noalloc, dy, x, gamma, mean, rstd, dx_in, rows, d, ld):
    dev = x.device
    dx = torch.full((rows, ld), float("nan"), dtype=torch.float32, device=dev)
    dxb = torch.full((rows, ld), float("nan"), dtype=torch.bfloat16, device=dev)
    dg = torch.full((d,), float("nan"), dtype=torch.float32, device=dev)
    db = torch.full((d,), float("nan"), dtype=torch.float32, device=dev)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    stream = torch.cuda.current_stream().cuda_stream
    head = [vp(dy.data_ptr()), i64(ld), i32(0 if dy.dtype == torch.bfloat16 else 1), vp(x.data_ptr()), i64(ld), vp(gamma.data_ptr()),
            vp(mean.data_ptr()), vp(rstd.data_ptr()), vp(dx_in.data_ptr() if dx_in is not None else None), vp(dx.data_ptr()), i64(ld),
            vp(dxb.data_ptr()), i64(ld), vp(dg.data_ptr()), vp(db.data_ptr())]
    if noalloc:
        rc = lib.vita_layernorm_bwd_noalloc(*head, i64(rows), i64(d), vp(stream))
    else:
        nblk = lib.vita_layernorm_bwd_nblk(rows)
        partial = torch.empty(2 * nblk * d, dtype=torch.float32, device=dev)
        rc = lib.vita_layernorm_bwd(*head, vp(partial.data_ptr()), i64(rows), i64(d), vp(stream))
    torch.cuda.synchronize()
    assert rc == 0, lib.vita_last_error().decode()
    return dx, dxb, dg, db


def test_layernorm_bwd_noalloc_matches_the_allocating_entry_point():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from vita_amd import _lib, ops
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    # first a large call (13 826 rows x 1025: every one of the 768 partial rows is used), then a smaller one (fewer
    # blocks, narrower rows): the second call must not see anything the first one left in the library's scratch
    for rows, d, bf16_dy, with_res in ((13826, 1025, True, True), (300, 129, False, False), (37, 2048, True, True)):
        ld = ops.rup(d, 64)
        x = torch.zeros(rows, ld, device="cuda"); x[:, :d] = torch.randn(rows, d, device="cuda", generator=g)
        dy = torch.zeros(rows, ld, device="cuda"); dy[:, :d] = torch.randn(rows, d, device="cuda", generator=g)
        if bf16_dy:
            dy = dy.to(torch.bfloat16)
        gamma = 1 + 0.1 * torch.randn(d, device="cuda", generator=g)
        mean = x[:, :d].mean(1).contiguous(); rstd = (x[:, :d].var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
        dx_in = torch.randn(rows, ld, device="cuda", generator=g) if with_res else None
        ref = _call(lib, False, dy, x, gamma, mean, rstd, dx_in, rows, d, ld)
        got = _call(lib, True, dy, x, gamma, mean, rstd, dx_in, rows, d, ld)
        for a, b in zip(got, ref):
            assert torch.isfinite(b.float()).all() and torch.equal(a, b)
        # dgamma / dbeta against a plain fp32 restatement as well (the two entry points could be wrong together)
        xh = (x[:, :d] - mean[:, None]) * rstd[:, None]
        torch.testing.assert_close(got[3], dy[:, :d].float().sum(0), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(got[2], (dy[:, :d].float() * xh).sum(0), rtol=1e-4, atol=2e-3)
    # a width beyond the built-in scratch is rejected with a clear error, nothing is launched
    fn = lib.vita_layernorm_bwd_noalloc
    z = torch.zeros(8, device="cuda")
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    p = vp(z.data_ptr())
    rc = fn(p, i64(4096), i32(1), p, i64(4096), p, p, p, None, p, i64(4096), None, i64(0), p, p, i64(4), i64(4096),
            vp(torch.cuda.current_stream().cuda_stream))
    assert rc == -2 and b"2048" in lib.vita_last_error()
