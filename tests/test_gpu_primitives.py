"""GPU parity of the individual C-ABI kernels against the oracle / plain torch fp32 on CPU.
Tolerances: integer/index work bit-exact; fp32 kernels rtol 1e-4; bf16 MFMA kernels rtol 2e-2
(BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vita_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from vita_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def bf(t):
    return t.to(torch.bfloat16)


def _gelu_grad(x):
    x = x.detach().clone().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    return x.grad


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("la,lb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (300, 200, 136), (1000, 1088, 1088), (77, 40, 8)])
def test_gemm_layouts(ops, la, lb, m, n, k):
    torch.manual_seed(m + n + k + la * 2 + lb)
    A = torch.randn(m, k); B = torch.randn(k, n)
    Ab, Bb = bf(A), bf(B)
    ref = Ab.float() @ Bb.float()
    r8 = lambda x: (x + 7) // 8 * 8
    if la == 0:
        a = torch.zeros(m, r8(k), dtype=torch.bfloat16); a[:, :k] = Ab; lda = r8(k)
    else:
        a = torch.randn(k, r8(m)).to(torch.bfloat16); a[:, :m] = Ab.t(); lda = r8(m)
    if lb == 0:
        b = torch.zeros(n, r8(k), dtype=torch.bfloat16); b[:, :k] = Bb.t(); ldb = r8(k)
    else:
        b = torch.randn(k, r8(n)).to(torch.bfloat16); b[:, :n] = Bb; ldb = r8(n)
    ldc = r8(n)
    c = torch.full((m, ldc), float("nan"), device="cuda")
    ops.gemm(dev(a), dev(b), c, m, r8(n) if False else n - n % 4 if n % 4 else n, k, lda, ldb, ldc, a_layout=la, b_layout=lb)
    nn = n - n % 4 if n % 4 else n
    torch.cuda.synchronize()
    out = c.cpu()[:, :nn]
    torch.testing.assert_close(out, ref[:, :nn], rtol=1e-3, atol=1e-2)
    assert _rel(out, ref[:, :nn]) < 1e-5 * 50


def test_gemm_asymmetric_identity(ops):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    m = n = k = 128
    a = torch.eye(m).to(torch.bfloat16)
    Bm = (torch.arange(k)[:, None] * 2 + torch.arange(n)[None, :] % 7).float() % 61
    b = Bm.t().contiguous().to(torch.bfloat16)    # [N][K]
    c = torch.empty(m, n, device="cuda")
    ops.gemm(dev(a), dev(b), c, m, n, k, k, k, n)
    assert torch.equal(c.cpu(), Bm)


def test_gemm_epilogues_and_batch(ops):
    torch.manual_seed(0)
    m, n, k = 200, 264, 192
    A = bf(torch.randn(m, k)); W = bf(torch.randn(n, k) * 0.1); bias = torch.randn(n)
    pre = A.float() @ W.float().t() + bias
    # bias + GELU (pre-activation saved)
    c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); u = torch.empty_like(c)
    ops.gemm(dev(A), dev(W), c, m, n, k, k, k, n, bias=dev(bias), epilogue=ops.EPI_GELU, aux_out=u)
    torch.testing.assert_close(u.cpu().float(), _gelu_grad(pre), rtol=2e-2, atol=2e-2)     # saved GELU'(pre)
    torch.testing.assert_close(c.cpu().float(), torch.nn.functional.gelu(pre), rtol=2e-2, atol=2e-2)
    # bias + residual, fp32 out
    res = torch.randn(m, n)
    c2 = torch.empty(m, n, device="cuda")
    ops.gemm(dev(A), dev(W), c2, m, n, k, k, k, n, bias=dev(bias), epilogue=ops.EPI_RESIDUAL, aux_in=dev(res), ld_aux=n)
    torch.testing.assert_close(c2.cpu(), pre + res, rtol=1e-3, atol=1e-3)
    # GELU backward multiply by the saved derivative
    uu = bf(_gelu_grad(torch.randn(m, n)))
    c3 = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    ops.gemm(dev(A), dev(W), c3, m, n, k, k, k, n, epilogue=ops.EPI_GELU_BWD, aux_in=dev(uu), ld_aux=n)
    torch.testing.assert_close(c3.cpu().float(), (A.float() @ W.float().t()) * uu.float(), rtol=2e-2, atol=2e-2)
    # split-k atomics + segment maps (weight-grad un-padding): rows 8->5 per segment
    dY = bf(torch.randn(k, 24)); X = bf(torch.randn(k, 40))
    full = dY.float().t() @ X.float()            # [24][40]
    keep_rows = [r for r in range(24) if r % 8 < 5]
    c4 = torch.zeros(15, 40, device="cuda")
    ops.gemm(dev(dY), dev(X), c4, 24, 40, k, 24, 40, 40, a_layout=1, b_layout=1, split_k=3, row_seg=(8, 5))
    torch.testing.assert_close(c4.cpu(), full[keep_rows], rtol=1e-3, atol=1e-3)
    # two-level batch
    Bo, Bi, mm, nn, kk = 2, 3, 70, 72, 64
    Aq = bf(torch.randn(Bo, mm, Bi, kk)); Bk = bf(torch.randn(Bo, nn, Bi, kk))
    cb = torch.empty(Bo, Bi, mm, nn, device="cuda")
    ops.gemm(dev(Aq), dev(Bk), cb, mm, nn, kk, Bi * kk, Bi * kk, nn, batch=(Bo, Bi), a_strides=(mm * Bi * kk, kk),
             b_strides=(nn * Bi * kk, kk), c_strides=(Bi * mm * nn, mm * nn), alpha=0.5)
    ref = 0.5 * torch.einsum("omik,onik->oimn", Aq.float(), Bk.float())
    torch.testing.assert_close(cb.cpu(), ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("rows,d", [(37, 129), (300, 1025), (5, 8)])
def test_layernorm(ops, rows, d):
    torch.manual_seed(d)
    ld = ops.rup(d, 64)
    x = torch.zeros(rows, ld); x[:, :d] = torch.randn(rows, d) * 2 + 0.5
    g = torch.randn(d); b = torch.randn(d)
    xr = x[:, :d].clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = O.layer_norm(xr, gr, br)
    y, mean, rstd = ops.layernorm_fwd(dev(x), ld, dev(g), dev(b), rows, d, ld, out_dtype=torch.float32)
    torch.testing.assert_close(y.cpu()[:, :d], ref.detach(), rtol=1e-4, atol=1e-5)
    assert not y.cpu()[:, d:].any()
    yb, _, _ = ops.layernorm_fwd(dev(x), ld, dev(g), dev(b), rows, d, ld)
    torch.testing.assert_close(yb.cpu().float()[:, :d], ref.detach(), rtol=2e-2, atol=2e-2)
    dy = torch.zeros(rows, ld); dy[:, :d] = torch.randn(rows, d)
    ref.backward(dy[:, :d])
    dxin = torch.zeros(rows, ld); dxin[:, :d] = torch.randn(rows, d)
    dx, dxb, dg, db = ops.layernorm_bwd(dev(dy), ld, dev(x), ld, dev(g), mean, rstd, rows, d, ld, dx_in=dev(dxin), want_bf16=True, ld_dxb=ld)
    torch.testing.assert_close(dx.cpu()[:, :d], xr.grad + dxin[:, :d], rtol=1e-4, atol=1e-4)
    assert not dx.cpu()[:, d:].any()
    torch.testing.assert_close(dxb.cpu().float(), dx.cpu(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)
    # accumulating form (vita_layernorm_bwd_acc): the same sums, bit for bit, ADDED to what the buffers hold
    accg = torch.full((d,), 0.25, device="cuda"); accb = torch.full((d,), -1.5, device="cuda")
    dx2, _, none_g, none_b = ops.layernorm_bwd(dev(dy), ld, dev(x), ld, dev(g), mean, rstd, rows, d, ld, dx_in=dev(dxin),
                                               want_bf16=True, ld_dxb=ld, acc=(accg, accb))
    assert none_g is None and none_b is None and torch.equal(dx2, dx)
    assert torch.equal(accg, 0.25 + dg) and torch.equal(accb, -1.5 + db)


def test_argsort_mask_bit_exact(ops, golden):
    g = golden("g3_masker.npz")
    for key in [k[len("noise_"):] for k in g.files if k.startswith("noise_")]:
        noise = g[f"noise_{key}"]; ratio = int(key.split("_r")[1]) / 10
        mask, restore, keep = O.random_masking(noise, ratio)
        sh, rs, mk = ops.argsort_mask(dev(torch.from_numpy(noise)), keep.shape[1])
        assert np.array_equal(rs.cpu().numpy(), restore)
        assert np.array_equal(mk.cpu().numpy(), mask)
        assert np.array_equal(sh.cpu().numpy()[:, :keep.shape[1]], keep)
    # forced ties (stable rule) and a ragged / tiny row
    mask, restore, keep = O.random_masking(g["ties_noise"], 0.5)
    sh, rs, mk = ops.argsort_mask(dev(torch.from_numpy(g["ties_noise"])), 8)
    assert np.array_equal(rs.cpu().numpy(), restore) and np.array_equal(mk.cpu().numpy(), mask)
    noise = np.random.RandomState(0).rand(3, 1).astype(np.float32)
    sh, rs, mk = ops.argsort_mask(dev(torch.from_numpy(noise)), 0)
    assert not rs.cpu().numpy().any() and mk.cpu().numpy().all()
    noise = np.random.RandomState(1).rand(5, 777).astype(np.float32); noise[:, ::3] = 0.25
    mask, restore, keep = O.random_masking(noise, 0.7)
    sh, rs, mk = ops.argsort_mask(dev(torch.from_numpy(noise)), keep.shape[1])
    assert np.array_equal(rs.cpu().numpy(), restore) and np.array_equal(mk.cpu().numpy(), mask)


def test_gather_scatter_bit_exact(ops, golden):
    g = golden("g4_gather.npz")
    x = torch.from_numpy(g["x"]); nb, L, d = x.shape; ld = 20
    xp = torch.zeros(nb, L, ld); xp[..., :d] = x
    mask, restore, keep = O.random_masking(g["noise"], 0.7)
    ids = dev(torch.from_numpy(np.ascontiguousarray(keep)))
    out = ops.row_gather(dev(xp), ld, L, ids, keep.shape[1], keep.shape[1], nb, d, ld)
    assert torch.equal(out.cpu()[..., :d], torch.from_numpy(g["xm"])) and not out.cpu()[..., d:].any()
    cot = torch.zeros(nb, keep.shape[1], ld); cot[..., :d] = torch.from_numpy(g["cot"])
    dx, _ = ops.row_scatter(dev(cot), ld, keep.shape[1], ids, keep.shape[1], L, nb, d, ld)
    assert torch.equal(dx.cpu()[..., :d], torch.from_numpy(g["dx"]))
    # decoder un-shuffle with mask-token fill (ids >= l_in)
    fill = torch.randn(d)
    rs = dev(torch.from_numpy(restore))
    k = keep.shape[1]
    out2 = ops.row_gather(out, ld, k, rs, L, L, nb, d, ld, fill=dev(fill))
    shuf = torch.cat([torch.from_numpy(g["xm"]), fill.expand(nb, L - k, d)], 1)
    ref = O.gather_rows(shuf, torch.from_numpy(restore))
    assert torch.equal(out2.cpu()[..., :d], ref)
    dxx, dfill = ops.row_scatter(out2, ld, L, rs, L, k, nb, d, ld, want_dfill=True)
    assert torch.equal(dxx.cpu()[..., :d], torch.from_numpy(g["xm"]))
    torch.testing.assert_close(dfill.cpu(), fill * nb * (L - k), rtol=1e-5, atol=1e-5)


def test_im2col_and_assemble(ops, golden):
    g = golden("g6_patchify.npz")
    x = torch.from_numpy(g["x"]); patches = torch.from_numpy(g["patches"])
    nb, L, P = patches.shape
    out = ops.im2col(dev(x), (5, 8, 8), None, L, 320)
    assert torch.equal(out.cpu().float().reshape(nb, L, 320), patches.to(torch.bfloat16).float())
    ids = torch.stack([torch.randperm(L)[:20] for _ in range(nb)])
    out = ops.im2col(dev(x), (5, 8, 8), dev(ids), 20, 328)
    ref = O.gather_rows(patches, ids).to(torch.bfloat16).float()
    assert torch.equal(out.cpu().float().reshape(nb, 20, 328)[..., :320], ref) and not out.cpu().float()[..., 320:].any()
    d, ld = 17, 24
    y = torch.randn(nb * 20, ld); cls = torch.randn(d); pos = torch.randn(1 + L, d)
    o = ops.tokens_assemble(dev(y), ld, dev(cls), dev(pos), dev(ids), nb, 20, d, ld).cpu()
    assert torch.equal(o[:, 0, :d], (cls + pos[0]).expand(nb, d))
    assert torch.equal(o[:, 1:, :d], y.reshape(nb, 20, ld)[..., :d] + pos[1 + ids])
    assert not o[..., d:].any()


def test_cast_colsum_softmax(ops):
    torch.manual_seed(3)
    rows, d, ld = 130, 1025, 1088
    x = torch.randn(rows, ld)
    y = ops.cast_bf16(dev(x), ld, rows, d, ld).cpu()
    assert torch.equal(y[:, :d], x[:, :d].to(torch.bfloat16)) and not y[:, d:].float().any()
    cs = ops.colsum(dev(y), ld, rows, d).cpu()
    torch.testing.assert_close(cs, y[:, :d].float().sum(0), rtol=1e-4, atol=1e-4)
    cs = ops.colsum(dev(x), ld, rows, d).cpu()
    torch.testing.assert_close(cs, x[:, :d].sum(0), rtol=1e-4, atol=1e-4)
    n, ldp = 333, 336
    s = torch.randn(50, n) * 3
    p = torch.empty(50, ldp, dtype=torch.bfloat16, device="cuda")
    ops.softmax_rows(dev(s), n, p, ldp, 50, n, 0.7)
    ref = torch.softmax(s * 0.7, -1)
    torch.testing.assert_close(p.cpu().float()[:, :n], ref, rtol=1e-2, atol=1e-4)
    assert not p.cpu().float()[:, n:].any()
    dp = torch.randn(50, n)
    ds = torch.empty(50, ldp, dtype=torch.bfloat16, device="cuda")
    ops.softmax_bwd_rows(p, ldp, dev(dp), n, ds, ldp, 50, n, 0.7)
    pf = p.cpu().float()[:, :n]
    refds = pf * (dp - (pf * dp).sum(-1, keepdim=True)) * 0.7
    torch.testing.assert_close(ds.cpu().float()[:, :n], refds, rtol=2e-2, atol=1e-4)


def test_pack_weight_and_adam(ops):
    torch.manual_seed(4)
    H, hd, hdp, D = 5, 205, 224, 1025
    w = torch.randn(3 * D, D)
    sh = ops.pack_weight(dev(w), 3 * H * hdp, 1088, row_seg=(hdp, hd)).cpu().float()
    wv = w.to(torch.bfloat16).float().reshape(3 * H, hd, D)
    shv = sh.reshape(3 * H, hdp, 1088)
    assert torch.equal(shv[:, :hd, :D], wv) and not shv[:, hd:].any() and not shv[..., D:].any()
    b = torch.randn(3 * D)
    bs = ops.pack_vec(dev(b), 3 * H * hdp, seg=(hdp, hd))
    assert torch.equal(ops.unpack_vec(bs, 3 * D, seg=(hdp, hd)).cpu(), b)
    n = 10007
    p = torch.randn(n); g = torch.randn(n); st = {}
    pd, gd, md, vd = dev(p), dev(g), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ref = {"w": p.clone()}
    for step in (1, 2, 3):
        ops.adam_step(pd, gd, md, vd, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, step)
        O.adam_step(ref, {"w": g}, st, 1e-3)
    torch.testing.assert_close(pd.cpu(), ref["w"], rtol=1e-5, atol=1e-6)


def test_clip_pieces_and_mse(ops, golden):
    g = golden("g5_clip.npz")
    a = torch.from_numpy(g["a_B256_l3"]); b = torch.from_numpy(g["b_B256_l3"])
    an, ia = ops.l2_normalize_fwd(dev(a)); bn, ib = ops.l2_normalize_fwd(dev(b))
    torch.testing.assert_close(an.cpu(), torch.nn.functional.normalize(a, dim=1), rtol=1e-5, atol=1e-6)
    B = 256
    logits = torch.empty(B, B, device="cuda")
    ops.sgemm(an, bn, logits, B, B, 128, 128, 128, B, 0, 0, alpha=10.0)
    torch.testing.assert_close(logits.cpu()[0], torch.from_numpy(g["logits_row0_B256_l3"]), rtol=1e-4, atol=1e-4)
    loss = torch.zeros(1, device="cuda"); dl = torch.empty(B, B, device="cuda")
    ops.ce_rows(logits, B, B, B, 0, 0.3 / B, loss, dlogits=dl)
    lt = logits.t().contiguous(); dlt = torch.empty(B, B, device="cuda")
    ops.ce_rows(lt, B, B, B, 0, 0.7 / B, loss, dlogits=dlt)
    torch.testing.assert_close(loss.cpu()[0], torch.from_numpy(g["loss_B256_l3"]), rtol=1e-4, atol=1e-5)
    # mse + sigmoid
    g6 = golden("g6_patchify.npz")
    x = torch.from_numpy(g6["x"]); tgt = torch.from_numpy(g6["patches"]); nb, L, P = tgt.shape
    z = torch.randn(nb, 1 + L, 328)
    zr = z[:, 1:, :P].clone().requires_grad_(True)
    ref = O.reconstruction_mse(torch.sigmoid(zr), tgt); ref.backward()
    loss = torch.zeros(1, device="cuda"); dz = torch.empty(nb * (1 + L), 328, dtype=torch.bfloat16, device="cuda")
    pred = torch.empty(nb, L, P, device="cuda")
    ops.sigmoid_mse_patch(dev(z), 328, dev(x), (5, 8, 8), None, 1.0 / (nb * L * P), loss, pred=pred, dz=dz, ld_dz=328)
    torch.testing.assert_close(loss.cpu()[0], ref.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(pred.cpu(), torch.sigmoid(zr).detach(), rtol=1e-4, atol=1e-6)
    dzc = dz.cpu().float().reshape(nb, 1 + L, 328)
    torch.testing.assert_close(dzc[:, 1:, :P], zr.grad, rtol=2e-2, atol=1e-8)
    assert not dzc[:, 0].any() and not dzc[..., P:].any()
    mask = torch.from_numpy(g6["mask"])
    loss.zero_()
    ops.sigmoid_mse_patch(dev(z), 328, dev(x), (5, 8, 8), dev(mask), 1.0 / (mask.sum().item() * P), loss)
    torch.testing.assert_close(loss.cpu()[0], O.reconstruction_mse(torch.sigmoid(zr.detach()), tgt, mask, True), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,N,H,hd", [(2, 197, 5, 205), (1, 64, 3, 43), (2, 333, 2, 128), (1, 1000, 5, 205), (1, 31, 1, 32),
                                      (2, 1, 2, 205), (1, 33, 1, 43)])
def test_flash_attention_fwd(ops, B, N, H, hd):
    torch.manual_seed(N)
    hdp = ops.rup(hd, 32)
    assert ops.attn_supported(hd, hdp)
    qkv = torch.zeros(B, N, 3, H, hdp)
    qkv[..., :hd] = torch.randn(B, N, 3, H, hd)
    qkv[:, min(3, N - 1), 1, 0, :hd] *= 4.0     # one spiky key row: forces a late running-max jump (online-softmax rescale)
    qb = qkv.to(torch.bfloat16)
    q, k, v = [qb[:, :, i, :, :hd].float().permute(0, 2, 1, 3) for i in range(3)]
    scale = hd ** -0.5
    s = (q @ k.transpose(-1, -2)) * scale
    ref = torch.softmax(s, -1) @ v
    lse_ref = torch.logsumexp(s, -1)
    o = torch.full((B, N, H, hdp), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    ops.attn_fwd(qb.cuda(), o, lse, B, N, H, hd, hdp, scale)
    oc = o.cpu().float()
    assert not oc[..., hd:].any()
    torch.testing.assert_close(oc[..., :hd].permute(0, 2, 1, 3), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse.cpu(), lse_ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("handoff", [True, False], ids=["ds_handoff", "dq_recompute"])
@pytest.mark.parametrize("B,N,H,hd", [(2, 197, 5, 205), (1, 64, 3, 43), (2, 333, 2, 128), (1, 700, 5, 205), (1, 31, 1, 32),
                                      (1, 512, 2, 205), (2, 257, 1, 205), (1, 129, 3, 43), (2, 1, 2, 205), (1, 33, 1, 43)])
def test_flash_attention_bwd(ops, B, N, H, hd, handoff):
    """Both dQ paths: dS tiles handed from the dK/dV kernel through the HBM scratch (default), and the dQ kernel that
    recomputes S / dP (ws == NULL).  Sizes cover N below one tile, N = k*256, and one row past a 128 / 256 boundary."""
    torch.manual_seed(N + 1)
    hdp = ops.rup(hd, 32)
    qkv = torch.zeros(B, N, 3, H, hdp)
    qkv[..., :hd] = torch.randn(B, N, 3, H, hd)
    qb = qkv.to(torch.bfloat16)
    do = torch.zeros(B, N, H, hdp); do[..., :hd] = torch.randn(B, N, H, hd)
    dob = do.to(torch.bfloat16)
    leaf = qb.float().requires_grad_(True)
    q, k, v = [leaf[:, :, i, :, :hd].permute(0, 2, 1, 3) for i in range(3)]
    scale = hd ** -0.5
    ref = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1) @ v
    (ref * dob.float()[..., :hd].permute(0, 2, 1, 3)).sum().backward()
    o = torch.empty(B, N, H, hdp, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    qd = qb.cuda()
    ops.attn_fwd(qd, o, lse, B, N, H, hd, hdp, scale)
    dqkv = torch.full((B, N, 3, H, hdp), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.attn_bwd(qd, o, dob.cuda(), lse, dqkv, B, N, H, hd, hdp, scale, handoff=handoff)
    got = dqkv.cpu().float()
    assert torch.isfinite(got).all() and not got[..., hd:].any()
    for i, name in enumerate("qkv"):
        g, r = got[:, :, i, :, :hd], leaf.grad[:, :, i, :, :hd]
        if r.norm() == 0:        # N = 1: softmax over one key is constant, dQ = dK = 0 exactly
            assert g.abs().max() < 1e-3, (name, g.abs().max())
            continue
        err = ((g - r).norm() / r.norm()).item()
        assert err < 2e-2, (name, err)


# ---------------------------------------------------------------- large-tile (256 x 16*NF) LDS-DMA GEMM
def _mk_operand(M, free, k, layout, ld_pad=0):
    """logical [free, k] matrix -> storage for the given layout, padded leading dim, garbage-free pads."""
    r8 = lambda x: (x + 7) // 8 * 8
    if layout == 0:
        ld = r8(k) + ld_pad
        t = torch.zeros(free, ld, dtype=torch.bfloat16); t[:, :k] = M
    else:
        ld = r8(free) + ld_pad
        t = torch.randn(k, ld).to(torch.bfloat16); t[:, :free] = M.t()
    return t, ld


@pytest.mark.parametrize("variant", [8, 10, 12, 13, 14, 16])
@pytest.mark.parametrize("la,lb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(256, 224, 64), (700, 1088, 1120), (1000, 420, 200), (513, 3360, 1088), (2500, 1040, 4112)])
def test_gemm256_layouts(ops, variant, la, lb, m, n, k):
    torch.manual_seed(m + n + k + la * 2 + lb + variant)
    A = bf(torch.randn(m, k)); B = bf(torch.randn(n, k))
    ref = A.float() @ B.float().t()
    a, lda = _mk_operand(A, m, k, la, ld_pad=8 if la == 0 else 0)
    b, ldb = _mk_operand(B, n, k, lb, ld_pad=16 if lb == 1 else 0)
    ldc = n + 8
    c = torch.full((m, ldc), float("nan"), device="cuda")
    ops.gemm(dev(a), dev(b), c, m, n, k, lda, ldb, ldc, a_layout=la, b_layout=lb, variant=variant)
    torch.cuda.synchronize()
    out = c.cpu()
    assert torch.isnan(out[:, n:]).all()               # nothing written past n
    assert _rel(out[:, :n], ref) < 1e-5
    torch.testing.assert_close(out[:, :n], ref, rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("variant", [13, 14])
def test_gemm256_epilogues(ops, variant):
    torch.manual_seed(variant)
    m, n, k = 777, 1120, 456
    A = bf(torch.randn(m, k)); W = bf(torch.randn(n, k) * 0.05); bias = torch.randn(n)
    pre = A.float() @ W.float().t() + bias
    c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); u = torch.empty_like(c)
    ops.gemm(dev(A), dev(W), c, m, n, k, k, k, n, bias=dev(bias), epilogue=ops.EPI_GELU, aux_out=u, variant=variant)
    torch.testing.assert_close(u.cpu().float(), _gelu_grad(pre), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(c.cpu().float(), torch.nn.functional.gelu(pre), rtol=2e-2, atol=2e-2)
    res = torch.randn(m, n)
    c2 = torch.empty(m, n, device="cuda")
    ops.gemm(dev(A), dev(W), c2, m, n, k, k, k, n, bias=dev(bias), epilogue=ops.EPI_RESIDUAL, aux_in=dev(res), ld_aux=n,
             variant=variant)
    torch.testing.assert_close(c2.cpu(), pre + res, rtol=1e-3, atol=2e-3)
    uu = bf(_gelu_grad(torch.randn(m, n)))
    c3 = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    Wt = W.t().contiguous()      # [K][N] for the k-strided B path (dgrad layout)
    ops.gemm(dev(A), dev(Wt), c3, m, n, k, k, n, n, b_layout=1, epilogue=ops.EPI_GELU_BWD, aux_in=dev(uu), ld_aux=n,
             variant=variant)
    torch.testing.assert_close(c3.cpu().float(), (A.float() @ W.float().t()) * uu.float(), rtol=2e-2, atol=2e-2)
    # alpha + bf16 out on the asymmetric-identity check
    mm = 512
    eye = torch.eye(mm).to(torch.bfloat16)
    Bm = ((torch.arange(mm)[:, None] * 3 + torch.arange(448)[None, :]) % 61).float()
    c5 = torch.empty(mm, 448, device="cuda")
    ops.gemm(dev(eye), dev(Bm.t().contiguous().to(torch.bfloat16)), c5, mm, 448, mm, mm, mm, 448, alpha=2.0, variant=variant)
    assert torch.equal(c5.cpu(), 2.0 * Bm)


@pytest.mark.parametrize("variant", [108, 110, 112])
@pytest.mark.parametrize("m,n,k", [(128, 192, 64), (700, 1088, 1120), (1000, 420, 200), (513, 3360, 1088), (2500, 1040, 4112),
                                   (13826, 1088, 136)])
def test_gemm128x_two_workgroups_per_cu(ops, variant, m, n, k):
    """The 128-row variant of the LDS-DMA kernel (4 waves, 128 x (16 * (variant - 100)) x 64 tiles, stages small enough
    for two workgroups per CU): k-contiguous operands with padded leading dimensions and ragged M / N / K, and every
    epilogue (bias + GELU with the saved derivative, bias + residual fp32, GELU-backward factor, alpha)."""
    torch.manual_seed(m + n + k + variant)
    A = bf(torch.randn(m, k)); B = bf(torch.randn(n, k) * 0.1)
    ref = A.float() @ B.float().t()
    a, lda = _mk_operand(A, m, k, 0, ld_pad=8)
    b, ldb = _mk_operand(B, n, k, 0, ld_pad=0)
    ldc = n + 8
    c = torch.full((m, ldc), float("nan"), device="cuda")
    ops.gemm(dev(a), dev(b), c, m, n, k, lda, ldb, ldc, variant=variant)
    out = c.cpu()
    assert torch.isnan(out[:, n:]).all()               # nothing written past n
    assert _rel(out[:, :n], ref) < 1e-5
    if n % 8:
        return
    bias = torch.randn(n)
    pre = ref + bias
    cb = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); u = torch.empty_like(cb)
    ops.gemm(dev(a), dev(b), cb, m, n, k, lda, ldb, n, bias=dev(bias), epilogue=ops.EPI_GELU, aux_out=u, variant=variant)
    torch.testing.assert_close(u.cpu().float(), _gelu_grad(pre), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(cb.cpu().float(), torch.nn.functional.gelu(pre), rtol=2e-2, atol=2e-2)
    res = torch.randn(m, n)
    c2 = torch.empty(m, n, device="cuda")
    ops.gemm(dev(a), dev(b), c2, m, n, k, lda, ldb, n, bias=dev(bias), epilogue=ops.EPI_RESIDUAL, aux_in=dev(res), ld_aux=n,
             variant=variant)
    torch.testing.assert_close(c2.cpu(), pre + res, rtol=1e-3, atol=5e-3)
    uu = bf(_gelu_grad(torch.randn(m, n)))
    c3 = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    ops.gemm(dev(a), dev(b), c3, m, n, k, lda, ldb, n, epilogue=ops.EPI_GELU_BWD, aux_in=dev(uu), ld_aux=n, alpha=0.5,
             variant=variant)
    torch.testing.assert_close(c3.cpu().float(), 0.5 * ref * uu.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("variant", [8, 10, 12, 13, 14, 16])
@pytest.mark.parametrize("split_k", [1, 3])
def test_gemm256_wgrad_maps(ops, variant, split_k):
    """weight-gradient form: dW = dY^T X with head-padded rows un-padded through the segment maps."""
    torch.manual_seed(5 + variant + split_k)
    tokens, n_out_phys, k_in_phys = 1500, 3 * 96, 528          # rows: 3 segments of 96 with 77 real; cols: 520 real
    seg_phys, seg_log, k_in = 96, 77, 520
    dY = bf(torch.randn(tokens, n_out_phys)); X = bf(torch.randn(tokens, k_in_phys))
    full = dY.float().t() @ X.float()
    keep = [r for r in range(n_out_phys) if r % seg_phys < seg_log]
    ref = full[keep][:, :k_in]
    out = torch.zeros(len(keep), k_in, device="cuda")
    ops.gemm(dev(dY), dev(X), out, n_out_phys, k_in, tokens, n_out_phys, k_in_phys, k_in, a_layout=1, b_layout=1,
             split_k=split_k, row_seg=(seg_phys, seg_log), variant=variant)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=2e-2)
    # column map (proj.weight: K_in is head padded)
    keepc = [c for c in range(k_in_phys) if c % 88 < 80]
    out2 = torch.zeros(n_out_phys, len(keepc), device="cuda")
    ops.gemm(dev(dY), dev(X), out2, n_out_phys, k_in_phys, tokens, n_out_phys, k_in_phys, len(keepc), a_layout=1, b_layout=1,
             split_k=split_k, col_seg=(88, 80), variant=variant)
    torch.testing.assert_close(out2.cpu(), full[:, keepc], rtol=1e-3, atol=2e-2)


# ----------------------------------------------------------------------------- input pipeline
def test_cine_prepare_golden_bit_exact(ops, golden):
    """vita_cine_prepare == reference image_normalization on the assembled volume (g7), bit for bit."""
    from vita_amd.data import prepare_cine_batch
    g = golden("g7_cine.npz")
    for tag in ("a", "b"):
        lax, sax = g[f"{tag}_lax"], g[f"{tag}_sax"]
        z_min, z_num = (int(v) for v in g[f"{tag}_z"])
        T = lax.shape[-1]
        out, mm = prepare_cine_batch(dev(torch.from_numpy(lax))[None], dev(torch.from_numpy(sax))[None], z0=z_min,
                                     sax_slice_num=z_num, time_frame=T, return_minmax=True)
        assert np.array_equal(out[0].cpu().numpy(), g[f"{tag}_norm"])
        window = sax[:, :, z_min:z_min + z_num]
        assert mm[0, 0].item() == min(lax.min(), window.min()) and mm[0, 1].item() == max(lax.max(), window.max())
        for keep, ratio in ((5, 1), (3, 2)):
            tf = len(range(0, T, ratio))
            out = prepare_cine_batch(dev(torch.from_numpy(lax))[None], dev(torch.from_numpy(sax))[None], z0=z_min,
                                     sax_slice_num=z_num, time_frame=tf, frame_to_keep=keep, t_downsampling_ratio=ratio)
            ref = O.cine_getitem(lax, sax, z_min=z_min, z_num=z_num, time_frame=tf, frame_to_keep=keep, t_downsampling_ratio=ratio)
            assert np.array_equal(out[0].cpu().numpy(), ref)


def test_cine_prepare_batch_full_size(ops):
    """Two full-size subjects (128x128, 3 LAX + 6 of 11 SAX slices, 50 frames): per-subject extrema, oracle equality,
    range [0, 1] with both ends attained, and the un-mirrored variant is the H/W transpose."""
    from vita_amd.data import prepare_cine_batch
    rng = np.random.RandomState(5)
    lax = (rng.rand(2, 128, 128, 3, 50) * 700).astype(np.float32)
    sax = (rng.rand(2, 128, 128, 11, 50) * 1300 - 40).astype(np.float32)
    sax[1] *= 0.5
    out = prepare_cine_batch(dev(torch.from_numpy(lax)), dev(torch.from_numpy(sax)), z0=3, sax_slice_num=6, time_frame=50)
    assert out.shape == (2, 9, 50, 128, 128)
    for b in range(2):
        ref = O.cine_getitem(lax[b], sax[b], z_min=3, z_num=6)
        assert np.array_equal(out[b].cpu().numpy(), ref)
        assert out[b].min().item() == 0.0 and out[b].max().item() == 1.0
    plain = prepare_cine_batch(dev(torch.from_numpy(lax)), dev(torch.from_numpy(sax)), z0=3, sax_slice_num=6, time_frame=50,
                               frame_to_keep=5, mirror=False)
    idx = [round(i * 49 / 4) for i in range(5)]
    assert torch.equal(plain, out[:, :, idx].transpose(-1, -2))
    with pytest.raises(ValueError):
        prepare_cine_batch(dev(torch.from_numpy(lax)), dev(torch.from_numpy(sax)), z0=8, sax_slice_num=6)
    with pytest.raises(RuntimeError):
        prepare_cine_batch(torch.from_numpy(lax), torch.from_numpy(sax))


@pytest.mark.parametrize("lb", [0, 1])
def test_gemm_large_m_epilogues(ops, lb):
    """M beyond one round of 256-row tiles (43 x 6 tiles, ragged last row panel) through the shape-chosen kernel with
    every epilogue (residual / saved GELU derivative / GELU-backward factor)."""
    torch.manual_seed(7 + lb)
    m, n, k = 10752 + 200, 1088, 136
    A = bf(torch.randn(m, k)); W = bf(torch.randn(n, k) * 0.1); bias = torch.randn(n)
    wt = dev(W) if lb == 0 else dev(W.t().contiguous())
    ldb = k if lb == 0 else n
    pre_nb = A.float() @ W.float().t()
    pre = pre_nb + bias
    kw = dict(b_layout=lb)
    c0 = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm(dev(A), wt, c0, m, n, k, k, ldb, n, **kw)
    torch.testing.assert_close(c0.cpu().float(), pre_nb, rtol=2e-2, atol=2e-2)
    c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); u = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm(dev(A), wt, c, m, n, k, k, ldb, n, bias=dev(bias), epilogue=ops.EPI_GELU, aux_out=u, **kw)
    torch.testing.assert_close(u.cpu().float(), _gelu_grad(pre), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(c.cpu().float(), torch.nn.functional.gelu(pre), rtol=2e-2, atol=2e-2)
    res = torch.randn(m, n)
    c2 = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm(dev(A), wt, c2, m, n, k, k, ldb, n, bias=dev(bias), epilogue=ops.EPI_RESIDUAL, aux_in=dev(res), ld_aux=n, **kw)
    torch.testing.assert_close(c2.cpu(), pre + res, rtol=1e-3, atol=2e-3)
    uu = bf(_gelu_grad(torch.randn(m, n)))
    c3 = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    ops.gemm(dev(A), wt, c3, m, n, k, k, ldb, n, epilogue=ops.EPI_GELU_BWD, aux_in=dev(uu), ld_aux=n, **kw)
    torch.testing.assert_close(c3.cpu().float(), pre_nb * uu.float(), rtol=2e-2, atol=2e-2)


def test_empty_inputs_are_rejected_before_any_launch(ops):
    """Empty batches / sequences never reach a kernel: every entry point validates sizes and pointers on the host and the
    wrapper raises RuntimeError("... invalid argument ...").  (Stock torch would return empty tensors; the reference never
    feeds an empty batch - its DataLoaders drop nothing and B >= 1 - so the drop-in fails loudly instead of guessing.)"""
    d = "cuda"
    bf = torch.bfloat16
    from vita_amd.losses import CLIPLoss
    cases = {
        "layernorm rows=0": lambda: ops.layernorm_fwd(torch.zeros(0, 64, device=d), 64, torch.ones(33, device=d), torch.zeros(33, device=d), 0, 33, 64),
        "gemm m=0": lambda: ops.gemm(torch.zeros(0, 64, device=d, dtype=bf), torch.zeros(64, 64, device=d, dtype=bf),
                                     torch.zeros(0, 64, device=d, dtype=bf), 0, 64, 64, 64, 64, 64),
        "argsort L=0": lambda: ops.argsort_mask(torch.zeros(2, 0, device=d), 0),
        "argsort n=0": lambda: ops.argsort_mask(torch.zeros(0, 16, device=d), 4),
        "cast rows=0": lambda: ops.cast_bf16(torch.zeros(0, 64, device=d), 64, 0, 33, 64),
        "colsum rows=0": lambda: ops.colsum(torch.zeros(0, 64, device=d, dtype=bf), 64, 0, 64),
        "attention N=0": lambda: ops.attn_fwd(torch.zeros(1, 0, 3, 1, 32, device=d, dtype=bf), torch.zeros(1, 0, 1, 32, device=d, dtype=bf),
                                              torch.zeros(1, 1, 0, device=d), 1, 0, 1, 32, 32, 1.0),
        "clip B=0": lambda: CLIPLoss(0.1)(torch.zeros(0, 128, device=d), torch.zeros(0, 128, device=d)),
    }
    for name, fn in cases.items():
        with pytest.raises(RuntimeError, match="invalid argument"):
            fn()
    torch.cuda.synchronize()       # nothing was launched, nothing faulted
