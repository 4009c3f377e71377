"""Op-level parity of the HIP kernels (through the C ABI) against plain fp32 PyTorch on CPU.
f32 mode: tight tolerances (exact-f32 kernels). bf16 mode: inputs are pre-rounded to bf16 on both
sides, so the tolerance only has to cover bf16 output rounding + accumulation order."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    from gslora_hip import ops as _ops
    from gslora_hip import _lib
    _lib.load()
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 1000 * len(shape) + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def as_dt(t, dt):
    """value actually seen by the kernel (bf16-rounded in bf16 mode), as f32 on CPU"""
    return t.to(dt).float()


def tol(dt, f32, bf16):
    return f32 if dt == torch.float32 else bf16


def relerr(a, b):
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


DTS = [torch.float32, torch.bfloat16, torch.float16]      # parity mode and the two operand formats of the speed mode


@pytest.mark.parametrize("dt", DTS)
def test_gemm_store_asymmetric(ops, dt):
    """A = I with an ASYMMETRIC W catches transposed / permuted MFMA fragment layouts."""
    from gslora_hip import _lib as L
    M, N, K = 128, 256, 128
    A = torch.zeros(M, K); A[torch.arange(M), torch.arange(M) % K] = 1.0
    W = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 16.0 - 4.0
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(A.cuda().to(dt), W.cuda().to(dt), out)
    ref = as_dt(A, dt) @ as_dt(W, dt).t()
    assert relerr(out.float().cpu(), ref) < tol(dt, 1e-6, 4e-3)


@pytest.mark.parametrize("dt", DTS)
# the last two shapes (170 images x 197 tokens: 131 M-tiles of 256 with a ragged last tile) run the 8-phase 256x256 kernel and its
# staged full-row epilogues in bf16 mode
# tile choice by shape (bf16): the first four run the 64x64 ring kernel (<= 256 tiles of 128x128), (2600, 2048) the 128x128 kernel,
# (9000, 64) the 256x128 ring, the 33 490-row shapes the 256x256 8-phase kernel
# ((1576, 2048, ..): 800 tiles of 64x64 exceed the resident workgroups -> the 64x128 form of the ring kernel)
@pytest.mark.parametrize("M,N,K1,K2", [(591, 192, 128, 64), (130, 64, 64, 0), (257, 2048, 512, 64), (788, 512, 2048, 64), (1576, 2048, 512, 64), (1576, 512, 1536, 0), (900, 512, 1088, 64),
                                       (2600, 2048, 512, 64), (9000, 64, 512, 0), (33490, 512, 192, 0), (33490, 2048, 512, 64)])
def test_gemm_epilogues(ops, dt, M, N, K1, K2):
    from gslora_hip import _lib as L
    A1, W1 = rnd(M, K1, seed=1), rnd(N, K1, seed=2, scale=K1 ** -0.5)
    A2 = W2 = None
    if K2:
        A2, W2 = rnd(M, K2, seed=3), rnd(N, K2, seed=4, scale=0.1)
        A2[:, 8:] = 0
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    aux = rnd(M, N, seed=7)
    acc = as_dt(A1, dt) @ as_dt(W1, dt).t()
    if K2:
        acc = acc + as_dt(A2, dt) @ as_dt(W2, dt).t()
    c = lambda t: None if t is None else t.cuda().to(dt)
    t_out, t_abs = tol(dt, 2e-5, 1.5e-2), tol(dt, 2e-5, 2e-2)
    # STORE with alpha + bias
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(c(A1), c(W1), out, A2=c(A2), W2=c(W2), alpha=0.5, bias=bias.cuda())
    assert relerr(out.float().cpu(), 0.5 * acc + bias) < t_out
    # STORE_F32
    outf = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt(c(A1), c(W1), outf, epilogue=L.EPI_STORE_F32, A2=c(A2), W2=c(W2))
    assert relerr(outf.cpu(), acc) < tol(dt, 2e-5, 2e-3)
    # BIAS_RES_F32
    ops.gemm_nt(c(A1), c(W1), outf, epilogue=L.EPI_BIAS_RES_F32, A2=c(A2), W2=c(W2), bias=bias.cuda(), res=res.cuda())
    assert relerr(outf.cpu(), acc + bias + res) < tol(dt, 2e-5, 2e-3)
    # BIAS_GELU (+ derivative)
    out2 = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(c(A1), c(W1), out, epilogue=L.EPI_BIAS_GELU, A2=c(A2), W2=c(W2), bias=bias.cuda(), out2=out2)
    a = (acc + bias).requires_grad_(True)
    g = F.gelu(a)
    gp, = torch.autograd.grad(g.sum(), a)
    ulp = 0.0 if dt == torch.float32 else 2.0 ** -8       # bf16 outputs: one rounding of the stored value on top of the absolute slack
    assert ((out.float().cpu() - g.detach()).abs() - ulp * g.detach().abs()).max() < t_abs
    assert ((out2.float().cpu() - gp).abs() - ulp * gp.abs()).max() < t_abs
    # MUL
    ops.gemm_nt(c(A1), c(W1), out, epilogue=L.EPI_MUL, A2=c(A2), W2=c(W2), aux=c(aux))
    assert relerr(out.float().cpu(), acc * as_dt(aux, dt)) < t_out
    # PATCH
    T = 197 if M % 197 == 0 else 13 if M % 13 == 0 else M
    pos, cls = rnd(T, N, seed=8), rnd(N, seed=9)
    ops.gemm_nt(c(A1), c(W1), outf, epilogue=L.EPI_PATCH, A2=c(A2), W2=c(W2), bias=bias.cuda(), pos=pos.cuda(), cls=cls.cuda(), T=T)
    tok = torch.arange(M) % T
    ref = torch.where((tok == 0)[:, None], cls[None, :].expand(M, N), acc + bias) + pos[tok]
    assert relerr(outf.cpu(), ref) < tol(dt, 2e-5, 2e-3)


@pytest.mark.parametrize("dt", DTS)
def test_gemm_dropout_epilogue(ops, dt):
    from gslora_hip import _lib as L
    M, N, K = 256, 128, 64
    A, W, bias, res = rnd(M, K), rnd(N, K), rnd(N), rnd(M, N)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt(A.cuda().to(dt), W.cuda().to(dt), out, epilogue=L.EPI_BIAS_RES_F32, bias=bias.cuda(), res=res.cuda(),
                p_drop=0.25, seed=77, site=5)
    keep = ops.dropout_mask(M * N, 0.25, 77, 5, "cuda").cpu().reshape(M, N).float()
    ref = (as_dt(A, dt) @ as_dt(W, dt).t() + bias) * keep / 0.75 + res
    assert relerr(out.cpu(), ref) < tol(dt, 2e-5, 2e-3)
    assert abs(keep.mean().item() - 0.75) < 0.01
    keep2 = ops.dropout_mask(M * N, 0.25, 77, 6, "cuda").cpu().reshape(M, N).float()
    assert (keep2 != keep).float().mean() > 0.2          # sites are independent streams
    big = ops.dropout_mask(1 << 22, 0.1, 1, 0, "cuda").float()
    assert abs(big.mean().item() - 0.9) < 1e-3


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("D", [64, 128, 512, 768])
def test_layernorm(ops, dt, D):
    M = 37
    x, g, b = rnd(M, D, seed=1, scale=2.0) + 0.5, 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), D, M, D, g.cuda(), b.cuda(), 1e-5, dt)
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    assert (y.float().cpu() - ref).abs().max() < tol(dt, 1e-5, 3e-2)
    assert (mean.cpu() - x.mean(1)).abs().max() < 1e-5
    assert relerr(rstd.cpu(), 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)) < 1e-5
    # backward
    dy, dres = rnd(M, D, seed=4), rnd(M, D, seed=5)
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr, (D,), g, b, 1e-5).backward(as_dt(dy, dt))
    dx, dxb = ops.layernorm_bwd(dy.cuda().to(dt), x.cuda(), D, g.cuda(), mean, rstd, dres.cuda())
    assert (dx.cpu() - (xr.grad + dres)).abs().max() < 2e-5
    assert (dxb.float().cpu() - (xr.grad + dres)).abs().max() < tol(dt, 2e-5, 3e-2)
    # masked copy
    dx2, dxb2 = ops.layernorm_bwd(dy.cuda().to(dt), x.cuda(), D, g.cuda(), mean, rstd, None, p_drop=0.5, seed=3, site=9)
    keep = ops.dropout_mask(M * D, 0.5, 3, 9, "cuda").cpu().reshape(M, D).float()
    assert (dx2.cpu() - xr.grad).abs().max() < 2e-5
    assert (dxb2.float().cpu() - xr.grad * keep * 2).abs().max() < tol(dt, 4e-5, 6e-2)


def attn_ref(qkv, B, T, H, scale):
    q, k, v = qkv.reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = (torch.einsum("bhid,bhjd->bhij", q, k) * scale).softmax(-1)
    return torch.einsum("bhij,bhjd->bhid", p, v).permute(0, 2, 1, 3).reshape(B * T, H * 64)


@pytest.mark.parametrize("dt", DTS)
# (193 / 208: the edges of the compile-time tail-mask specialisation of the bf16 kernels, 192 / 209: the generic form just outside)
@pytest.mark.parametrize("B,T,H", [(3, 197, 2), (2, 26, 1), (2, 37, 2), (1, 64, 1), (1, 224, 1), (2, 193, 2), (2, 208, 1), (2, 192, 1), (1, 209, 2)])
def test_attention(ops, dt, B, T, H):
    scale = (H * 64) ** -0.5 * 3.0
    qkv = rnd(B * T, 3 * H * 64, seed=B + T, scale=1.5)
    qkv_d = as_dt(qkv, dt).requires_grad_(True)
    ref = attn_ref(qkv_d, B, T, H, scale)
    o, lse = ops.attention_fwd(qkv.cuda().to(dt), B, T, H, scale)
    assert (o.float().cpu() - ref.detach()).abs().max() < tol(dt, 2e-5, 2e-2)
    q, k, _ = qkv_d.detach().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp(torch.einsum("bhid,bhjd->bhij", q, k) * scale, -1)
    assert (lse.cpu() - lse_ref).abs().max() < tol(dt, 2e-5, 2e-3)
    d_o = rnd(B * T, H * 64, seed=5)
    ref.backward(as_dt(d_o, dt))
    dqkv = ops.attention_bwd(qkv.cuda().to(dt), o, d_o.cuda().to(dt), lse, B, T, H, scale)
    err = (dqkv.float().cpu() - qkv_d.grad).abs().max().item()
    assert err < tol(dt, 5e-5, 3e-2) * max(1.0, qkv_d.grad.abs().max().item()), err


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,r", [(591, 512, 8), (1000, 2048, 8), (333, 128, 4), (700, 768, 16)])
def test_lora_grad(ops, dt, M, N, r):
    Y, U = rnd(M, N, seed=1), rnd(M, 64, seed=2)
    U[:, r:] = 0
    ref = as_dt(Y, dt).t() @ as_dt(U, dt)[:, :r]          # [N, r]
    G = torch.zeros(N * r, device="cuda")
    ops.lora_grad(Y.cuda().to(dt), U.cuda().to(dt), G, r, 1, r, accumulate=False)
    assert relerr(G.cpu().reshape(N, r), ref) < 1e-5
    ops.lora_grad(Y.cuda().to(dt), U.cuda().to(dt), G, r, 1, r, accumulate=True)
    assert relerr(G.cpu().reshape(N, r), 2 * ref) < 1e-5
    Gt = torch.zeros(r * N, device="cuda")
    ops.lora_grad(Y.cuda().to(dt), U.cuda().to(dt), Gt, 1, N, r, accumulate=False)     # transposed output [r, N]
    assert relerr(Gt.cpu().reshape(r, N), ref.t()) < 1e-5


def test_patchify_and_pack(ops):
    img = rnd(3, 3, 40, 40)
    p = 8
    out = ops.patchify(img.cuda(), p, torch.float32).cpu().reshape(3, 26, 192)
    ref = img.reshape(3, 3, 5, p, 5, p).permute(0, 2, 4, 3, 5, 1).reshape(3, 25, 192)
    assert torch.equal(out[:, 1:], ref) and (out[:, 0] == 0).all()
    outb = ops.patchify(img.cuda(), p, torch.bfloat16).float().cpu().reshape(3, 26, 192)
    assert torch.equal(outb[:, 1:], ref.bfloat16().float())
    W = rnd(40, 24)
    assert torch.equal(ops.transpose_cast(W.cuda(), torch.float32).cpu(), W.t().contiguous())
    assert torch.equal(ops.transpose_cast(W.cuda(), torch.bfloat16).cpu(), W.t().contiguous().bfloat16())
    assert torch.equal(ops.cast(W.cuda(), torch.bfloat16).cpu(), W.bfloat16())
    A = rnd(8, 128)    # lora_A [r, K]
    pk = ops.pack_pad(A.cuda(), 1, 128, 128, 8, 128, 64, torch.float32).cpu()      # AT_cols: [K, 64]
    assert torch.equal(pk[:, :8], A.t()) and (pk[:, 8:] == 0).all()
    pk = ops.pack_pad(A.cuda(), 128, 1, 8, 128, 64, 128, torch.float32, scale=0.5).cpu()   # A_rows: [64, K]
    assert torch.equal(pk[:8], 0.5 * A) and (pk[8:] == 0).all()


def test_head_and_losses(ops):
    B, T, D, C = 5, 26, 512, 100
    x = rnd(B * T, D, seed=1, scale=2.0)
    g, b, W = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3), rnd(C, D, seed=4)
    y = torch.tensor([3, 99, 0, 42, 42])
    Wn = ops.cosface_prep(W.cuda())
    assert (Wn.cpu() - F.normalize(W)).abs().max() < 1e-6
    logits, emb, mean, rstd = ops.head_fwd(x.cuda(), B, T, D, g.cuda(), b.cuda(), 1e-5, Wn, y.cuda(), 64.0, 0.35)
    xr = x.clone().requires_grad_(True)
    emb_ref = F.layer_norm(xr.reshape(B, T, D)[:, 0], (D,), g, b, 1e-5)
    cos = F.linear(F.normalize(emb_ref), F.normalize(W))
    oh = F.one_hot(y, C).float()
    logits_ref = 64.0 * (cos - 0.35 * oh)
    assert (emb.cpu() - emb_ref.detach()).abs().max() < 1e-5
    assert (logits.cpu() - logits_ref.detach()).abs().max() < 5e-5
    # CE fwd/bwd
    out2 = ops.ce_fwd(logits, y.cuda()).cpu()
    ce = F.cross_entropy(logits_ref, y, reduction="sum")
    assert abs(out2[0].item() - ce.item()) < 1e-3 * max(1, ce.item())
    assert out2[1].item() == (logits_ref.argmax(1) == y).sum().item()
    coef = torch.tensor([0.7], device="cuda")
    dl = ops.ce_bwd(logits, y.cuda(), coef, 1.0 / B)
    proto = rnd(C, D, seed=7)
    kl = ops.proto_kl_fwd(emb, y.cuda(), proto.cuda()).cpu()
    kl_ref = F.kl_div(F.log_softmax(emb_ref, 1), F.log_softmax(proto[y], 1), reduction="sum", log_target=True)
    assert abs(kl.item() - kl_ref.item()) < 1e-4 * max(1, abs(kl_ref.item()))
    coef2 = torch.tensor([-0.3], device="cuda")
    de = ops.proto_kl_bwd(emb, y.cuda(), proto.cuda(), coef2, 1.0 / B)
    total = 0.7 * ce / B - 0.3 * kl_ref / B
    total.backward()
    # head backward consumes dlogits + demb and must reproduce autograd's dx
    dx, dxb = ops.head_bwd(dl, de, x.cuda(), B, T, D, g.cuda(), mean, rstd, emb, Wn, 64.0, torch.float32)
    assert (dx.cpu() - xr.grad).abs().max() < 2e-5 * max(1.0, xr.grad.abs().max().item())
    assert torch.equal(dx, dxb)
    nz = dx.cpu().reshape(B, T, D)
    assert (nz[:, 1:] == 0).all() and (nz[:, 0] != 0).any()
    # no-label path
    lg, emb2, _, _ = ops.head_fwd(x.cuda(), B, T, D, g.cuda(), b.cuda(), 1e-5, None, None, 64.0, 0.35)
    assert lg is None and torch.equal(emb2, emb)


def test_group_norms_and_mask(ops):
    sizes = [(8, 512), (2048, 8), (8, 2048), (512, 8)] * 3
    ts = [rnd(*s, seed=i) * (0.0 if i // 4 == 1 else 1.0) for i, s in enumerate(sizes)]   # group 1 exactly zero
    flat = torch.cat([t.reshape(-1) for t in ts]).cuda()
    offs = np.cumsum([0] + [t.numel() for t in ts])[:-1]
    toff = torch.tensor(offs, dtype=torch.int64).cuda()
    tnum = torch.tensor([t.numel() for t in ts], dtype=torch.int64).cuda()
    tgrp = torch.tensor([i // 4 for i in range(12)], dtype=torch.int32).cuda()
    out = ops.group_norms_fwd(flat, toff, tnum, tgrp, 3, tau=0.0)
    gn_ref = torch.stack([torch.sqrt(sum((t ** 2).sum() for t in ts[4 * g:4 * g + 4])) for g in range(3)])
    cn_ref = torch.stack([sum(t.norm() for t in ts[4 * g:4 * g + 4]) for g in range(3)])
    assert relerr(out["group_norm"].cpu(), gn_ref) < 1e-6
    assert relerr(out["cal_norm"].cpu(), cn_ref) < 1e-6
    assert abs(out["loss"].item() - gn_ref.sum().item()) < 1e-4
    # bit-exact selection mask vs the same predicate on the oracle norms
    assert out["mask"].cpu().tolist() == (gn_ref > 0.0).to(torch.uint8).tolist() == [1, 0, 1]
    tau = float(gn_ref[0]) + 1e-3
    out2 = ops.group_norms_fwd(flat, toff, tnum, tgrp, 3, tau=tau)
    assert out2["mask"].cpu().tolist() == (gn_ref > tau).to(torch.uint8).tolist()
    # backward: grad += coef*scale * t/||g|| ; zero group -> 0 (no NaN)
    grad = torch.ones_like(flat)
    coef = torch.tensor([2.0], device="cuda")
    ops.group_norms_bwd(flat, toff, tnum, tgrp, out["group_norm"], coef, 0.5, grad)
    ref = torch.cat([(1.0 + (t / gn_ref[i // 4] if gn_ref[i // 4] > 0 else torch.zeros_like(t))).reshape(-1)
                     for i, t in enumerate(ts)])
    assert (grad.cpu() - ref).abs().max() < 1e-6
    assert torch.isfinite(grad).all()


def test_adamw_matches_torch(ops):
    n = 245760
    p0, g = rnd(n, seed=1), rnd(n, seed=2)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-2, weight_decay=0.05, eps=1e-8)
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in (1, 2, 3):
        pt.grad = g * step
        opt.step()
        ops.adamw_flat(p, (g * step).cuda(), m, v, 1e-2, 0.9, 0.999, 1e-8, 0.05, step)
        assert (p.cpu() - pt.detach()).abs().max() < 2e-6


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,T,H", [(3, 197, 2), (2, 26, 1)])
def test_attention_bwd_cls_equals_dense_backward(ops, dt, B, T, H):
    """cls-only output gradient: the rank-1 kernel must equal the dense backward fed with zeros elsewhere."""
    scale = (H * 64) ** -0.5 * 2.0
    qkv = rnd(B * T, 3 * H * 64, seed=11, scale=1.2).cuda().to(dt)
    o, lse = ops.attention_fwd(qkv, B, T, H, scale)
    d_cls = rnd(B, H * 64, seed=12)
    d_full = torch.zeros(B, T, H * 64); d_full[:, 0] = d_cls
    ref = ops.attention_bwd(qkv, o, d_full.reshape(B * T, -1).cuda().to(dt), lse, B, T, H, scale).float().cpu()
    got = ops.attention_bwd_cls(qkv, o, d_cls.cuda().to(dt), lse, B, T, H, scale).float().cpu()
    assert (got - ref).abs().max() < tol(dt, 2e-5, 2e-2) * max(1.0, ref.abs().max().item())
    dq = got.reshape(B, T, 3, H * 64)[:, 1:, 0]
    assert (dq == 0).all()
    # and against autograd in f32
    if dt == torch.float32:
        q = qkv.float().cpu().requires_grad_(True)
        attn_ref(q, B, T, H, scale).backward(d_full.reshape(B * T, -1))
        assert (got - q.grad).abs().max() < 5e-5 * max(1.0, q.grad.abs().max().item())


@pytest.mark.parametrize("B,T,H", [(80, 197, 8), (131, 150, 4), (70, 208, 8), (67, 65, 8), (70, 193, 8), (66, 192, 8)])
def test_attention_fwd_persistent_bit_identical_to_per_item_kernel(ops, B, T, H, monkeypatch):
    """bf16, 64 < T <= 208, B*H >= 2 x CUs: the persistent wave-specialised forward (3 loader waves + 13 compute waves per CU) must
    reproduce the one-item-per-workgroup kernel bit for bit (same fragments, same operation order) — including the ragged last round
    of items and the zero rows of the panels."""
    dt = torch.bfloat16
    scale = 64 ** -0.5
    qkv = rnd(B * T, 3 * H * 64, seed=31, scale=1.3).cuda().to(dt)
    o1, lse1 = ops.attention_fwd(qkv, B, T, H, scale)
    o1b, lse1b = ops.attention_fwd(qkv, B, T, H, scale)
    from gslora_hip import _lib as L
    monkeypatch.setenv("GSL_ATTN_PERSISTENT", "0")      # a knob of the development build only
    with L.use_dev():
        o0, lse0 = ops.attention_fwd(qkv, B, T, H, scale)
    assert torch.equal(o1, o0) and torch.equal(lse1, lse0)
    assert torch.equal(o1, o1b) and torch.equal(lse1, lse1b)


@pytest.mark.parametrize("B,T,H", [(3, 197, 8), (2, 150, 4), (5, 224, 2), (150, 197, 8), (131, 150, 12), (300, 224, 2), (3, 193, 4), (3, 192, 4), (5, 208, 4), (3, 209, 2)])
def test_attention_bwd_fused_bit_identical_to_two_kernel_form(ops, B, T, H, monkeypatch):
    """bf16, T > 64: the single-launch backward (dQ phase then dK/dV phase over the same LDS panels) == the dQ kernel + the dK/dV kernel."""
    dt = torch.bfloat16
    scale = 64 ** -0.5
    qkv = rnd(B * T, 3 * H * 64, seed=21, scale=1.3).cuda().to(dt)
    o, lse = ops.attention_fwd(qkv, B, T, H, scale)
    d_o = rnd(B * T, H * 64, seed=22).cuda().to(dt)
    fused = ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale)
    from gslora_hip import _lib as L
    monkeypatch.setenv("GSL_ATTN_BWD_SPLIT", "1")       # a knob of the development build only
    with L.use_dev():
        split = ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale)
    assert torch.equal(fused, split)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,H,hm", [(70, 197, 8, 0), (67, 197, 8, 1), (64, 208, 8, 0), (128, 193, 4, 1), (43, 197, 12, 1), (72, 200, 8, 0)])
def test_attention_bwd_merged_bit_identical_to_fused_kernel(ops, dt, B, T, H, hm, monkeypatch):
    """16-bit, 192 < T <= 208, at least two items per CU: the merged backward (every score tile computed once by the wave that owns its
    key tile, dS parked in LDS for the dQ products, loader waves one item ahead) must reproduce the fused two-phase kernel bit for bit:
    same MFMA operands in the same order, delta accumulated in the same order. Ragged last round of items (B*H not a multiple of the
    CU count), batch sizes with and without the XCD item remap, both input layouts, and against the fp32 torch reference."""
    scale = 64 ** -0.5
    qkv32 = rnd(B * T, 3 * H * 64, seed=41, scale=1.3)
    qkv = qkv32.cuda().to(dt)
    qin = _to_head_major(qkv, B, T, H) if hm else qkv
    o, lse = ops.attention_fwd(qin, B, T, H, scale, layout=hm)
    d_o32 = rnd(B * T, H * 64, seed=42)
    d_o = d_o32.cuda().to(dt)
    from gslora_hip import _lib as L
    # (the product library takes the merged kernel from 8 items per CU; the development build's knob lowers that to 2 for these sizes)
    monkeypatch.setenv("GSL_ATTN_BWD_MERGED_MIN", "2")
    with L.use_dev():
        merged = ops.attention_bwd(qin, o, d_o, lse, B, T, H, scale, layout=hm)
        again = ops.attention_bwd(qin, o, d_o, lse, B, T, H, scale, layout=hm)
    monkeypatch.setenv("GSL_ATTN_BWD_MERGED", "0")
    with L.use_dev():
        fused = ops.attention_bwd(qin, o, d_o, lse, B, T, H, scale, layout=hm)
    assert torch.equal(merged, again)
    assert torch.equal(merged, fused)
    q = as_dt(qkv32, dt).requires_grad_(True)
    attn_ref(q, B, T, H, scale).backward(as_dt(d_o32, dt))
    err = (merged.float().cpu() - q.grad).abs().max().item()
    assert err < 3e-2 * max(1.0, q.grad.abs().max().item()), err


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,D,hmT,mu,outlier", [(4096, 1536, 512, 0, 0.5, 0.0), (197 * 24, 1536, 512, 197, 2.0, 8.0), (300, 512, 768, 0, 0.5, 0.0), (70, 256, 128, 0, 0.0, 0.0)])
def test_gemm_store_with_consumer_side_layernorm(ops, dt, M, N, D, hmT, mu, outlier):
    """EPI_STORE_LN / EPI_STORE_QKV_HM_LN: out = rstd (x W'^T - mean c) + d with W' = W gamma, c = rowsum(W') of the rounded W', d = W beta + bias,
    on the RAW stream x — against LN(x) W^T + bias in f64, and against the two-kernel form (LayerNorm kernel, then the plain GEMM): the folded form
    must be at least as accurate (it skips one 16-bit rounding of LN(x)), also with row means of 2 sigma and 8-sigma outlier channels; partial
    tiles, the small-tile and the 8-phase kernels, the head-major copy-out; the statistics-only LayerNorm call returns the LayerNorm kernel's."""
    from gslora_hip import _lib as L
    g = torch.Generator().manual_seed(3)
    x32 = torch.randn(M, D, generator=g) + mu * torch.randn(M, 1, generator=g)
    if outlier:
        x32[:, 7] += outlier; x32[:, D // 2] -= outlier
    x = x32.to(dt).cuda()
    gam = (1 + 0.2 * torch.randn(D, generator=g)).cuda(); bet = (0.1 * torch.randn(D, generator=g)).cuda()
    W = (torch.randn(N, D, generator=g) * D ** -0.5).cuda(); bias = (0.1 * torch.randn(N, generator=g)).cuda()
    xd = x.double(); m1 = xd.mean(1, keepdim=True); v1 = ((xd - m1) ** 2).mean(1, keepdim=True)
    ref = (((xd - m1) / (v1 + 1e-5).sqrt()) * gam.double() + bet.double()) @ W.double().t() + bias.double()
    xn, mean, rstd = ops.layernorm_fwd(x, D, M, D, gam, bet, 1e-5, dt)
    mean2, rstd2 = ops.layernorm_stats(x, D, M, D, gam, bet, 1e-5, dt)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    y0 = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(xn, W.to(dt), y0, bias=bias, epilogue=L.EPI_STORE_QKV_HM if hmT else L.EPI_STORE, T=hmT)
    wf = (W * gam[None, :]).to(dt); c = wf.float().sum(1).contiguous(); d = (W @ bet + bias).contiguous()
    y1 = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(x, wf, y1, epilogue=L.EPI_STORE_QKV_HM_LN if hmT else L.EPI_STORE_LN, T=hmT, pos=mean2, cls=rstd2, aux=c, bias=d)
    if hmT:      # both outputs are head-major [B][H][3][T][64]: compare in that layout against the permuted reference
        ref = _to_head_major(ref, M // hmT, hmT, N // 192)
    e0 = (y0.double() - ref).pow(2).mean().sqrt().item(); e1 = (y1.double() - ref).pow(2).mean().sqrt().item()
    print(f"[LN fold {dt} M={M}] rms error vs f64: two kernels {e0:.3e}, folded {e1:.3e}")
    assert e1 <= e0 * 1.02
    assert (y1.double() - ref).abs().max().item() < (3e-2 if dt == torch.bfloat16 else 4e-3) * max(1.0, ref.abs().max().item() / 4)
    with pytest.raises(RuntimeError):      # the four f32 vectors are required
        ops.gemm_nt(x, wf, y1, epilogue=L.EPI_STORE_LN, pos=mean2, cls=rstd2, aux=c)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_store_ln_on_the_cls_rows_with_strided_statistics(ops, dt):
    """EPI_STORE_LN with T > 0: A = the cls rows of a [B*T, D] stream (lda = T*D) and row m takes mean[m*T], rstd[m*T] of the full-tensor
    statistics — the last block's Q projection (cls rows only) without a gather: identical to the call on gathered rows / statistics."""
    from gslora_hip import _lib as L
    B, T, D, N = 96, 197, 512, 512
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(B * T, D, generator=g) + 0.5).to(dt).cuda()
    gam = (1 + 0.2 * torch.randn(D, generator=g)).cuda(); bet = (0.1 * torch.randn(D, generator=g)).cuda()
    W = (torch.randn(N, D, generator=g) * D ** -0.5).cuda()
    mean, rstd = ops.layernorm_stats(x, D, B * T, D, gam, bet, 1e-5, dt)
    wf = (W * gam[None, :]).to(dt); c = wf.float().sum(1).contiguous(); d = (W @ bet).contiguous()
    a = torch.empty(B, N, device="cuda", dtype=dt); b = torch.empty(B, N, device="cuda", dtype=dt)
    ops.gemm_nt(x.view(B, T * D)[:, :D], wf, a, epilogue=L.EPI_STORE_LN, T=T, pos=mean, cls=rstd, aux=c, bias=d)
    ops.gemm_nt(x.view(B, T, D)[:, 0].contiguous(), wf, b, epilogue=L.EPI_STORE_LN, pos=mean.view(B, T)[:, 0].contiguous(),
                cls=rstd.view(B, T)[:, 0].contiguous(), aux=c, bias=d)
    assert torch.equal(a, b)
    xd = x.view(B, T, D)[:, 0].double(); m1 = xd.mean(1, keepdim=True); v1 = ((xd - m1) ** 2).mean(1, keepdim=True)
    ref = (((xd - m1) / (v1 + 1e-5).sqrt()) * gam.double() + bet.double()) @ W.double().t()
    assert (a.double() - ref).abs().max().item() < (3e-2 if dt == torch.bfloat16 else 4e-3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [9000, 1000, 300])
def test_gemm_store_compact_second_output(ops, dt, M):
    """EPI_STORE with out2: a compact [M, 16] copy of output columns 0..15 (the operand form of the LoRA down-projection u1 that the
    gradient-fused FFN2-dX epilogue reads as contiguous 1 KB pieces): bit-identical to the primary output's columns at every M (the call
    is routed to the ring kernel whatever the row count), and rejected where it is not defined."""
    from gslora_hip import _lib as L
    K, N = 512, 64
    A = rnd(M, K, seed=3).cuda().to(dt)
    W = (rnd(N, K, seed=4) * 0.05).cuda().to(dt)
    W[8:] = 0                                   # rank 8 zero-padded, as lora_pack("A_rows") delivers it
    out = torch.zeros(M, N, device="cuda", dtype=dt)
    out2 = torch.full((M, 16), 7.0, device="cuda", dtype=dt)
    ref = torch.zeros(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(A, W, ref, alpha=0.125)
    ops.gemm_nt(A, W, out, alpha=0.125, out2=out2)
    assert torch.equal(out, ref)
    assert torch.equal(out2, ref[:, :16])
    exp = (A.float() @ W.float().t()) * 0.125
    assert (out2.float() - exp[:, :16]).abs().max() < 2e-2 * max(1.0, exp.abs().max().item())
    Wb = torch.zeros(256, K, device="cuda", dtype=dt)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(A, Wb, torch.zeros(M, 256, device="cuda", dtype=dt), out2=out2)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(A, W, out, bias=torch.zeros(N, device="cuda"), out2=out2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_attention_bwd_product_library_at_step_size_equals_fused_kernel(ops, dt, monkeypatch):
    """The PRODUCT library at a batch that takes the merged kernel there (8 items per CU: 264 images x 8 heads), head-major input as in the step,
    against the development build's fused kernel: bit-identical."""
    B, T, H = 264, 197, 8
    scale = 64 ** -0.5
    qkv = rnd(B * T, 3 * H * 64, seed=51, scale=1.2).cuda().to(dt)
    qin = _to_head_major(qkv, B, T, H)
    o, lse = ops.attention_fwd(qin, B, T, H, scale, layout=1)
    d_o = rnd(B * T, H * 64, seed=52).cuda().to(dt)
    prod = ops.attention_bwd(qin, o, d_o, lse, B, T, H, scale, layout=1)
    from gslora_hip import _lib as L
    monkeypatch.setenv("GSL_ATTN_BWD_MERGED", "0")      # a knob of the development build only
    with L.use_dev():
        fused = ops.attention_bwd(qin, o, d_o, lse, B, T, H, scale, layout=1)
    assert torch.equal(prod, fused)


@pytest.mark.parametrize("dt", DTS)
def test_layernorm_bwd_strided_inplace(ops, dt):
    """cls-row form: x / dres / dx rows are T*D apart, dy and the masked copy are compact."""
    B, T, D = 5, 7, 128
    x = rnd(B * T, D, seed=1, scale=2.0); g = 1 + 0.1 * rnd(D, seed=2); b = 0.1 * rnd(D, seed=3)
    _, mean, rstd = ops.layernorm_fwd(x.cuda(), D, B * T, D, g.cuda(), b.cuda(), 1e-5, dt)
    dy = rnd(B, D, seed=4)
    dres = rnd(B * T, D, seed=5)
    dense = dres.clone().cuda()
    pick = lambda t: t.view(B, T, -1)[:, 0].contiguous()
    dx, dxb = ops.layernorm_bwd(dy.cuda().to(dt), x.cuda(), T * D, g.cuda(), pick(mean.view(-1, 1)).view(-1), pick(rstd.view(-1, 1)).view(-1),
                                dense, dx=dense, io_row_stride=T * D, p_drop=0.5, seed=3, site=9, drop_row_stride=T * D)
    xr = x.view(B, T, D)[:, 0].clone().requires_grad_(True)
    F.layer_norm(xr, (D,), g, b, 1e-5).backward(as_dt(dy, dt))
    want = dres.clone().view(B, T, D)
    want[:, 0] += xr.grad
    assert (dx.cpu().view(B, T, D) - want).abs().max() < 2e-5
    keep = ops.dropout_mask(B * T * D, 0.5, 3, 9, "cuda").cpu().view(B, T, D)[:, 0].float()
    assert (dxb.float().cpu() - want[:, 0] * keep * 2).abs().max() < tol(dt, 4e-5, 6e-2)


# (the first four shapes run on the 64x64 ring kernel, the last two on the 256x256 8-phase kernel: the tile rule of gsl_gemm_nt_lora)
@pytest.mark.parametrize("M,N,K,r", [(2100, 512, 2048, 8), (1300, 2048, 512, 8), (1111, 768, 256, 16), (300, 256, 64, 4), (1576, 2048, 512, 8), (1600, 3072, 768, 16), (1576, 512, 2048, 8), (700, 768, 3136, 16), (1000, 512, 1088, 8),
                                     (20000, 512, 1024, 8), (16500, 2048, 512, 16)])
def test_gemm_nt_lora_in_kernel(ops, M, N, K, r):
    """out = epilogue(A W^T + t Q^T), t = s*(A P^T) computed inside the kernel; t is also returned (bf16, padded to 64)."""
    from gslora_hip import _lib as L
    dt = torch.bfloat16
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    P = torch.zeros(16, K); P[:r] = rnd(r, K, seed=3, scale=K ** -0.5)
    Q = torch.zeros(N, 32); Q[:, :r] = rnd(N, r, seed=4, scale=0.3)
    s = 1.0 / r
    bias, res, aux = rnd(N, seed=5), rnd(M, N, seed=6), rnd(M, N, seed=7)
    t_ref = (s * (as_dt(A, dt) @ as_dt(P, dt).t())).bfloat16().float()          # the kernel rounds t to bf16 before the update
    acc = as_dt(A, dt) @ as_dt(W, dt).t() + t_ref @ as_dt(Q, dt)[:, :16].t()
    c = lambda t: t.cuda().to(dt)
    tout = torch.full((M, 64), 7.0, device="cuda", dtype=dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt_lora(c(A), c(W), c(P), c(Q), s, tout, out)
    assert relerr(out.float().cpu(), acc) < 1.5e-2
    assert (tout.float().cpu()[:, :16] - t_ref).abs().max() < 2e-2 * max(1.0, t_ref.abs().max().item())
    assert (tout[:, r:] == 0).all()
    outf = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt_lora(c(A), c(W), c(P), c(Q), s, tout, outf, epilogue=L.EPI_BIAS_RES_F32, bias=bias.cuda(), res=res.cuda())
    assert relerr(outf.cpu(), acc + bias + res) < 4e-3
    ops.gemm_nt_lora(c(A), c(W), c(P), c(Q), s, None, out, epilogue=L.EPI_MUL, aux=c(aux))
    assert relerr(out.float().cpu(), acc * as_dt(aux, dt)) < 1.5e-2
    # agrees with the two-launch form it replaces
    t2 = torch.zeros(M, 64, device="cuda", dtype=dt)
    P64 = torch.zeros(64, K); P64[:16] = P
    Q64 = torch.zeros(N, 64); Q64[:, :32] = Q
    ops.gemm_nt(c(A), c(P64), t2, alpha=s)
    out2 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt(c(A), c(W), out2, epilogue=L.EPI_STORE_F32, A2=t2, W2=c(Q64))
    ops.gemm_nt_lora(c(A), c(W), c(P), c(Q), s, tout, outf, epilogue=L.EPI_BIAS_RES_F32, bias=torch.zeros(N).cuda(), res=torch.zeros(M, N).cuda())
    assert relerr(outf.cpu(), out2.cpu()) < 2e-3


@pytest.mark.parametrize("M,N,K,r", [(1000, 512, 256, 8), (2560, 768, 128, 16), (4099, 2048, 512, 8), (1300, 640, 192, 5)])
def test_gemm_nt_lora_mulgrad_fused_reductions(ops, M, N, K, r, monkeypatch):
    """FFN2-dX with both LoRA-gradient reductions of its tiles fused into the epilogue: `out` and `tout` are bit-identical to the
    unfused MUL GEMM (run without its K-tile rotation: the fused kernel keeps one K order for all N tiles so that the tile-local
    t it contracts equals tout bit for bit), and the gradients equal gsl_lora_grad on the same bf16 operands (f32 accumulation,
    different summation order)."""
    from gslora_hip import _lib as L
    dt = torch.bfloat16
    c = lambda t: t.cuda().to(dt)
    A, W = c(rnd(M, K, seed=1)), c(rnd(N, K, seed=2, scale=K ** -0.5))
    P = torch.zeros(16, K); P[:r] = rnd(r, K, seed=3, scale=K ** -0.5)
    Q = torch.zeros(N, 32); Q[:, :r] = rnd(N, r, seed=4, scale=0.3)
    P, Q = c(P), c(Q)
    aux, Y2 = c(rnd(M, N, seed=7)), c(rnd(M, N, seed=8))
    U1 = torch.zeros(M, 64); U1[:, :r] = rnd(M, r, seed=9)
    U1 = c(U1)
    s = 1.0 / r
    tout0 = torch.empty(M, 64, device="cuda", dtype=dt); out0 = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt_lora(A, W, P, Q, s, tout0, out0, epilogue=L.EPI_MUL, aux=aux)
    G1r = torch.zeros(N, r, device="cuda"); G2r = torch.zeros(r, N, device="cuda")
    ops.lora_grad(out0, U1, G1r, r, 1, r, accumulate=False)
    ops.lora_grad(Y2, tout0, G2r, 1, N, r, accumulate=False)
    for acc_flag in (False, True):
        tout = torch.full((M, 64), 3.0, device="cuda", dtype=dt); out = torch.empty(M, N, device="cuda", dtype=dt)
        G1 = torch.full((N, r), 0.5, device="cuda"); G2 = torch.full((r, N), -0.25, device="cuda")
        ops.gemm_nt_lora_mulgrad(A, W, P, Q, s, tout, out, aux, U1, G1, (r, 1), Y2, G2, (1, N), r, accumulate=acc_flag)
        assert torch.equal(out, out0) and torch.equal(tout, tout0)
        e1 = (G1 - (G1r + (0.5 if acc_flag else 0.0))).abs().max().item() / G1r.abs().max().item()
        e2 = (G2 - (G2r + (-0.25 if acc_flag else 0.0))).abs().max().item() / G2r.abs().max().item()
        assert e1 < 2e-5 and e2 < 2e-5, (e1, e2)
    # and against an f64 contraction of the same bf16 values
    ref1 = (out0.double().t() @ U1[:, :r].double()).float()
    ref2 = (tout0[:, :r].double().t() @ Y2.double()).float()
    assert relerr(G1r.cpu(), ref1.cpu()) < 1e-4 and relerr(G2r.cpu(), ref2.cpu()) < 1e-4


def test_dropout_mask_matches_the_documented_hash(ops):
    """DESIGN.md §5 / gsl_common.h: key = mix64(seed, site) (low 32 bits), w = (pair + key) * 0x9E3779B1 mod 2^32 with pair = i // 2,
    h = (w ^ (w >> 15)) * 0x85EBCA77 mod 2^32, element i keeps iff its 16-bit half of h (low for even i, high for odd i) >= round(p * 2^16).
    A numpy restatement must reproduce gsl_dropout_mask bit for bit (this pins the stream the epilogues advance incrementally)."""
    import numpy as np
    U = np.uint64
    M64 = (1 << 64) - 1

    def mix(seed, site):
        z = (seed * 0x9E3779B97F4A7C15 + site * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & M64
        z ^= z >> 29; z = (z * 0xD6E8FEB86659FD93) & M64; z ^= z >> 32
        return z & 0xFFFFFFFF

    for (n, p, seed, site) in [(10_000, 0.1, 1234, 5), (4099, 0.5, (0x5EED << 20) + 77, 9), (1 << 20, 0.25, 7, 1_000_000)]:
        key = mix(seed, site)
        i = np.arange(n, dtype=np.uint64)
        w = (((i >> U(1)) + U(key)) * U(0x9E3779B1)) & U(0xFFFFFFFF)
        h = ((w ^ (w >> U(15))) * U(0x85EBCA77)) & U(0xFFFFFFFF)
        sample = np.where((i & U(1)) == 0, h & U(0xFFFF), h >> U(16))
        keep = sample >= U(int(p * 65536.0 + 0.5))
        got = ops.dropout_mask(n, p, seed, site, "cuda").cpu().numpy().astype(bool)
        assert (got == keep).all(), (n, p, seed, site, int((got != keep).sum()))


# ---- entry points added for HIP-graph replay and launch-count reduction ---------------------------------------------------
def test_adamw_dev_bit_identical_to_value_form(ops):
    """gsl_adamw_flat_dev (step count / lr read from device memory) == gsl_adamw_flat for the same (step, lr)."""
    n = 10_001
    p0, g, m0, v0 = (rnd(n, seed=s).cuda() for s in (1, 2, 3, 4))
    v0 = v0.abs()
    for step, lr in ((1, 1e-2), (7, 5e-3), (1234, 1.24647e-5)):
        pa, ma, va = p0.clone(), m0.clone(), v0.clone()
        pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
        ops.adamw_flat(pa, g, ma, va, lr, 0.9, 0.999, 1e-8, 0.05, step)
        ops.adamw_flat_dev(pb, g, mb, vb, torch.tensor([lr], device="cuda", dtype=torch.float32), 0.9, 0.999, 1e-8, 0.05,
                           torch.tensor([step], device="cuda", dtype=torch.int64))
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), step


def test_dropout_seed_on_device_equals_seed_by_value(ops):
    """bit 31 of `site` turns `seed` into a device pointer: same masks, same outputs — for the GEMM epilogues, LayerNorm-bwd and the
    mask helper."""
    from gslora_hip import _lib as L
    seed = (0x5EED << 20) + 77
    sdev = torch.tensor([seed], device="cuda", dtype=torch.int64)
    k1 = ops.dropout_mask(4096, 0.1, seed, 9, "cuda")
    k2 = ops.dropout_mask(4096, 0.1, sdev.data_ptr(), 9 | L.SEED_ON_DEVICE, "cuda")
    assert torch.equal(k1, k2) and 0.05 < 1.0 - k1.float().mean().item() < 0.15
    M, N, K = 1280, 512, 256
    A = rnd(M, K, seed=5).cuda().bfloat16(); W = (rnd(N, K, seed=6) * K ** -0.5).cuda().bfloat16()
    bias, res = rnd(N, seed=7).cuda(), rnd(M, N, seed=8).cuda()
    outs = []
    for sd, st in ((seed, 3), (sdev.data_ptr(), 3 | L.SEED_ON_DEVICE)):
        o = torch.empty(M, N, device="cuda")
        ops.gemm_nt(A, W, o, epilogue=L.EPI_BIAS_RES_F32, bias=bias, res=res, p_drop=0.1, seed=sd, site=st)
        h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); gp = torch.empty_like(h)
        ops.gemm_nt(A, W, h, epilogue=L.EPI_BIAS_GELU, bias=bias, out2=gp, p_drop=0.1, seed=sd, site=st)
        outs.append((o, h, gp))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert not torch.equal(outs[0][0], res + (A.float() @ W.float().t() + bias))      # dropout really is on
    D = 512
    dy = rnd(64, D, seed=9).cuda().bfloat16(); x = rnd(64, D, seed=10).cuda(); dres = rnd(64, D, seed=11).cuda()
    gam = rnd(D, seed=12).cuda()
    _, mean, rstd = ops.layernorm_fwd(x, D, 64, D, gam, gam, 1e-5, torch.bfloat16)
    r1 = ops.layernorm_bwd(dy, x, D, gam, mean, rstd, dres, p_drop=0.1, seed=seed, site=5)
    r2 = ops.layernorm_bwd(dy, x, D, gam, mean, rstd, dres, p_drop=0.1, seed=sdev.data_ptr(), site=5 | L.SEED_ON_DEVICE)
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])


def test_pack_pad_batch_equals_single_packs(ops):
    srcs = [rnd(8, 512, seed=1).cuda(), rnd(2048, 8, seed=2).cuda(), rnd(16, 768, seed=3).cuda()]
    geoms = [(512, 1, 8, 512, 64, 512), (8, 1, 2048, 8, 2048, 64), (1, 768, 768, 16, 768, 32)]       # si, sj, rows, cols, rows_out, ld_out
    singles = [ops.pack_pad(s_, si, sj, r_, c_, ro, ld, torch.bfloat16) for s_, (si, sj, r_, c_, ro, ld) in zip(srcs, geoms)]
    outs = [torch.full_like(t, 7.0) for t in singles]
    table, mx = ops.pack_desc_table([(s_, si, sj, r_, c_, 1.0, o) for s_, (si, sj, r_, c_, ro, ld), o in zip(srcs, geoms, outs)], "cuda")
    ops.pack_pad_batch(table, len(outs), mx, torch.bfloat16)
    for a, b in zip(singles, outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,r", [(3000, 512, 8), (4096, 768, 16), (1000, 256, 4)])
def test_lora_grad_mfma_matches_valu_kernel(ops, M, N, r, monkeypatch):
    """The matrix-core reduction (N % 256 == 0, bf16) against the VALU kernel and an fp32 reference, ragged last slab included."""
    Y = rnd(M, N, seed=21).cuda().bfloat16()
    U = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16)
    U[:, :r] = rnd(M, r, seed=22).cuda().bfloat16()
    ref = Y.float().cpu().t() @ U[:, :r].float().cpu()
    got = {}
    from gslora_hip import _lib as L
    import contextlib
    for mode in ("1", "0"):      # "1": the product library's matrix-core kernel; "0": the VALU kernel, forced through the development build
        monkeypatch.setenv("GSL_LORA_GRAD_MFMA", mode)
        G = torch.zeros(N, r, device="cuda")
        with (L.use_dev() if mode == "0" else contextlib.nullcontext()):
            ops.lora_grad(Y, U, G, r, 1, r, accumulate=False)
            ops.lora_grad(Y, U, G, r, 1, r, accumulate=True)
        got[mode] = G.cpu() / 2
        assert relerr(got[mode], ref) < 2e-5, mode
    assert relerr(got["1"], got["0"]) < 2e-6


def test_data_prefetcher_ring_delivers_batches_in_order(ops):
    """Pinned-buffer ring on a copy stream: host batches arrive on the device in order and intact (more batches than ring slots, a
    ragged last batch), `(None, None)` marks exhaustion, device-resident batches pass through, prefetch=False takes the plain path."""
    from util.data_prefetcher import data_prefetcher
    host = [(torch.full((4 if i < 6 else 2, 3, 8, 8), float(i)), torch.full((4 if i < 6 else 2,), i, dtype=torch.int64)) for i in range(7)]
    for prefetch in (True, False):
        pf = data_prefetcher(host, torch.device("cuda"), prefetch=prefetch)
        seen = []
        while True:
            x, y = pf.next()
            if x is None:
                assert y is None
                break
            assert x.is_cuda and y.is_cuda
            seen.append((x.clone(), y.clone()))
        assert len(seen) == 7
        for i, (x, y) in enumerate(seen):
            assert x.shape[0] == (4 if i < 6 else 2) and bool((x == float(i)).all()) and bool((y == i).all())
        assert pf.next() == (None, None)
    dev_batches = [(torch.ones(2, 3, device="cuda") * i, torch.tensor([i, i], device="cuda")) for i in range(3)]
    pf = data_prefetcher(dev_batches, torch.device("cuda"), prefetch=True)
    x0, _ = pf.next()
    assert x0.data_ptr() == dev_batches[0][0].data_ptr()


@pytest.mark.parametrize("M,N,K1,K2", [(5713, 2048, 512, 64), (1031, 1536, 512, 0), (2600, 512, 2048, 0), (640, 256, 192, 0)])
def test_gemm_pingpong_variant(ops, monkeypatch, dev_lib, M, N, K1, K2):
    """The persistent ping-pong kernel (GSL_GEMM_VARIANT=10; measured, not the default — profiles/r02_pp_pingpong.md): STORE and
    BIAS_GELU (+ dropout) results against fp32 torch on the bf16-rounded operands, ragged M and N-tile counts of 8 / 6 / 2 / 1."""
    from gslora_hip import _lib as L
    monkeypatch.setenv("GSL_GEMM_VARIANT", "10")      # the kernel lives in the development build only
    dev_lib(L)
    dt = torch.bfloat16
    A1, W1 = rnd(M, K1, seed=11), rnd(N, K1, seed=12, scale=K1 ** -0.5)
    A2 = W2 = None
    acc = as_dt(A1, dt) @ as_dt(W1, dt).t()
    if K2:
        A2, W2 = rnd(M, K2, seed=13), rnd(N, K2, seed=14, scale=0.1)
        A2[:, 8:] = 0
        acc = acc + as_dt(A2, dt) @ as_dt(W2, dt).t()
    c = lambda t: None if t is None else t.cuda().to(dt)
    bias = rnd(N, seed=15)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(c(A1), c(W1), out, A2=c(A2), W2=c(W2), alpha=0.5, bias=bias.cuda())
    assert relerr(out.float().cpu(), 0.5 * acc + bias) < 1.5e-2
    out2 = torch.empty(M, N, device="cuda", dtype=dt)
    p, seed, site = 0.1, 991, 9
    ops.gemm_nt(c(A1), c(W1), out, epilogue=L.EPI_BIAS_GELU, A2=c(A2), W2=c(W2), bias=bias.cuda(), out2=out2, p_drop=p, seed=seed, site=site)
    keep = ops.dropout_mask(M * N, p, seed, site, "cuda").cpu().reshape(M, N).float()
    a = (acc + bias).requires_grad_(True)
    g = F.gelu(a)
    gp, = torch.autograd.grad(g.sum(), a)
    ref_h, ref_g = g.detach() * keep / (1 - p), gp * keep / (1 - p)
    hh, gg = out.float().cpu(), out2.float().cpu()
    assert (hh[keep == 0] == 0).all() and (gg[keep == 0] == 0).all()
    assert ((hh - ref_h).abs() - 2.0 ** -8 * ref_h.abs()).max() < 2e-3
    assert ((gg - ref_g).abs() - 2.0 ** -8 * ref_g.abs()).max() < 2e-3


@pytest.mark.parametrize("M,N,K1,K2", [(2560, 2048, 512, 64), (1536, 1536, 512, 0), (1280, 256, 512, 0)])
def test_gemm_inwave_pipelined_variant(ops, monkeypatch, dev_lib, M, N, K1, K2):
    """The in-wave software-pipelined kernel (GSL_GEMM_VARIANT=11; measured, not the default — profiles/r02_pp_pingpong.md): full tiles
    only (M % 128 == 0, N % 256 == 0, K in {512, 576}); N-tile counts 8 / 6 / 1, workgroups with one and with several tiles."""
    from gslora_hip import _lib as L
    monkeypatch.setenv("GSL_GEMM_VARIANT", "11")      # the kernel lives in the development build only
    dev_lib(L)
    dt = torch.bfloat16
    A1, W1 = rnd(M, K1, seed=21), rnd(N, K1, seed=22, scale=K1 ** -0.5)
    A2 = W2 = None
    acc = as_dt(A1, dt) @ as_dt(W1, dt).t()
    if K2:
        A2, W2 = rnd(M, K2, seed=23), rnd(N, K2, seed=24, scale=0.1)
        A2[:, 8:] = 0
        acc = acc + as_dt(A2, dt) @ as_dt(W2, dt).t()
    c = lambda t: None if t is None else t.cuda().to(dt)
    bias = rnd(N, seed=25)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(c(A1), c(W1), out, A2=c(A2), W2=c(W2), alpha=0.5, bias=bias.cuda())
    assert relerr(out.float().cpu(), 0.5 * acc + bias) < 1.5e-2
    out2 = torch.empty(M, N, device="cuda", dtype=dt)
    p, seed, site = 0.1, 1234, 13
    ops.gemm_nt(c(A1), c(W1), out, epilogue=L.EPI_BIAS_GELU, A2=c(A2), W2=c(W2), bias=bias.cuda(), out2=out2, p_drop=p, seed=seed, site=site)
    keep = ops.dropout_mask(M * N, p, seed, site, "cuda").cpu().reshape(M, N).float()
    a = (acc + bias).requires_grad_(True)
    g = F.gelu(a)
    gp, = torch.autograd.grad(g.sum(), a)
    ref_h, ref_g = g.detach() * keep / (1 - p), gp * keep / (1 - p)
    hh, gg = out.float().cpu(), out2.float().cpu()
    assert (hh[keep == 0] == 0).all() and (gg[keep == 0] == 0).all()
    assert ((hh - ref_h).abs() - 2.0 ** -8 * ref_h.abs()).max() < 2e-3
    assert ((gg - ref_g).abs() - 2.0 ** -8 * ref_g.abs()).max() < 2e-3


def test_layernorm_bwd_bf16_gradient_stream(ops):
    """Speed mode carries the residual-gradient stream (dres in, dx out) in bf16: same arithmetic in f32 registers, one bf16 rounding of dx
    (ADVICE / VERDICT r01 #4: the f32 stream was 2 x 413 MB of every LayerNorm backward). Against the f32-stream kernel on the same inputs."""
    M, D = 301, 512
    dt = torch.bfloat16
    x, g = rnd(M, D, seed=41, scale=2.0) + 0.5, 1 + 0.1 * rnd(D, seed=42)
    dy, dres = rnd(M, D, seed=43), rnd(M, D, seed=44)
    _, mean, rstd = ops.layernorm_fwd(x.cuda(), D, M, D, g.cuda(), torch.zeros(D).cuda(), 1e-5, dt)
    dres_b = dres.cuda().to(dt)
    dx32, dxb32 = ops.layernorm_bwd(dy.cuda().to(dt), x.cuda(), D, g.cuda(), mean, rstd, dres_b.float(), p_drop=0.1, seed=5, site=3)
    dx16, dxb16 = ops.layernorm_bwd(dy.cuda().to(dt), x.cuda(), D, g.cuda(), mean, rstd, dres_b, p_drop=0.1, seed=5, site=3)
    assert dx16.dtype == dt and dx32.dtype == torch.float32
    assert torch.equal(dx16, dx32.to(dt))            # the same f32 value, rounded once
    assert torch.equal(dxb16, dxb32)                 # the masked operand copy does not depend on the stream dtype
    dxh, _ = ops.head_bwd(None, rnd(4, D, seed=45).cuda(), rnd(4 * 7, D, seed=46).cuda(), 4, 7, D, g.cuda(), torch.zeros(4).cuda(),
                          torch.ones(4).cuda(), rnd(4, D, seed=47).cuda(), None, 64.0, dt, stream_dtype=dt)
    assert dxh.dtype == dt and (dxh.view(4, 7, D)[:, 1:] == 0).all()


# ---------------------------------------------------------------------------------------------------------------- head-major qkv
def _to_head_major(qkv, B, T, H):
    """[B*T, 3*H*64] (q|k|v, head-major inside each) -> [B][H][3][T][64] flattened back to the same 2-D shape."""
    return qkv.view(B, T, 3, H, 64).permute(0, 3, 2, 1, 4).contiguous().view(B * T, 3 * H * 64)


@pytest.mark.parametrize("B,T,H,K2", [(3, 197, 8, 0), (40, 197, 8, 0), (7, 50, 12, 64), (2, 13, 2, 0), (110, 197, 8, 0)])
def test_gemm_store_qkv_head_major_is_a_permutation_of_the_plain_store(ops, B, T, H, K2):
    """EPI_STORE_QKV_HM writes exactly the values of EPI_STORE, at [b][h][which][t][64] (every tile kernel: 128x128 at small M, the 8-phase
    256x256 kernel from ~150 tiles on; ragged last tiles)."""
    from gslora_hip import _lib as L
    M, N, K = B * T, 3 * H * 64, 512
    A, W = rnd(M, K, seed=1).cuda().bfloat16(), rnd(N, K, seed=2, scale=K ** -0.5).cuda().bfloat16()
    A2 = W2 = None
    if K2:
        A2, W2 = rnd(M, K2, seed=3).cuda().bfloat16(), rnd(N, K2, seed=4, scale=0.1).cuda().bfloat16()
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(A, W, ref, A2=A2, W2=W2)
    ops.gemm_nt(A, W, out, A2=A2, W2=W2, epilogue=L.EPI_STORE_QKV_HM, T=T)
    assert torch.equal(out, _to_head_major(ref, B, T, H))


@pytest.mark.parametrize("B,T,H", [(3, 197, 8), (2, 26, 1), (70, 197, 8), (5, 150, 4), (3, 224, 2), (2, 64, 2)])
def test_attention_head_major_input_bit_identical_to_token_major(ops, B, T, H):
    """Forward, backward and the cls-row backward of the bf16 attention kernels read the head-major qkv layout with the same arithmetic
    in the same order: o, lse and (token-major) dqkv are bit-identical."""
    scale = 64 ** -0.5
    qkv = (rnd(B * T, 3 * H * 64, seed=5)).cuda().bfloat16()
    d_o = rnd(B * T, H * 64, seed=6).cuda().bfloat16()
    hm = _to_head_major(qkv, B, T, H)
    o0, l0 = ops.attention_fwd(qkv, B, T, H, scale)
    o1, l1 = ops.attention_fwd(hm, B, T, H, scale, layout=1)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    g0 = ops.attention_bwd(qkv, o0, d_o, l0, B, T, H, scale)
    g1 = ops.attention_bwd(hm, o0, d_o, l0, B, T, H, scale, layout=1)
    assert torch.equal(g0, g1)
    d_cls = d_o.view(B, T, H * 64)[:, 0].contiguous()
    c0 = ops.attention_bwd_cls(qkv, o0, d_cls, l0, B, T, H, scale)
    c1 = ops.attention_bwd_cls(hm, o0, d_cls, l0, B, T, H, scale, layout=1)
    assert torch.equal(c0, c1)


# ---------------------------------------------------------------------------------------------------------------- bf16 forward stream
@pytest.mark.parametrize("sdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K1,K2,T", [(591, 192, 128, 64, 197), (257, 512, 512, 0, 257), (788, 512, 2048, 64, 197),
                                         (33490, 512, 512, 0, 197), (33490, 512, 192, 0, 197)])
def test_gemm_bf16_stream_epilogues_equal_the_rounded_f32_stream_epilogues(ops, M, N, K1, K2, T, sdt):
    """BIAS_RES_BF16 / PATCH_BF16 (bf16 speed mode, forward residual stream in bf16): with a residual that is exactly representable in
    bf16 both epilogues compute the same f32 value — the bf16-stream output must be its one-time rounding, bit for bit, and the dropout
    mask the same. Small shapes run the 128x128 kernel's fragment path, the 33 490-row shapes the 8-phase kernel's staged full-row
    epilogue (ragged last M tile). sdt = the stream's element type: bf16 (round 3) or IEEE fp16 (round 4: GSL_EPI_BIAS_RES_F16 /
    PATCH_F16, the same kernels with the conversion switched at run time; round to nearest even, clamped at +-65504)."""
    from gslora_hip import _lib as L
    dt = torch.bfloat16
    c = lambda t: None if t is None else t.cuda().to(dt)
    A1, W1 = c(rnd(M, K1, seed=1)), c(rnd(N, K1, seed=2, scale=K1 ** -0.5))
    A2 = W2 = None
    if K2:
        a2 = rnd(M, K2, seed=3); a2[:, 8:] = 0
        A2, W2 = c(a2), c(rnd(N, K2, seed=4, scale=0.1))
    bias, res = rnd(N, seed=5).cuda(), rnd(M, N, seed=6).cuda().to(sdt)
    epi_res, epi_patch = (L.EPI_BIAS_RES_F16, L.EPI_PATCH_F16) if sdt == torch.float16 else (L.EPI_BIAS_RES_BF16, L.EPI_PATCH_BF16)
    for p in (0.0, 0.1):
        o32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
        o16 = torch.full((M, N), 7.0, device="cuda", dtype=sdt)
        ops.gemm_nt(A1, W1, o32, epilogue=L.EPI_BIAS_RES_F32, A2=A2, W2=W2, bias=bias, res=res.float(), p_drop=p, seed=11, site=3)
        ops.gemm_nt(A1, W1, o16, epilogue=epi_res, A2=A2, W2=W2, bias=bias, res=res, p_drop=p, seed=11, site=3)
        assert torch.equal(o16, o32.to(sdt)), p
        pos, cls = rnd(T, N, seed=8).cuda(), rnd(N, seed=9).cuda()
        if M % T == 0:
            ops.gemm_nt(A1, W1, o32, epilogue=L.EPI_PATCH, A2=A2, W2=W2, bias=bias, pos=pos, cls=cls, T=T, p_drop=p, seed=12, site=1_000_000)
            ops.gemm_nt(A1, W1, o16, epilogue=epi_patch, A2=A2, W2=W2, bias=bias, pos=pos, cls=cls, T=T, p_drop=p, seed=12, site=1_000_000)
            assert torch.equal(o16, o32.to(sdt)), p
    if sdt == torch.float16:      # beyond fp16's range the stream saturates at +-65504 instead of turning into inf
        big = torch.full((M, N), 60000.0, device="cuda", dtype=sdt)
        ops.gemm_nt(A1, W1, o16, epilogue=epi_res, A2=A2, W2=W2, bias=bias + 30000.0, res=big)
        assert torch.isfinite(o16.float()).all() and float(o16.float().max()) == 65504.0
        # ... but a NaN / Inf that arrives stays one (a software v_med3 clamp turned NaN into -65504 and hid a diverged step;
        # the stores now saturate through MODE.FP16_OVFL, which keeps Inf and NaN)
        big[0, 0], big[1, 1], big[2, 2] = float("nan"), float("inf"), float("-inf")
        ops.gemm_nt(A1, W1, o16, epilogue=epi_res, A2=A2, W2=W2, bias=bias + 30000.0, res=big)
        assert torch.isnan(o16[0, 0]) and o16[1, 1] == float("inf") and o16[2, 2] == float("-inf")
        o16[0, 0] = o16[1, 1] = o16[2, 2] = 0
        assert torch.isfinite(o16.float()).all()


def test_gemm_nt_lora_bf16_stream_epilogue(ops):
    """The in-kernel-LoRA GEMM (FFN2 forward) with the bf16 residual stream == its f32-stream result, rounded once."""
    from gslora_hip import _lib as L
    dt = torch.bfloat16
    M, N, K, r = 2100, 512, 2048, 8
    c = lambda t: t.cuda().to(dt)
    A, W = c(rnd(M, K, seed=1)), c(rnd(N, K, seed=2, scale=K ** -0.5))
    P = torch.zeros(16, K); P[:r] = rnd(r, K, seed=3, scale=K ** -0.5)
    Q = torch.zeros(N, 32); Q[:, :r] = rnd(N, r, seed=4, scale=0.3)
    bias = rnd(N, seed=5).cuda()
    for sdt, epi in ((dt, L.EPI_BIAS_RES_BF16), (torch.float16, L.EPI_BIAS_RES_F16)):      # the stream in bf16 / in fp16
        res = rnd(M, N, seed=6).cuda().to(sdt)
        t32 = torch.empty(M, 64, device="cuda", dtype=dt); t16 = torch.empty(M, 64, device="cuda", dtype=dt)
        o32 = torch.empty(M, N, device="cuda", dtype=torch.float32); o16 = torch.empty(M, N, device="cuda", dtype=sdt)
        ops.gemm_nt_lora(A, W, c(P), c(Q), 1.0 / r, t32, o32, epilogue=L.EPI_BIAS_RES_F32, bias=bias, res=res.float(), p_drop=0.1, seed=5, site=2)
        ops.gemm_nt_lora(A, W, c(P), c(Q), 1.0 / r, t16, o16, epilogue=epi, bias=bias, res=res, p_drop=0.1, seed=5, site=2)
        assert torch.equal(o16, o32.to(sdt)) and torch.equal(t16, t32)


@pytest.mark.parametrize("sdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [128, 512, 768])
def test_layernorm_with_a_bf16_residual_stream(ops, D, sdt):
    """LayerNorm forward / backward reading the residual stream x in bf16 / fp16 == the same kernels on the widened f32 copy of that tensor."""
    dt = torch.bfloat16
    M = 301
    x = (rnd(M, D, seed=41, scale=2.0) + 0.5).cuda().to(sdt)
    g, b = (1 + 0.1 * rnd(D, seed=42)).cuda(), (0.1 * rnd(D, seed=43)).cuda()
    y16, m16, r16 = ops.layernorm_fwd(x, D, M, D, g, b, 1e-5, dt)
    y32, m32, r32 = ops.layernorm_fwd(x.float(), D, M, D, g, b, 1e-5, dt)
    assert torch.equal(y16, y32) and torch.equal(m16, m32) and torch.equal(r16, r32)
    dy, dres = rnd(M, D, seed=44).cuda().to(dt), rnd(M, D, seed=45).cuda().to(dt)
    a = ops.layernorm_bwd(dy, x, D, g, m16, r16, dres, p_drop=0.1, seed=5, site=3)
    bq = ops.layernorm_bwd(dy, x.float(), D, g, m16, r16, dres, p_drop=0.1, seed=5, site=3)
    for u, v in zip(a, bq):      # same arithmetic; the two template instantiations may contract multiply-adds differently: within one bf16 ulp
        assert (u.float() - v.float()).abs().max() <= 2.0 ** -7 * max(1.0, v.float().abs().max().item())
        assert (u != v).float().mean() < 0.01


@pytest.mark.parametrize("dt", DTS)
def test_layernorm_bwd_compact_cls_residual_gradient(ops, dt):
    """dres_cls_T: the incoming stream gradient is given as its cls rows only ([B, D]); the result must equal the dense call on the
    zero-filled [B*T, D] tensor, bit for bit."""
    B, T, D = 5, 7, 128
    M = B * T
    sdt = dt
    x = rnd(M, D, seed=1, scale=2.0).cuda().to(sdt)
    g = (1 + 0.1 * rnd(D, seed=2)).cuda()
    _, mean, rstd = ops.layernorm_fwd(x, D, M, D, g, torch.zeros(D).cuda(), 1e-5, dt)
    dy = rnd(M, D, seed=3).cuda().to(dt)
    compact = rnd(B, D, seed=4).cuda().to(sdt)
    dense = torch.zeros(M, D, device="cuda", dtype=sdt)
    dense.view(B, T, D)[:, 0] = compact
    a = ops.layernorm_bwd(dy, x, D, g, mean, rstd, dense, p_drop=0.2, seed=9, site=4)
    b = ops.layernorm_bwd(dy, x, D, g, mean, rstd, compact, p_drop=0.2, seed=9, site=4, dres_cls_T=T)
    assert b[0].shape == (M, D) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("dt", DTS)
def test_head_bwd_compact_equals_the_cls_rows_of_the_dense_form(ops, dt):
    B, T, D, C = 6, 9, 128, 10
    sdt = dt
    x = rnd(B * T, D, seed=1, scale=1.5).cuda().to(sdt)
    g, b = (1 + 0.1 * rnd(D, seed=2)).cuda(), (0.1 * rnd(D, seed=3)).cuda()
    Wn = ops.cosface_prep(rnd(C, D, seed=4).cuda())
    label = (torch.arange(B) % C).cuda()
    logits, emb, mean, rstd = ops.head_fwd(x, B, T, D, g, b, 1e-5, Wn, label, 64.0, 0.35)
    if dt == torch.bfloat16:      # the bf16 stream is read as the widened values
        l32, e32, _, _ = ops.head_fwd(x.float(), B, T, D, g, b, 1e-5, Wn, label, 64.0, 0.35)
        assert torch.equal(logits, l32) and torch.equal(emb, e32)
    dl, de = rnd(B, C, seed=5).cuda(), rnd(B, D, seed=6).cuda()
    kw = dict(p_drop=0.3, seed=21, site=6, stream_dtype=sdt)
    dx, dxb = ops.head_bwd(dl, de, x, B, T, D, g, mean, rstd, emb, Wn, 64.0, dt, **kw)
    cx, cxb = ops.head_bwd(dl, de, x, B, T, D, g, mean, rstd, emb, Wn, 64.0, dt, compact=True, **kw)
    assert cx.shape == (B, D) and cxb.shape == (B, D)
    assert torch.equal(cx, dx.view(B, T, D)[:, 0]) and torch.equal(cxb, dxb.view(B, T, D)[:, 0])
    assert (dx.view(B, T, D)[:, 1:] == 0).all()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,T,H,hm", [(3, 197, 2, 0), (2, 26, 1, 0), (5, 197, 8, 1), (2, 224, 2, 1), (130, 197, 8, 1)])
def test_attention_fwd_cls_equals_the_cls_rows_of_the_dense_forward(ops, dt, B, T, H, hm):
    """The last block's attention for the cls query alone: o / lse of token 0 of every (image, head), against fp32 torch and against the
    dense kernel; the cls backward accepts its compact o / lse and returns what it returns for the dense forward's tensors."""
    if hm and dt != torch.bfloat16:
        pytest.skip("head-major qkv is a bf16-mode layout")
    scale = 64 ** -0.5
    qkv = rnd(B * T, 3 * H * 64, seed=61, scale=1.2).cuda().to(dt)
    qin = _to_head_major(qkv, B, T, H) if hm else qkv
    o_d, lse_d = ops.attention_fwd(qin, B, T, H, scale, layout=hm)
    o_c, lse_c = ops.attention_fwd_cls(qin, B, T, H, scale, layout=hm)
    assert o_c.shape == (B, H * 64) and lse_c.shape == (B, H)
    q, k, v = [t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in qkv.float().cpu().chunk(3, -1)]
    s0 = torch.einsum("bhd,bhjd->bhj", q[:, :, 0], k) * scale
    ref_o = torch.einsum("bhj,bhjd->bhd", s0.softmax(-1), v).reshape(B, H * 64)
    assert (o_c.float().cpu() - ref_o).abs().max() < tol(dt, 2e-5, 2e-2)
    assert (lse_c.cpu() - s0.logsumexp(-1)).abs().max() < 2e-4
    assert (o_c.float() - o_d.view(B, T, -1)[:, 0].float()).abs().max() < tol(dt, 2e-5, 3e-2)
    assert (lse_c - lse_d[:, :, 0]).abs().max() < tol(dt, 2e-5, 2e-3)
    d_o = rnd(B, H * 64, seed=62).cuda().to(dt)
    a = ops.attention_bwd_cls(qin, o_c, d_o, lse_c, B, T, H, scale, layout=hm)
    dense_o = torch.zeros(B * T, H * 64, device="cuda", dtype=dt); dense_o.view(B, T, -1)[:, 0] = o_c
    dense_l = torch.zeros(B, H, T, device="cuda"); dense_l[:, :, 0] = lse_c
    b = ops.attention_bwd_cls(qin, dense_o, d_o, dense_l, B, T, H, scale, layout=hm)
    assert torch.equal(a, b)
    # layout 2: K | V token-major + the cls queries on their own (the last block projects Q for the cls rows only): same arithmetic
    inner = H * 64
    kv = qkv[:, inner:].contiguous()
    q_cls = qkv.view(B, T, 3 * inner)[:, 0, :inner].contiguous()
    o2, lse2 = ops.attention_fwd_cls(kv, B, T, H, scale, layout=2, q_cls=q_cls)
    assert torch.equal(o2, o_c) and torch.equal(lse2, lse_c)
    dkv, dq = ops.attention_bwd_cls(kv, o2, d_o, lse2, B, T, H, scale, layout=2, q_cls=q_cls)
    assert torch.equal(dkv, a[:, inner:]) and torch.equal(dq, a.view(B, T, 3 * inner)[:, 0, :inner])


# ---------------------------------------------------------------------------------------------------------------- 8-bit GELU'
def _unslab(q):
    """slab-major [N/64][M][64] code tensor (stored in a [M, N] uint8 buffer) -> row-major [M, N]"""
    M, N = q.shape
    return q.view(N // 64, M, 64).permute(1, 0, 2).reshape(M, N)


def _slab(q):
    M, N = q.shape
    return q.view(M, N // 64, 64).permute(1, 0, 2).contiguous().view(M, N)


@pytest.mark.parametrize("M,N,K1,K2,p", [(130, 128, 64, 0, 0.1), (591, 256, 128, 64, 0.0), (788, 2048, 512, 64, 0.1), (33490, 2048, 512, 64, 0.1),
                                         (2600, 2048, 512, 0, 0.25), (33490, 2112, 512, 0, 0.25)])
def test_gemm_gelu_g8_code_of_the_derivative(ops, M, N, K1, K2, p):
    """GSL_EPI_BIAS_GELU_G8: the first output h equals BIAS_GELU's (bit for bit on the fragment-path kernels, within the table band on the
    8-phase kernel); the second is the 8-bit fixed-point code of
    gelu'(a) * keep — q = round(gelu' keep 200 + 26), decoded (q - 26) 0.005 / (1 - p): within half a step (0.0025 / (1 - p)) of the
    bf16 kernel's own f32 value (compared through BIAS_GELU's bf16 output: + its rounding), a dropped element decodes to exactly 0,
    and the code error has no bias. The code tensor is slab-major [N/64][M][64]. Fragment paths (64x64 and 128x128 kernels) and the 8-phase
    kernel's staged byte path."""
    from gslora_hip import _lib as L
    dt = torch.bfloat16
    c = lambda t: None if t is None else t.cuda().to(dt)
    A1, W1 = c(rnd(M, K1, seed=1)), c(rnd(N, K1, seed=2, scale=K1 ** -0.5))
    A2 = W2 = None
    if K2:
        a2 = rnd(M, K2, seed=3); a2[:, 8:] = 0
        A2, W2 = c(a2), c(rnd(N, K2, seed=4, scale=0.1))
    bias = rnd(N, seed=5).cuda()
    h0 = torch.empty(M, N, device="cuda", dtype=dt); g0 = torch.empty(M, N, device="cuda", dtype=dt)
    h1 = torch.empty(M, N, device="cuda", dtype=dt); q = torch.full((M, N), 255, device="cuda", dtype=torch.uint8)
    ops.gemm_nt(A1, W1, h0, epilogue=L.EPI_BIAS_GELU, A2=A2, W2=W2, bias=bias, out2=g0, p_drop=p, seed=7, site=5)
    ops.gemm_nt(A1, W1, h1, epilogue=L.EPI_BIAS_GELU_G8, A2=A2, W2=W2, bias=bias, out2=q, p_drop=p, seed=7, site=5)
    # h: the fragment-path kernels share BIAS_GELU's arithmetic (bit-identical); the 8-phase kernel (round 4) takes Phi(a) from the 4096-entry
    # LDS table instead of the A&S erf: |a| * |Phi_tab - Phi| <= max|a phi(a)| * D / 2 = 0.242 * 0.0011 + rounding = 2.8e-4 (x 1 / (1 - p)) on top of
    # one bf16 rounding of each of the two values compared
    tab = 2.8e-4 / (1 - p)
    dh = (h1.float() - h0.float()).abs()
    assert (dh - 2.0 ** -7 * h0.float().abs()).max() <= tab * 1.05
    assert (h1 != h0).float().mean() < 0.2          # (most elements still agree bit for bit)
    qs, q = q, _unslab(q)          # the code tensor is slab-major [N/64][M][64]
    assert int(q.max()) <= 252
    dec = (q.float() - 26.0) * (0.005 / (1 - p))
    keep = ops.dropout_mask(M * N, p, 7, 5, "cuda").reshape(M, N).bool() if p > 0 else torch.ones(M, N, device="cuda", dtype=torch.bool)
    assert (q[~keep] == 26).all() and (dec[~keep] == 0).all()
    assert (h1[~keep] == 0).all()
    err = dec - g0.float()
    half = 0.0025 / (1 - p)
    gtab = 0.8 * 0.0011 / (1 - p)                   # table path: GELU' at the nearest grid point, |GELU''| <= 0.8, D / 2 = 0.0011
    assert (err.abs() - 2.0 ** -8 * g0.float().abs()).max() <= (half + gtab) * 1.02
    assert abs(err[keep].mean().item()) < 0.1 * half                       # round-to-nearest: no bias beyond the saturated tails (gelu' -> 0-, 1-)
    # MUL_G8 decodes it: (A W^T) * decode(q) in f32, one bf16 rounding
    acc = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt(A1, W1, acc, epilogue=L.EPI_STORE_F32, A2=A2, W2=W2)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(A1, W1, out, epilogue=L.EPI_MUL_G8, A2=A2, W2=W2, aux=qs, p_drop=p)
    want = acc * dec
    assert ((out.float() - want).abs() - 2.0 ** -8 * want.abs()).max() < 1e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_gelu_table_epilogue_dropped_elements_are_exact_zeros_for_any_finite_preactivation(ops, dt):
    """the table epilogue of the 8-phase kernel multiplies a / (1 - p) by the table entry, and a dropped element's entry is the
    bit pattern 0x0000001A (code 26 | a float DENORMAL). With f32 denormals flushed for the epilogue the product is exactly +-0 for every
    finite pre-activation — here |a| up to 3e4 — and a non-finite pre-activation stays non-finite (kept: +-Inf / NaN; dropped: NaN, as
    torch's Inf * 0), never a silent finite value. Kept large values: GELU(a) = a for a >> 0, ~0 for a << 0 (the table clamps at +-4.5)."""
    from gslora_hip import _lib as L
    M, N, K, p = 33490, 2048, 512, 0.1            # the 8-phase kernel (table path): full and partial M tiles
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    A[1::7] *= 40.0                                # |a| up to ~200
    bias = torch.randn(N, generator=g)
    bias[3], bias[64 + 5], bias[300], bias[301], bias[700] = 3.0e4, -3.0e4, float("inf"), float("-inf"), float("nan")
    h = torch.empty(M, N, device="cuda", dtype=dt); q = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    ops.gemm_nt(A.cuda().to(dt), W.cuda().to(dt), h, epilogue=L.EPI_BIAS_GELU_G8, bias=bias.cuda(), out2=q, p_drop=p, seed=11, site=2)
    keep = ops.dropout_mask(M * N, p, 11, 2, "cuda").reshape(M, N).bool()
    q = _unslab(q)
    finite_col = torch.ones(N, dtype=torch.bool, device="cuda"); finite_col[[300, 301, 700]] = False
    hd = h[:, finite_col][~keep[:, finite_col]]
    assert hd.numel() > 1000 and (hd == 0).all()                          # exactly +-0, also at a = +-3e4
    assert (q[~keep] == 26).all()
    for col in (300, 301, 700):
        assert not torch.isfinite(h[:, col].float()).any(), col          # kept or dropped: never a finite value
    big = h[:, 3].float()[keep[:, 3]]
    assert ((big / (3.0e4 / (1 - p)) - 1).abs() < 2e-2).all()             # GELU(a) = a
    neg = h[:, 64 + 5].float()[keep[:, 64 + 5]]
    assert (neg.abs() <= 3.0e4 / (1 - p) * 3.5e-6 * 1.05).all()           # a * Phi(-4.5): the table's clamp, |error| <= |a| 3.4e-6


@pytest.mark.parametrize("M,N,K,r,p", [(1000, 512, 256, 8, 0.1), (4099, 2048, 512, 8, 0.1), (1300, 640, 192, 5, 0.0)])
def test_gemm_nt_lora_mulgrad_with_the_8bit_gelu_derivative(ops, M, N, K, r, p):
    """gsl_gemm_nt_lora_mulgrad(aux_u8): `out` / `tout` bit-identical to the unfused in-kernel-LoRA GEMM with GSL_EPI_MUL_G8, gradients
    equal to gsl_lora_grad on the same tensors — the fused reductions see the decoded multiplier."""
    from gslora_hip import _lib as L
    dt = torch.bfloat16
    c = lambda t: t.cuda().to(dt)
    A, W = c(rnd(M, K, seed=1)), c(rnd(N, K, seed=2, scale=K ** -0.5))
    P = torch.zeros(16, K); P[:r] = rnd(r, K, seed=3, scale=K ** -0.5)
    Q = torch.zeros(N, 32); Q[:, :r] = rnd(N, r, seed=4, scale=0.3)
    P, Q = c(P), c(Q)
    g = torch.Generator().manual_seed(5)
    q = torch.randint(0, 253, (M, N), generator=g, dtype=torch.uint8).cuda()      # the slab-major buffer; its row-major meaning is _unslab(q)
    Y2 = c(rnd(M, N, seed=8))
    U1 = torch.zeros(M, 64); U1[:, :r] = rnd(M, r, seed=9)
    U1 = c(U1)
    s = 1.0 / r
    tout0 = torch.empty(M, 64, device="cuda", dtype=dt); out0 = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt_lora(A, W, P, Q, s, tout0, out0, epilogue=L.EPI_MUL_G8, aux=q, p_drop=p)
    G1r = torch.zeros(N, r, device="cuda"); G2r = torch.zeros(r, N, device="cuda")
    ops.lora_grad(out0, U1, G1r, r, 1, r, accumulate=False)
    ops.lora_grad(Y2, tout0, G2r, 1, N, r, accumulate=False)
    tout = torch.full((M, 64), 3.0, device="cuda", dtype=dt); out = torch.empty(M, N, device="cuda", dtype=dt)
    G1 = torch.zeros(N, r, device="cuda"); G2 = torch.zeros(r, N, device="cuda")
    ops.gemm_nt_lora_mulgrad(A, W, P, Q, s, tout, out, q, U1, G1, (r, 1), Y2, G2, (1, N), r, accumulate=False, p_drop=p)
    assert torch.equal(out, out0) and torch.equal(tout, tout0)
    assert (G1 - G1r).abs().max().item() < 2e-5 * G1r.abs().max().item() and (G2 - G2r).abs().max().item() < 2e-5 * G2r.abs().max().item()
    dec = (_unslab(q).float() - 26.0) * (0.005 / (1 - p))
    acc = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt_lora(A, W, P, Q, s, None, acc.to(dt), epilogue=L.EPI_STORE)      # (shape check of the plain form)
    plain = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt_lora(A, W, P, Q, s, None, plain, epilogue=L.EPI_STORE)
    assert relerr((out0.float()).cpu(), (plain.float() * dec).cpu()) < 1.5e-2


@pytest.mark.parametrize("M,D,r", [(1000, 512, 8), (37, 512, 8), (16 * 300 + 5, 768, 16), (1, 512, 4)])
def test_layernorm_fwd_with_lora_down_projection(ops, M, D, r):
    """gsl_layernorm_fwd_lora: LayerNorm + u = alpha * LN(x) P^T in one pass over the bf16 stream, against the two launches it replaces
    (gsl_layernorm_fwd, then the skinny gsl_gemm_nt over its output) and against torch fp32. Ragged row counts exercise the partial
    16-row group; the 64-column K-segment buffer must come back with columns >= 16 zeroed (it is torch.empty)."""
    dt = torch.bfloat16
    x = (rnd(M, D, seed=1, scale=2.0) + 0.5).to(dt).cuda()
    g, b = (1 + 0.1 * rnd(D, seed=2)).cuda(), (0.1 * rnd(D, seed=3)).cuda()
    P = torch.zeros(64, D); P[:r] = rnd(r, D, seed=4, scale=D ** -0.5)
    P = P.to(dt).cuda()
    alpha = 1.0 / r
    torch.empty(M, 64, device="cuda", dtype=dt).fill_(7.0)          # poison the allocator's next [M, 64] block
    y, mean, rstd, u = ops.layernorm_fwd_lora(x, D, M, D, g, b, 1e-5, P, alpha)
    y0, mean0, rstd0 = ops.layernorm_fwd(x, D, M, D, g, b, 1e-5, dt)
    u0 = torch.empty(M, 64, device="cuda", dtype=dt)
    ops.gemm_nt(y0, P, u0, alpha=alpha)
    ref = F.layer_norm(x.float().cpu(), (D,), g.cpu(), b.cpu(), 1e-5)
    assert (y.float().cpu() - ref).abs().max() < 3e-2
    assert (mean - mean0).abs().max() < 1e-5 and relerr(rstd, rstd0) < 1e-5
    # the two kernels sum the row in different orders: at most one bf16 ulp apart, and only where the f32 value sits on a rounding edge
    d = (y.float() - y0.float()).abs()
    assert (d <= y0.float().abs() * 2.0 ** -7 + 1e-6).all() and (d > 0).float().mean() < 0.02
    assert torch.count_nonzero(u[:, 16:]) == 0 and torch.count_nonzero(u[:, r:16]) == 0
    u_ref = alpha * (y.float() @ P[:16].float().t())                 # from the bf16 rows the kernel itself wrote
    assert (u[:, :16].float() - u_ref).abs().max() <= u_ref.abs().max() * 2.0 ** -8 + 1e-6
    assert (u.float() - u0.float()).abs().max() < 2e-2 * max(1.0, u0.float().abs().max().item())


def test_lora_grad_batch_equals_the_single_reductions(ops):
    """gsl_lora_grad_batch: a backward pass's LoRA-gradient reductions in two launches (descriptors by value in the kernel arguments)
    against one gsl_lora_grad per reduction and torch fp32. Shapes of a few-shot step: dense rows, the 8 cls rows of the last block, both
    output strides, r = 8 and 16, accumulate on and off, column blocks of wider tensors."""
    dt = torch.bfloat16
    cases = [(1576, 512, 8, True), (1576, 2048, 8, True), (8, 512, 8, True), (8, 2048, 8, False), (700, 768, 16, True), (33, 256, 4, False),
             (5000, 512, 8, True)]
    ents, single, refs = [], [], []
    for k, (M, N, r, acc) in enumerate(cases):
        Yw = rnd(M, N + 256, seed=10 + k).to(dt).cuda()
        Y = Yw[:, 256:] if k % 2 else Yw[:, :N]                      # a column block of a wider tensor
        U = torch.zeros(M, 64, dtype=dt, device="cuda"); U[:, :r] = rnd(M, r, seed=40 + k).to(dt).cuda()
        g0 = rnd(N * r, seed=70 + k).cuda()
        gsn, gsj = (r, 1) if k % 3 else (1, N)
        G1, G2 = g0.clone(), g0.clone()
        ents.append((Y, U, G1, gsn, gsj, r, acc))
        single.append((Y, U, G2, gsn, gsj, r, acc))
        want = Y.float().t() @ U[:, :r].float()                       # [N, r]
        want = want.reshape(-1) if (gsn, gsj) == (r, 1) else want.t().reshape(-1)
        refs.append(want + (g0 if acc else 0))
    assert all(ops.lora_grad_batchable(e[0], e[1], e[5]) for e in ents)
    ops.lora_grad_batch(ents)
    for (Y, U, G, gsn, gsj, r, acc) in single:
        ops.lora_grad(Y, U, G, gsn, gsj, r, accumulate=acc)
    for k, (e, s1, ref) in enumerate(zip(ents, single, refs)):
        scale = ref.abs().max().item()
        assert (e[2] - ref).abs().max().item() < 2e-5 * max(1.0, scale) * cases[k][0] ** 0.5, k
        assert (e[2] - s1[2]).abs().max().item() < 1e-5 * max(1.0, scale), k      # same partials, at most a different summation order
    # more descriptors than one launch carries (24)
    many = [(ents[0][0], ents[0][1], torch.zeros(512 * 8, device="cuda"), 8, 1, 8, False) for _ in range(30)]
    ops.lora_grad_batch(many)
    # (the two launches split the rows differently: same sums up to the f32 summation order)
    assert all(torch.allclose(m[2], many[0][2], rtol=1e-5, atol=1e-4) for m in many) and many[0][2].abs().max() > 0


@pytest.mark.parametrize("N,nr,C,D,proto,struct", [(8, 4, 100, 512, True, True), (96, 48, 100, 768, True, False), (37, 5, 12, 128, False, True),
                                                  (256, 255, 100, 512, True, True)])
def test_loss_tail_equals_the_separate_kernels(ops, N, nr, C, D, proto, struct):
    """gsl_loss_tail (the loss section of a single-process step in one launch) against gsl_ce_fwd / gsl_proto_kl_fwd / gsl_loss_combine /
    gsl_ce_bwd / gsl_proto_kl_bwd on the two row ranges: the same meters, bit-identical coefficients and gradients, for active and inactive hinges."""
    logits = (rnd(N, C, seed=1, scale=3.0)).cuda()
    labels = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(2)).cuda()
    emb = rnd(N, D, seed=3).cuda() if proto else None
    table = rnd(C, D, seed=4).cuda() if proto else None
    st = torch.tensor(13.5, device="cuda") if struct else None
    for BND, BND_pro in ((105.0, 50.0), (0.5, 1e-4)):      # hinges active / inactive
        hyper = dict(beta=0.15, BND=BND, alpha=1e-2, w_f=0.05, w_r=0.1, BND_pro=BND_pro)
        total, meters, coefs, dl, de = ops.loss_tail(logits, labels, nr, emb, table, st, **hyper)
        cr, cf = ops.ce_fwd(logits[:nr], labels[:nr]), ops.ce_fwd(logits[nr:], labels[nr:])
        kf = ops.proto_kl_fwd(emb[nr:], labels[nr:], table)[0] if proto else None
        kr = ops.proto_kl_fwd(emb[:nr], labels[:nr], table)[0] if proto else None
        t0, m0, c0 = ops.loss_combine(cr[0], cf[0], kf, kr, st, cr[1], cf[1], float(nr), float(N - nr), **hyper)
        # (the scalar tail is compiled twice: an fma contraction may differ in the last bit of the total; coefficients and gradients are exact)
        assert torch.allclose(total, t0, rtol=3e-7, atol=0) and torch.allclose(meters, m0, rtol=3e-7, atol=0) and torch.equal(coefs, c0)
        dl0 = torch.empty_like(logits)
        ops.ce_bwd(logits[:nr], labels[:nr], c0[0:1].contiguous(), 1.0, dlogits=dl0[:nr], accumulate=False)
        ops.ce_bwd(logits[nr:], labels[nr:], c0[1:2].contiguous(), 1.0, dlogits=dl0[nr:], accumulate=False)
        assert torch.equal(dl, dl0)
        if proto:
            de0 = torch.empty_like(emb)
            ops.proto_kl_bwd(emb[:nr], labels[:nr], table, c0[3:4].contiguous(), 1.0, demb=de0[:nr], accumulate=False)
            ops.proto_kl_bwd(emb[nr:], labels[nr:], table, c0[2:3].contiguous(), 1.0, demb=de0[nr:], accumulate=False)
            assert torch.equal(de, de0)
        else:
            assert de is None


@pytest.mark.parametrize("r", [8, 16])
def test_fused_ffn1_rank_update_equals_the_k_segment_form(ops, r):
    """Fused FFN1 on the 8-phase kernel: a 64-column LoRA K segment is applied as one rank-32 k-step behind the K loop (gemm.hip, GSL_TUPD).
    The same call with the segment zero-padded to 128 columns takes the K-tile path: both outputs (h and the 8-bit GELU' code) must be
    bit-identical, and close to torch fp32. (The update form is compiled in with -DGSL_TUPD=1 only: measured slower, profiles/r03_notes.md;
    with the default build both calls take the K-tile path and the test pins that zero padding changes nothing.)"""
    from gslora_hip import _lib as L
    dt = torch.bfloat16
    M, N, K = 16500, 2048, 512
    c = lambda t: t.cuda().to(dt)
    A, W, bias = c(rnd(M, K, seed=1)), c(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3).cuda()
    u = torch.zeros(M, 128); u[:, :r] = rnd(M, r, seed=4)
    B = torch.zeros(N, 128); B[:, :r] = rnd(N, r, seed=5, scale=0.3)
    u, B = c(u), c(B)
    outs = []
    for width in (64, 128):
        h = torch.empty(M, N, device="cuda", dtype=dt); gp = torch.empty(M, N, device="cuda", dtype=torch.uint8)
        ops.gemm_nt(A, W, h, epilogue=L.EPI_BIAS_GELU_G8, A2=u[:, :width].contiguous(), W2=B[:, :width].contiguous(), bias=bias, out2=gp,
                    p_drop=0.1, seed=11, site=5)
        outs.append((h, gp))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    keep = ops.dropout_mask(M * N, 0.1, 11, 5, "cuda").view(M, N).float() / 0.9
    ref = F.gelu(A.float() @ W.float().t() + u.float() @ B.float().t() + bias) * keep
    assert relerr(outs[0][0].float(), ref) < 1.5e-2


@pytest.mark.parametrize("M,N,K1,K2,epi", [(333, 200, 192, 64, "store"), (130, 384, 128, 0, "res"), (257, 512, 64, 64, "gelu"), (700, 768, 768, 0, "mul"),
                                           (65, 136, 2048, 0, "store"), (1000, 2048, 512, 64, "gelu")])
def test_f32_gemm_mfma_equals_forced_valu_kernel_bitwise(ops, M, N, K1, K2, epi, monkeypatch):
    """the development build's GSL_F32_VALU=1 forces the VALU f32 GEMM at every shape (the parity-debugging switch). The
    matrix-core kernel of the product library must equal it bit for bit — partial tiles in M and N, the K1 | K2 split, every epilogue of
    the parity mode."""
    from gslora_hip import _lib as L
    A1, W1 = rnd(M, K1, seed=11).cuda(), rnd(N, K1, seed=12, scale=K1 ** -0.5).cuda()
    A2 = W2 = None
    if K2:
        A2, W2 = rnd(M, K2, seed=13).cuda(), rnd(N, K2, seed=14, scale=0.1).cuda()
    bias, res, aux = rnd(N, seed=15).cuda(), rnd(M, N, seed=16).cuda(), rnd(M, N, seed=17).cuda()

    def call():
        out, out2 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
        if epi == "store":
            ops.gemm_nt(A1, W1, out, A2=A2, W2=W2, alpha=0.5)
        elif epi == "gelu":
            ops.gemm_nt(A1, W1, out, epilogue=L.EPI_BIAS_GELU, A2=A2, W2=W2, bias=bias, out2=out2, p_drop=0.1, seed=5, site=3)
        elif epi == "mul":
            ops.gemm_nt(A1, W1, out, epilogue=L.EPI_MUL, A2=A2, W2=W2, aux=aux)
        else:
            ops.gemm_nt(A1, W1, out, epilogue=L.EPI_BIAS_RES_F32, A2=A2, W2=W2, bias=bias, res=res, p_drop=0.1, seed=5, site=2)
        return out, out2
    mf, mf2 = call()
    monkeypatch.setenv("GSL_F32_VALU", "1")      # a knob of the development build only
    with L.use_dev():
        va, va2 = call()
    assert torch.equal(mf, va) and torch.equal(mf2, va2)


@pytest.mark.parametrize("M,N,K1,K2,epi", [(300, 512, 192, 0, "store"), (1576, 2048, 512, 64, "gelu"), (1000, 512, 2048, 64, "res"), (197 * 3, 1536, 512, 0, "store")])
def test_f32_gemm_mfma_equals_valu_bitwise(ops, M, N, K1, K2, epi):
    """Parity mode, round 4: gemm_f32_mfma_kernel (v_mfma_f32_16x16x4_f32, N >= 128) against gemm_f32_kernel (VALU fmaf, taken for N = 64):
    an f32 MFMA is a k-ordered fmaf chain, so the two kernels must agree BIT FOR BIT. The same product is computed once as one N-wide GEMM
    (matrix cores) and once as N / 64 column slices (VALU kernel); also against torch in f64 to f32 round-off."""
    from gslora_hip import _lib as L
    A1, W1 = rnd(M, K1, seed=1).cuda(), rnd(N, K1, seed=2, scale=K1 ** -0.5).cuda()
    A2 = W2 = None
    if K2:
        A2, W2 = rnd(M, K2, seed=3).cuda(), rnd(N, K2, seed=4, scale=0.1).cuda()
    bias = rnd(N, seed=5).cuda()
    res = rnd(M, N, seed=6).cuda()

    def call(w1, w2, b, r, out, out2):
        if epi == "store":
            ops.gemm_nt(A1, w1, out, A2=A2, W2=w2)
        elif epi == "gelu":
            ops.gemm_nt(A1, w1, out, epilogue=L.EPI_BIAS_GELU, A2=A2, W2=w2, bias=b, out2=out2)
        else:
            ops.gemm_nt(A1, w1, out, epilogue=L.EPI_BIAS_RES_F32, A2=A2, W2=w2, bias=b, res=r)
    full, full2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    call(W1, W2, bias, res, full, full2)
    for c0 in range(0, N, 64):
        sl = slice(c0, c0 + 64)
        part, part2 = torch.empty(M, 64, device="cuda"), torch.empty(M, 64, device="cuda")
        call(W1[sl].contiguous(), None if W2 is None else W2[sl].contiguous(), bias[sl].contiguous(), res[:, sl].contiguous(), part, part2)
        assert torch.equal(part, full[:, sl]), c0
        if epi == "gelu":
            assert torch.equal(part2, full2[:, sl]), c0
    acc = A1.double() @ W1.double().t() + (A2.double() @ W2.double().t() if K2 else 0)
    want = acc if epi == "store" else (torch.nn.functional.gelu(acc + bias.double()) if epi == "gelu" else acc + bias.double() + res.double())
    assert (full.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_head_with_an_fp16_residual_stream(ops):
    """gsl_head_fwd / gsl_head_bwd reading the forward residual stream in fp16 (x_dtype = GSL_F16) == the same kernels on the widened f32
    copy of that tensor, bit for bit (the stream is converted on load; nothing else changes)."""
    B, T, D, C = 6, 9, 128, 10
    dt = torch.bfloat16
    x = rnd(B * T, D, seed=1, scale=1.5).cuda().to(torch.float16)
    g, b = (1 + 0.1 * rnd(D, seed=2)).cuda(), (0.1 * rnd(D, seed=3)).cuda()
    Wn = ops.cosface_prep(rnd(C, D, seed=4).cuda())
    label = (torch.arange(B) % C).cuda()
    l16, e16, m16, r16 = ops.head_fwd(x, B, T, D, g, b, 1e-5, Wn, label, 64.0, 0.35)
    l32, e32, m32, r32 = ops.head_fwd(x.float(), B, T, D, g, b, 1e-5, Wn, label, 64.0, 0.35)
    assert torch.equal(l16, l32) and torch.equal(e16, e32) and torch.equal(m16, m32) and torch.equal(r16, r32)
    dl, de = rnd(B, C, seed=5).cuda(), rnd(B, D, seed=6).cuda()
    for sdt in (torch.bfloat16, torch.float32):
        kw = dict(p_drop=0.3, seed=21, site=6, stream_dtype=sdt, compact=True)
        a = ops.head_bwd(dl, de, x, B, T, D, g, m16, r16, e16, Wn, 64.0, dt, **kw)
        c = ops.head_bwd(dl, de, x.float(), B, T, D, g, m16, r16, e16, Wn, 64.0, dt, **kw)
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


def test_out_of_range_labels_turn_the_loss_nan_instead_of_reading_out_of_bounds(ops):
    """a label outside [0, C) (the reference's CrossEntropyLoss raises) must not index the logits row out of bounds: the CE
    kernels — separate (gsl_ce_fwd / gsl_ce_bwd) and fused (gsl_loss_tail) — produce NaN for that row's loss and gradient, which the
    engines' deferred meter read (MeterQueue.flush) turns into a FloatingPointError."""
    B, C, D = 8, 10, 64
    logits = rnd(B, C, seed=1).cuda()
    good = (torch.arange(B) % C).cuda()
    for bad_value in (C, -1, 1 << 20):
        bad = good.clone(); bad[3] = bad_value
        assert torch.isfinite(ops.ce_fwd(logits, good)).all()
        out = ops.ce_fwd(logits, bad)
        assert torch.isnan(out[0]) and torch.isfinite(out[1])                       # summed loss NaN, hit count intact
        g = ops.ce_bwd(logits, bad, torch.ones(1, device="cuda"), 1.0)
        assert torch.isnan(g[3]).all() and torch.isfinite(g[[0, 1, 2, 4, 5, 6, 7]]).all()
        emb, proto = rnd(B, D, seed=2).cuda(), rnd(C, D, seed=3).cuda()
        total, meters, coefs, dl, de = ops.loss_tail(logits, bad, 4, emb, proto, torch.ones(1, device="cuda"), 0.15, 105.0, 1e-2, 0.05, 0.1, 2.0)
        assert torch.isnan(total) and torch.isnan(dl[3]).all() and torch.isfinite(dl[0]).all()


def test_fp16_operand_mode_stores_saturate_and_keep_nan(ops):
    """dtype GSL_F16 (round 5): every 16-bit store of the fp16 operand mode saturates finite overflow at +-65504 in hardware
    (MODE.FP16_OVFL set at kernel entry) — the GEMM STORE epilogue on the 8-phase, ring and small-tile kernels, the LayerNorm forward
    output, the cast helper — and the VALU paths keep NaN / Inf. (NaN / Inf in an MFMA OPERAND is a different matter and not asserted: the
    gfx950 matrix core itself drops a NaN operand's 8-element k-group and clamps Inf to the largest finite value, in bf16 and fp16 alike —
    tools/probes/nan_probe.py; NaN reaches the 16-bit tensors through the f32 epilogue arithmetic: see the residual-stream test above.)"""
    dt = torch.float16
    for M, N, K in ((33490, 512, 512), (2100, 128, 128), (300, 256, 64)):
        A = torch.full((M, K), 200.0, device="cuda", dtype=dt)
        W = torch.full((N, K), 200.0, device="cuda", dtype=dt)      # every output = K * 4e4 >> 65504
        W[1] = -200.0
        out = torch.empty(M, N, device="cuda", dtype=dt)
        ops.gemm_nt(A, W, out)
        assert torch.isfinite(out.float()).all() and float(out[:, 0].float().min()) == 65504.0 and float(out[:, 1].float().max()) == -65504.0, (M, N, K)
    x = rnd(64, 512, seed=1).cuda()
    y, _, _ = ops.layernorm_fwd(x, 512, 64, 512, torch.full((512,), 1e6).cuda(), torch.zeros(512).cuda(), 1e-5, dt)
    assert torch.isfinite(y.float()).all() and float(y.float().abs().max()) == 65504.0
    x[3, 7] = float("nan")
    y, _, _ = ops.layernorm_fwd(x, 512, 64, 512, torch.ones(512).cuda(), torch.zeros(512).cuda(), 1e-5, dt)
    assert torch.isnan(y[3]).all() and torch.isfinite(y[:3].float()).all()
    c = ops.cast(torch.tensor([1e6, -1e6, float("inf"), float("nan"), 1.0]).cuda(), dt)
    assert c[0] == 65504.0 and c[1] == -65504.0 and c[2] == float("inf") and torch.isnan(c[3]) and c[4] == 1.0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,T,bias", [(33490, 512, 512, 0, False), (33490, 1536, 512, 197, False), (33490, 1536, 768, 197, True), (33490, 512, 1536, 0, False),
                                          (4096, 512, 128, 0, True), (65536, 1024, 512, 0, False)])
def test_gemm_w4_store_kernel(ops, dev_lib, monkeypatch, M, N, K, T, bias, dt):
    """The 4-wave 32x32x16 kernel of the plain-store GEMMs (csrc/gemm_w4.inc: a measured alternative in the dev build, GSL_W4=1; N % 256 == 0, no K
    segment, K >= 128) against torch fp32 on the operands the kernel sees, and against the product's 8-wave 8-phase kernel: same products,
    another summation order (k in steps of 16 instead of 32) — equal within f32 accumulation noise, i.e. almost always the same 16-bit value.
    Ragged last M tile, row-major and head-major (STORE_QKV_HM) outputs, optional bias, strided operand views."""
    from gslora_hip import _lib as L
    A = rnd(M, K, seed=1).cuda().to(dt)
    Wfull = rnd(N, K + 64, seed=2, scale=K ** -0.5).cuda().to(dt)
    W = Wfull[:, 32:32 + K]                                   # a column block: ldw != K
    b = rnd(N, seed=3).cuda() if bias else None
    epi = L.EPI_STORE_QKV_HM if T else L.EPI_STORE
    old = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(A, W, old, epilogue=epi, T=T, bias=b)         # product library: the 8-phase kernel
    dev_lib(L)
    monkeypatch.setenv("GSL_W4", "1")
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(A, W, out, epilogue=epi, T=T, bias=b)
    ref = A.float() @ W.float().t() + (b if bias else 0)
    if T:
        Bn, H = M // T, N // 192
        ref = ref.view(Bn, T, 3, H, 64).permute(0, 3, 2, 1, 4).reshape(M, N)
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert ((out.float() - ref).abs() <= eps * ref.abs() + 1e-3).all()
    assert (old != out).float().mean() < 0.02 and ((old.float() - out.float()).abs() <= 2 * eps * ref.abs() + 1e-3).all()


# ---------------------------------------------------------------------------------------------------------------- the overlap GEMM (round 6)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K1,K2,T,mode", [(33490, 512, 512, 0, 0, "plain"), (33490, 512, 1536, 0, 0, "bias"), (33490, 1536, 512, 0, 197, "hm"),
                                               (33490, 1536, 512, 0, 197, "hm_ln"), (33490, 512, 512, 0, 0, "ln"), (33490, 512, 128, 64, 0, "alpha"),
                                               (201728, 512, 512, 0, 0, "plain"), (65536, 1024, 64, 64, 0, "bias")])
def test_gemm_o4_overlap_kernel_store(ops, dev_lib, monkeypatch, M, N, K1, K2, T, mode, dt):
    """The overlap GEMM (csrc/gemm_o4.inc: 4-wave workgroups on 256 x 128 tiles of v_mfma_f32_32x32x16, two resident per CU — a measured alternative
    in the dev build, GSL_O4=1, for the 8-phase class) against torch on the operands the kernel sees, and against the product's 8-phase kernel:
    same products, another summation order — equal within f32 accumulation noise, i.e. almost always the same 16-bit value. Ragged last M tile (33490 = 130 x 256 + 210), K segments, bias, alpha, the consumer-side LayerNorm, the head-major
    QKV copy-out, strided operand views. Reference: F.linear of vit_pytorch_face/vit_face.py:349-356."""
    from gslora_hip import _lib as L
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(M, K1, generator=g) + (0.5 * torch.randn(M, 1, generator=g) if "ln" in mode else 0)).cuda().to(dt)
    Wfull = rnd(N, K1 + 64, seed=2, scale=K1 ** -0.5).cuda().to(dt)
    W = Wfull[:, 32:32 + K1]                                   # a column block: ldw != K
    A2 = W2 = None
    if K2:
        a2 = rnd(M, K2, seed=3); a2[:, 8:] = 0
        A2, W2 = a2.cuda().to(dt), rnd(N, K2, seed=4, scale=0.1).cuda().to(dt)
    kw = dict(A2=A2, W2=W2)
    ref = A.double() @ W.double().t() + (A2.double() @ W2.double().t() if K2 else 0)
    if mode == "bias":
        kw["bias"] = rnd(N, seed=5).cuda(); ref = ref + kw["bias"].double()
    if mode == "alpha":
        kw["alpha"] = 0.125; ref = ref * 0.125
    epi = L.EPI_STORE_QKV_HM if T else L.EPI_STORE
    if "ln" in mode:
        mean, rstd = ops.layernorm_stats(A, K1, M, K1, torch.ones(K1).cuda(), torch.zeros(K1).cuda(), 1e-5, dt)
        c = W.float().sum(1).contiguous(); d = rnd(N, seed=6).cuda()
        kw.update(pos=mean, cls=rstd, aux=c, bias=d)
        epi = L.EPI_STORE_QKV_HM_LN if T else L.EPI_STORE_LN
        ref = rstd.double()[:, None] * (ref - mean.double()[:, None] * c.double()[None, :]) + d.double()[None, :]
    if T:
        M = (M // T) * T
        A, ref = A[:M], _to_head_major(ref[:M], M // T, T, N // 192)
        if "ln" in mode:
            kw["pos"], kw["cls"] = kw["pos"][:M].contiguous(), kw["cls"][:M].contiguous()
    old = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(A, W, old, epilogue=epi, T=T, **kw)            # product library: the 8-phase kernel
    dev_lib(L)
    monkeypatch.setenv("GSL_O4", "1")
    out = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
    ops.gemm_nt(A, W, out, epilogue=epi, T=T, **kw)            # dev build, GSL_O4=1: the overlap kernel
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert torch.isfinite(out.float()).all()
    assert ((out.double() - ref).abs() <= eps * ref.abs() + 2e-3).all(), float((out.double() - ref).abs().max())
    print(f"[o4 vs 8-phase {mode} {dt}] outputs that differ: {(old != out).float().mean().item():.2e}")
    assert (old != out).float().mean() < 0.03 and ((old.float() - out.float()).abs() <= 2 * eps * ref.abs().float() + 2e-3).all()


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K1,K2,p", [(33490, 2048, 512, 64, 0.1), (33490, 512, 512, 0, 0.0), (201728, 2048, 512, 64, 0.1), (40000, 1024, 64, 64, 0.25)])
def test_gemm_o4_overlap_kernel_fused_ffn1(ops, dev_lib, monkeypatch, M, N, K1, K2, p, dt):
    """BIAS_GELU_G8 on the overlap GEMM (dev build, GSL_O4=1; the fused FFN1 of the benchmark: bias + table GELU + 8-bit GELU' code + dropout, two
    outputs) against the product's 8-phase kernel: the SAME dropout mask (dropped elements exactly 0 / code 26 in both), h within one 16-bit
    ulp + the table step where the f32 sums differ in the last bits, codes equal or one step apart; and against GELU in f64 on the operands the
    kernel sees. Reference: vit_pytorch_face/vit_face.py:326-338."""
    from gslora_hip import _lib as L
    c = lambda t: None if t is None else t.cuda().to(dt)
    A1, W1 = c(rnd(M, K1, seed=1)), c(rnd(N, K1, seed=2, scale=K1 ** -0.5))
    A2 = W2 = None
    if K2:
        a2 = rnd(M, K2, seed=3); a2[:, 8:] = 0
        A2, W2 = c(a2), c(rnd(N, K2, seed=4, scale=0.1))
    bias = rnd(N, seed=5).cuda()
    h0 = torch.empty(M, N, device="cuda", dtype=dt); q0 = torch.full((M, N), 255, device="cuda", dtype=torch.uint8)
    ops.gemm_nt(A1, W1, h0, epilogue=L.EPI_BIAS_GELU_G8, A2=A2, W2=W2, bias=bias, out2=q0, p_drop=p, seed=7, site=5)      # product: the 8-phase kernel
    dev_lib(L)
    monkeypatch.setenv("GSL_O4", "1")
    h1 = torch.full((M, N), float("nan"), device="cuda", dtype=dt); q1 = torch.full((M, N), 255, device="cuda", dtype=torch.uint8)
    ops.gemm_nt(A1, W1, h1, epilogue=L.EPI_BIAS_GELU_G8, A2=A2, W2=W2, bias=bias, out2=q1, p_drop=p, seed=7, site=5)
    keep = ops.dropout_mask(M * N, p, 7, 5, "cuda").reshape(M, N).bool() if p > 0 else torch.ones(M, N, device="cuda", dtype=torch.bool)
    qr = _unslab(q1)
    assert torch.isfinite(h1.float()).all() and int(qr.max()) <= 252
    assert (qr[~keep] == 26).all() and (h1[~keep] == 0).all()
    # f64 reference in row chunks (the [M, N] f64 tensors of the largest case do not have to exist at once)
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    tab = 2.8e-4 / (1 - p)
    half = 0.0025 / (1 - p); gtab = 0.8 * 0.0011 / (1 - p)
    for r0 in range(0, M, 16384):
        sl = slice(r0, min(M, r0 + 16384))
        a = A1[sl].double() @ W1.double().t() + (A2[sl].double() @ W2.double().t() if K2 else 0) + bias.double()
        cdf = 0.5 * (1 + torch.erf(a * 0.5 ** 0.5)); pdf = torch.exp(-0.5 * a * a) * 0.3989422804014327
        k = keep[sl].double() / (1 - p)
        want_h, want_g = a * cdf * k, (cdf + a * pdf) * k
        assert ((h1[sl].double() - want_h).abs() - eps * want_h.abs()).max().item() <= tab + 1e-3
        dec = (qr[sl].double() - 26.0) * (0.005 / (1 - p))
        assert (dec - want_g).abs().max().item() <= (half + gtab) * 1.05 + 1e-3
    q0 = _unslab(q0)
    print(f"[o4 vs 8-phase fused FFN1 {dt}] h values that differ: {(h0 != h1).float().mean().item():.2e}, codes: {(q0 != qr).float().mean().item():.2e}")
    assert torch.equal(q0 == 26, qr == 26) or ((q0 == 26) != (qr == 26)).float().mean() < 1e-3      # (a kept element may also code to 26: GELU' = 0)
    assert ((q0.int() - qr.int()).abs() <= 1).all() and (q0 != qr).float().mean() < 0.02
    assert (h0 != h1).float().mean() < 0.05 and ((h0.float() - h1.float()).abs() <= 2 * eps * h0.float().abs() + 2 * tab).all()


# ---------------------------------------------------------------------------------------------------------------- fp16 overflow guard (round 6)
def test_fp16_overflow_guard_ln_bwd_reports_the_largest_gradient_and_adamw_skips_a_saturated_step(ops):
    """every fp16 store saturates in hardware, so a clipped gradient used to be silent. gsl_layernorm_bwd(gmax) raises a
    device float to the largest |dy| it read / |dx| it stored (every gradient of the chain passes a LayerNorm backward); gsl_adamw_flat(guard)
    leaves p / m / v untouched when the guard holds >= 65504 (a saturated store) or a non-finite value — GradScaler.step semantics, no host sync."""
    M, D = 300, 512
    dt = torch.float16
    x = rnd(M, D, seed=1).cuda().to(dt)
    gam = (1 + 0.1 * rnd(D, seed=2)).cuda()
    _, mean, rstd = ops.layernorm_fwd(x, D, M, D, gam, torch.zeros(D).cuda(), 1e-5, dt)
    dy = (rnd(M, D, seed=3) * 40).cuda().to(dt)
    dres = (rnd(M, D, seed=4) * 300).cuda().to(dt)
    gmax = torch.zeros(2, device="cuda")
    dx, _ = ops.layernorm_bwd(dy, x, D, gam, mean, rstd, dres, want_copy=False, gmax=gmax[:1])
    want = max(float(dy.float().abs().max()), float(dx.float().abs().max()))
    seen = float(gmax[0])
    assert gmax[1] == 0 and abs(seen - want) <= 2.0 ** -10 * want + 1e-6, (seen, want)      # (|dx| is taken on the f32 value, before its one rounding)
    small = (rnd(M, D, seed=5)).cuda().to(dt)
    ops.layernorm_bwd(small, x, D, gam, mean, rstd, small, want_copy=False, gmax=gmax[:1])      # a maximum, not the last value
    assert float(gmax[0]) == seen
    dy2 = dy.clone(); dy2[17, 5] = 65504.0                                                   # what a saturated GEMM store upstream leaves behind
    ops.layernorm_bwd(dy2, x, D, gam, mean, rstd, dres, want_copy=False, gmax=gmax[:1])
    assert float(gmax[0]) >= 65504.0
    # the optimizer under the guard
    n = 4096
    p = rnd(n, seed=6).cuda(); g = rnd(n, seed=7).cuda(); m = torch.zeros(n).cuda(); v = torch.zeros(n).cuda()
    p0 = p.clone()
    ops.adamw_flat(p, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.05, 1, guard=gmax[:1])              # saturated: skipped
    assert torch.equal(p, p0) and not m.any() and not v.any()
    for bad in (float("inf"), float("nan")):
        gmax[0] = bad
        ops.adamw_flat(p, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.05, 1, guard=gmax[:1])
        assert torch.equal(p, p0)
    gmax[0] = 3000.0
    ops.adamw_flat(p, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.05, 1, guard=gmax[:1])              # clean: the plain update
    q = p0.clone(); mq = torch.zeros(n).cuda(); vq = torch.zeros(n).cuda()
    ops.adamw_flat(q, g, mq, vq, 1e-2, 0.9, 0.999, 1e-8, 0.05, 1)
    assert torch.equal(p, q) and torch.equal(m, mq) and torch.equal(v, vq) and not torch.equal(p, p0)
    step_dev = torch.ones(1, device="cuda", dtype=torch.int64); lr_dev = torch.full((1,), 1e-2, device="cuda")
    gmax[0] = 65504.0
    p1 = p.clone()
    ops.adamw_flat_dev(p, g, m, v, lr_dev, 0.9, 0.999, 1e-8, 0.05, step_dev, guard=gmax[:1])   # the HIP-graph form
    assert torch.equal(p, p1)


def test_fp16_loss_scale_backs_off_on_the_device_after_a_saturated_backward(ops):
    """gsl_head_bwd keeps {S, 1/S, seen maximum, exponent E} in a persistent device buffer: S * max|head gradient| lands in [2^(E-1), 2^E); a
    backward whose LayerNorm backwards saw >= 65504 lowers E by 2 for the next one (floor 4), a quiet one (< 2^9) raises it by 1 up to the
    target (default 11, or target_exp). No host value enters: the same sequence under HIP-graph replay."""
    B, T, D, C = 6, 9, 128, 10
    dt = torch.float16
    x = rnd(B * T, D, seed=1, scale=1.5).cuda().to(dt)
    g, b = (1 + 0.1 * rnd(D, seed=2)).cuda(), (0.1 * rnd(D, seed=3)).cuda()
    Wn = ops.cosface_prep(rnd(C, D, seed=4).cuda())
    label = (torch.arange(B) % C).cuda()
    _, emb, mean, rstd = ops.head_fwd(x, B, T, D, g, b, 1e-5, Wn, label, 64.0, 0.35)
    dl, de = (rnd(B, C, seed=5) * 1e-3).cuda(), (rnd(B, D, seed=6) * 1e-3).cuda()
    ref, _ = ops.head_bwd(dl, de, x.float(), B, T, D, g, mean, rstd, emb, Wn, 64.0, torch.float32, compact=True)      # unscaled f32 gradients
    amax = float(ref.abs().max())
    gs = torch.zeros(4, device="cuda")
    call = lambda **kw: ops.head_bwd(dl, de, x, B, T, D, g, mean, rstd, emb, Wn, 64.0, dt, stream_dtype=dt, compact=True, gscale=gs, **kw)

    def check(E):
        S, inv, seen, e = gs.tolist()
        assert e == E and seen == 0.0 and S * inv == 1.0 and 2.0 ** (E - 1) <= S * amax < 2.0 ** E, (gs.tolist(), amax, E)
    dx, _ = call()
    check(11)                                                     # the zeroed buffer starts at the default
    assert (dx.float() - ref * gs[0]).abs().max() <= 2.0 ** -10 * float((ref * gs[0]).abs().max())
    gs[2] = 65504.0; call(); check(9)                             # the previous backward saturated: two binades down
    gs[2] = float("inf"); call(); check(7)
    gs[2] = 3000.0; call(); check(7)                              # in range: stays
    gs[2] = 100.0; call(); check(8)                               # quiet: one binade up per step ...
    for E in (9, 10, 11, 11):
        gs[2] = 100.0; call(); check(E)                           # ... up to the target, not beyond
    gs.zero_(); call(target_exp=14); check(14)                    # a configured target (GSLORA_GRAD_TARGET_EXP)
    for _ in range(8):
        gs[2] = 7e4; call(target_exp=14)
    check(4)                                                      # floor
    with pytest.raises(RuntimeError):
        call(target_exp=3)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_in_kernel_lora_gemms_are_run_to_run_identical(ops, dt):
    """The kernels that hand t = s A P^T over between waves through LDS (8-phase LoRA tail, its gradient-fused form, the 64 x 64 kernel's tail — the
    tail split sends 5 120 rows of every in-kernel-LoRA residual GEMM of the step through the latter), repeated: every repetition must reproduce the
    first one bit for bit (round 6: a raw barrier behind the t stores let one wave read a row early, about once in ten process runs; the ISA lint in
    tests/test_host_logic.py is the deterministic guard, this is the empirical one)."""
    from gslora_hip import _lib as L
    M, N, K, r = 201728, 512, 512, 8
    g = torch.Generator(device="cuda").manual_seed(9)
    A = torch.randn(M, K, device="cuda", generator=g).to(dt)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.float16)
    bias = torch.randn(N, device="cuda", generator=g)
    P = torch.zeros(16, K, device="cuda"); P[:r] = torch.randn(r, K, device="cuda", generator=g) * K ** -0.5
    Q = torch.zeros(N, 32, device="cuda"); Q[:, :r] = torch.randn(N, r, device="cuda", generator=g) * 0.3
    P, Q = P.to(dt), Q.to(dt)
    small = slice(0, 1576)                      # the 64 x 64 kernel on its own (few-shot row count)
    first = None
    for _ in range(12):
        o1 = torch.empty(M, N, device="cuda", dtype=dt); t1 = torch.empty(M, 64, device="cuda", dtype=dt)
        o2 = torch.empty(M, N, device="cuda", dtype=torch.float16); t2 = torch.empty(M, 64, device="cuda", dtype=dt)
        o3 = torch.empty(1576, N, device="cuda", dtype=dt); t3 = torch.empty(1576, 64, device="cuda", dtype=dt)
        ops.gemm_nt_lora(A, W, P, Q, 1.0 / r, t1, o1)
        ops.gemm_nt_lora(A, W, P, Q, 1.0 / r, t2, o2, epilogue=L.EPI_BIAS_RES_F16, bias=bias, res=res, p_drop=0.1, seed=77, site=6)
        ops.gemm_nt_lora(A[small], W, P, Q, 1.0 / r, t3, o3)
        cur = (o1, t1, o2, t2, o3, t3)
        if first is None:
            first = cur
            assert all(torch.isfinite(x.float()).all() for x in cur)
        else:
            for a, b in zip(cur, first):
                assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------- tail split (round 6)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,lora", [(512, False), (1536, False), (2048, True), (512, True)])
def test_gemm_tail_split_bit_identical_to_the_single_launch(ops, dev_lib, monkeypatch, dt, K, lora):
    """Round 6: a GEMM of the 8-phase class whose last round of 256 x 256 tiles would be at most 30 % full (M = 201 728, N = 512: 1 576 tiles on
    256 CUs = 6.16 rounds) is launched as whole rounds + a tail of 64 x 64 tiles for the remaining row panels (csrc/gemm.hip: tail_split_rows). The
    tail rows must be BIT-IDENTICAL to what the single 8-phase launch writes (same MFMA instruction, k order and epilogue arithmetic) — the fused
    two-batch forward has to equal two forwards bit for bit —, including the dropout mask of the residual epilogue (the tail's counters continue
    at its first row) and the in-kernel-LoRA form with its t = s A P^T output. Single launch = dev build with GSL_TAIL_SPLIT=0."""
    from gslora_hip import _lib as L
    M, N, r = 201728, 512, 8
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(M, K, device="cuda", generator=g).to(dt)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.float16)
    bias = torch.randn(N, device="cuda", generator=g)
    P = torch.zeros(16, K, device="cuda"); P[:r] = torch.randn(r, K, device="cuda", generator=g) * K ** -0.5
    Q = torch.zeros(N, 32, device="cuda"); Q[:, :r] = torch.randn(N, r, device="cuda", generator=g) * 0.3
    P, Q = P.to(dt), Q.to(dt)

    def run():
        o1 = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
        o2 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
        t1 = torch.full((M, 64), float("nan"), device="cuda", dtype=dt) if lora else None
        t2 = torch.full((M, 64), float("nan"), device="cuda", dtype=dt) if lora else None
        if lora:
            ops.gemm_nt_lora(A, W, P, Q, 1.0 / r, t1, o1)
            ops.gemm_nt_lora(A, W, P, Q, 1.0 / r, t2, o2, epilogue=L.EPI_BIAS_RES_F16, bias=bias, res=res, p_drop=0.1, seed=77, site=6)
        else:
            ops.gemm_nt(A, W, o1)
            ops.gemm_nt(A, W, o2, epilogue=L.EPI_BIAS_RES_F16, bias=bias, res=res, p_drop=0.1, seed=77, site=6)
        return o1, o2, t1, t2
    split = run()                                   # product library: 196 608 rows on the 8-phase kernel + 5 120 rows on the 64 x 64 ring
    dev_lib(L)
    monkeypatch.setenv("GSL_TAIL_SPLIT", "0")
    single = run()
    for a, b in zip(split, single):
        if a is not None:
            assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    # and the split does happen in the product build: the dev build with the split ON agrees with it as well (same code path)
    monkeypatch.setenv("GSL_TAIL_SPLIT", "1")
    again = run()
    assert torch.equal(again[1], split[1])
    # the mask really covers the tail with its own counters: keep rate of the tail rows
    ref_nodrop = torch.empty(M, N, device="cuda", dtype=torch.float16)
    if not lora:
        ops.gemm_nt(A, W, ref_nodrop, epilogue=L.EPI_BIAS_RES_F16, bias=bias, res=res)
        keep = ops.dropout_mask(M * N, 0.1, 77, 6, "cuda").reshape(M, N)
        tail = slice(M - 5120, M)
        dropped = (split[1][tail] == res[tail])     # a dropped element is exactly the residual
        assert torch.equal(dropped | (keep[tail] != 0), torch.ones_like(dropped)) and abs(float((keep[tail] == 0).float().mean()) - 0.1) < 5e-3
        assert (dropped & (keep[tail] != 0)).float().mean() < 1e-3      # (a kept element equals the residual only when the sum rounds away)
