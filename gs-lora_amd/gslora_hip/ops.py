"""Thin torch-tensor wrappers over the C ABI (include/gslora_hip.h). PyTorch supplies device
memory and the stream; every FLOP of the step runs in libgslora_hip.so. No fallbacks."""
import ctypes

import torch

from . import _lib as L

DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}      # (fp16: MFMA operand format of the "fp16" mode; also the forward residual stream of both 16-bit modes)


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need(*ts, rows_ok=False):
    """rows_ok: 2-D row slices / column blocks (unit inner stride) are accepted — the entry point takes the leading dimension."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("gslora_hip: tensors must live on a ROCm GPU (the HIP path has no CPU fallback)")
        if not (t.is_contiguous() or (rows_ok and t.dim() == 2 and t.stride(1) == 1)):
            raise RuntimeError("gslora_hip: tensors must be contiguous")


def code(dtype):
    try:
        return DT[dtype]
    except KeyError:
        raise RuntimeError(f"gslora_hip: unsupported compute dtype {dtype}")


def patchify(img, p, dtype):
    """img: one [B, C, H, W] batch or a list of batches of one image shape; the batches land in consecutive row ranges of the output."""
    parts = list(img) if isinstance(img, (tuple, list)) else [img]
    _need(*parts)
    _, Cc, H, W = parts[0].shape
    T = 1 + (H // p) * (W // p)
    out = torch.empty(sum(t.shape[0] for t in parts) * T, p * p * Cc, device=parts[0].device, dtype=dtype)
    row = 0
    for t in parts:
        L.check(L.load().gsl_patchify(_p(t), _p(out[row:]), t.shape[0], Cc, H, W, p, code(dtype), _stream()), "gsl_patchify")
        row += t.shape[0] * T
    return out


# optional per-kernel timing hook used by bench.py: {tag: [(start_event, end_event), ...]} recorded on the
# stream the kernel is launched on (torch's current stream == the hipStream_t handed to the C ABI)
PROFILE = None


def _profiled(tag, shape_of):
    """bench.py: when PROFILE holds `tag`, bracket the launch with HIP events on the launch stream and record (ev0, ev1, *shape_of(args)).
    (The GEMM wrappers do this per call-site tag; these are the memory-bound kernels of the step: LayerNorm and attention.)"""
    def deco(fn):
        def wrapped(*a, **kw):
            if PROFILE is None or tag not in PROFILE:
                return fn(*a, **kw)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            out = fn(*a, **kw)
            ev[1].record()
            PROFILE[tag].append((ev[0], ev[1]) + tuple(shape_of(*a, **kw)))
            return out
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco


def gemm_nt(A1, W1, out, *, epilogue=L.EPI_STORE, A2=None, W2=None, alpha=1.0, bias=None, res=None, aux=None, out2=None,
            pos=None, cls=None, T=0, p_drop=0.0, seed=0, site=0, tag=None):
    _need(A2, W2, bias, aux, out2, pos, cls)
    _need(A1, W1, out, res, rows_ok=True)      # lda1 / ldw1 / ldo travel with the call (res shares ldo with out)
    if res is not None and res.shape[0] > 1 and res.stride(0) != out.stride(0):
        raise RuntimeError("gemm_nt: res and out must share one row stride")
    M, K1 = A1.shape
    N = W1.shape[0]
    K2 = 0 if A2 is None else A2.shape[1]
    if PROFILE is not None and tag in PROFILE:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        gemm_nt(A1, W1, out, epilogue=epilogue, A2=A2, W2=W2, alpha=alpha, bias=bias, res=res, aux=aux, out2=out2, pos=pos,
                cls=cls, T=T, p_drop=p_drop, seed=seed, site=site, tag=None)
        ev[1].record()
        PROFILE[tag].append((ev[0], ev[1], M, N, K1, K2))
        return out
    L.check(L.load().gsl_gemm_nt(_p(A1), A1.stride(0), _p(W1), W1.stride(0), K1, _p(A2), 0 if A2 is None else A2.stride(0),
                                 _p(W2), 0 if W2 is None else W2.stride(0), K2, M, N, code(A1.dtype), epilogue, float(alpha),
                                 _p(bias), _p(res), _p(aux), _p(out), _p(out2), out.stride(0), _p(pos), _p(cls), int(T),
                                 float(p_drop), int(seed), int(site), _stream()), "gsl_gemm_nt")
    return out


def gemm_nt_lora(A, W, P, Q, lora_scale, tout, out, *, epilogue=L.EPI_STORE, bias=None, res=None, aux=None, out2=None, p_drop=0.0,
                 seed=0, site=0, tag=None):
    """out = epilogue(A W^T + t Q^T), t = lora_scale * A P^T computed inside the kernel and stored to tout [M,64] (bf16)."""
    _need(A, W, P, Q, tout, out, bias, res, aux, out2)
    M, K = A.shape
    N = W.shape[0]
    if PROFILE is not None and tag in PROFILE:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        gemm_nt_lora(A, W, P, Q, lora_scale, tout, out, epilogue=epilogue, bias=bias, res=res, aux=aux, out2=out2, p_drop=p_drop,
                     seed=seed, site=site)
        ev[1].record()
        PROFILE[tag].append((ev[0], ev[1], M, N, K, 0))
        return out
    L.check(L.load().gsl_gemm_nt_lora(_p(A), A.stride(0), _p(W), W.stride(0), K, _p(P), P.stride(0), _p(Q), Q.stride(0),
                                      float(lora_scale), _p(tout), 0 if tout is None else tout.stride(0), M, N, code(A.dtype), epilogue,
                                      _p(bias), _p(res), _p(aux), _p(out), _p(out2), out.stride(0), float(p_drop), int(seed), int(site),
                                      _stream()), "gsl_gemm_nt_lora")
    return out


def gemm_nt_lora_mulgrad(A, W, P, Q, lora_scale, tout, out, aux, U1, G1, g1s, Y2, G2, g2s, r, accumulate=True, tag=None, p_drop=0.0, gscale=None):
    """out = (A W^T + t Q^T) * aux with t = lora_scale * A P^T (as gemm_nt_lora, epilogue MUL) and, from the same tiles,
    G1[n*g1s[0] + j*g1s[1]] (+)= sum_m out[m,n] U1[m,j] and G2[n*g2s[0] + j*g2s[1]] (+)= sum_m Y2[m,n] t[m,j].
    A uint8 aux is the 8-bit GELU' code of EPI_BIAS_GELU_G8; p_drop is then the dropout rate of the forward that wrote it.
    gscale: device {S, 1/S} of a loss-scaled (fp16) backward — G1 / G2 receive the sums multiplied by 1/S."""
    _need(A, W, P, Q, tout, out, aux, U1, Y2, gscale)
    M, K = A.shape
    N = W.shape[0]
    if PROFILE is not None and tag in PROFILE:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        gemm_nt_lora_mulgrad(A, W, P, Q, lora_scale, tout, out, aux, U1, G1, g1s, Y2, G2, g2s, r, accumulate, p_drop=p_drop, gscale=gscale)
        ev[1].record()
        PROFILE[tag].append((ev[0], ev[1], M, N, K, 0))
        return out
    if not (out.stride(0) == aux.stride(0) == Y2.stride(0) and U1.stride(1) == 1):
        raise RuntimeError("gemm_nt_lora_mulgrad: out / aux / Y2 must share one row stride")
    lib = L.load()
    need = lib.gsl_gemm_mulgrad_ws_elems(M, N, r)
    key = (A.device.index, "mulgrad")
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _ws_retired.append(ws)
        ws = torch.empty(need, device=A.device, dtype=torch.float32)
        _ws_cache[key] = ws
    L.check(lib.gsl_gemm_nt_lora_mulgrad(_p(A), A.stride(0), _p(W), W.stride(0), K, _p(P), P.stride(0), _p(Q), Q.stride(0),
                                         float(lora_scale), _p(tout), 0 if tout is None else tout.stride(0), M, N, _p(aux), _p(out),
                                         out.stride(0), _p(U1), U1.stride(0), G1.data_ptr(), g1s[0], g1s[1], _p(Y2), G2.data_ptr(),
                                         g2s[0], g2s[1], r, 1 if accumulate else 0, _p(ws), 1 if aux.dtype == torch.uint8 else 0,
                                         float(p_drop), code(A.dtype), _p(gscale), _stream()), "gsl_gemm_nt_lora_mulgrad")
    return out


@_profiled("ln_fwd", lambda x, row_stride, M, D, *a, **k: (M, D, 0, 0))
def layernorm_fwd(x, row_stride, M, D, gamma, beta, eps, dtype):
    """x: the residual stream, f32 or (bf16 mode) bf16 — its dtype is handed to the kernel."""
    _need(x, gamma, beta)
    y = torch.empty(M, D, device=x.device, dtype=dtype)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    L.check(L.load().gsl_layernorm_fwd(_p(x), row_stride, _p(gamma), _p(beta), float(eps), _p(y), _p(mean), _p(rstd), M, D,
                                       code(dtype), code(x.dtype), _stream()), "gsl_layernorm_fwd")
    return y, mean, rstd


@_profiled("ln_stats", lambda x, row_stride, M, D, *a, **k: (M, D, 0, 0))
def layernorm_stats(x, row_stride, M, D, gamma, beta, eps, dtype):
    """(mean, rstd) of the rows of x — gsl_layernorm_fwd without its output: the row statistics a GEMM with a consumer-side LayerNorm
    (EPI_STORE_LN) finishes the normalisation with, and what the LayerNorm backward needs. One read of x, no write of LN(x)."""
    _need(x, gamma, beta)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    L.check(L.load().gsl_layernorm_fwd(_p(x), row_stride, _p(gamma), _p(beta), float(eps), None, _p(mean), _p(rstd), M, D,
                                       code(dtype), code(x.dtype), _stream()), "gsl_layernorm_fwd(stats)")
    return mean, rstd


def layernorm_fwd_lora(x, row_stride, M, D, gamma, beta, eps, P, alpha, pad=64):
    """bf16 mode: (LN(x), mean, rstd, u) with u = alpha * LN(x) P[:16]^T in a [M, 64] K-segment buffer (columns >= 16 zero): LayerNorm and
    the LoRA down-projection of the layer that consumes it in one pass over x (gsl_layernorm_fwd_lora)."""
    _need(x, gamma, beta, P)
    if x.dtype != torch.bfloat16 or P.dtype != torch.bfloat16 or P.shape[0] < 16 or P.shape[1] != D or pad != 64:
        raise ValueError("layernorm_fwd_lora: bf16 x, P [>= 16, D] bf16, 64-column u")
    y = torch.empty(M, D, device=x.device, dtype=torch.bfloat16)
    u = torch.empty(M, pad, device=x.device, dtype=torch.bfloat16)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    L.check(L.load().gsl_layernorm_fwd_lora(_p(x), row_stride, _p(gamma), _p(beta), float(eps), _p(y), _p(mean), _p(rstd), M, D,
                                            _p(P), P.stride(0), float(alpha), _p(u), _stream()), "gsl_layernorm_fwd_lora")
    return y, mean, rstd, u


@_profiled("ln_bwd", lambda dy, *a, **k: (dy.shape[0], dy.shape[1], 0, 0))
def layernorm_bwd(dy, x, row_stride, gamma, mean, rstd, dres, want_copy=True, p_drop=0.0, seed=0, site=0, dx=None,
                  io_row_stride=0, drop_row_stride=0, dres_cls_T=0, gmax=None):
    """dx = dres + LN'(dy). gmax: f32 device element (gscale[2:] of head_bwd) raised to the largest |dy| read / |dx| stored: the overflow guard. With dx given (and io_row_stride), the rows of an existing buffer are updated in place. The dtype of the
    residual-gradient stream (dres / dx: f32, or bf16 in bf16 mode) is taken from dres / dx, the dtype of the saved forward stream from x.
    dres_cls_T > 0: dres holds the cls rows only ([M / T, D]); the other rows of the incoming stream gradient are zero."""
    _need(dy, x, gamma, mean, rstd)
    M, D = dy.shape
    sdt = dx.dtype if dx is not None else (dres.dtype if dres is not None else torch.float32)
    own_dx = dx is None
    if dx is None:
        _need(dres)
        dx = torch.empty(M, D, device=dy.device, dtype=sdt)
    elif dres is not None and dres.dtype != dx.dtype:
        raise RuntimeError("layernorm_bwd: dres and dx must share one dtype")
    # the masked operand copy IS dx when no mask applies (dropout 0: ViT-B/16, eval-free training runs) and the stream already has the
    # operand dtype: one [M, D] write less per LayerNorm backward
    alias = bool(want_copy) and float(p_drop) == 0.0 and sdt == dy.dtype and own_dx and not io_row_stride
    dxb = torch.empty(M, D, device=dy.device, dtype=dy.dtype) if (want_copy and not alias) else None
    L.check(L.load().gsl_layernorm_bwd(_p(dy), _p(x), row_stride, _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx),
                                       int(io_row_stride), _p(dxb), M, D, code(dy.dtype), code(sdt), code(x.dtype), float(p_drop),
                                       int(seed), int(site), int(drop_row_stride), int(dres_cls_T), _p(gmax), _stream()), "gsl_layernorm_bwd")
    return dx, (dx if alias else dxb)


@_profiled("attn_fwd", lambda qkv, B, T, H, *a, **k: (B * T, H * 64, T, H))
def attention_fwd(qkv, B, T, H, scale, layout=0):
    """layout: 0 = qkv token-major [B*T, 3*H*64], 1 = head-major [B][H][3][T][64] (bf16; see gemm_nt(epilogue=EPI_STORE_QKV_HM))."""
    _need(qkv)
    o = torch.empty(B * T, H * 64, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32)
    L.check(L.load().gsl_attention_fwd(_p(qkv), _p(o), _p(lse), B, T, H, float(scale), code(qkv.dtype), int(layout), _stream()),
            "gsl_attention_fwd")
    return o, lse


@_profiled("attn_bwd", lambda qkv, o, d_o, lse, B, T, H, *a, **k: (B * T, H * 64, T, H))
def attention_bwd(qkv, o, d_o, lse, B, T, H, scale, layout=0):
    """dqkv is token-major [B*T, 3*H*64] whatever the layout of the qkv input."""
    _need(qkv, o, d_o, lse)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32)
    L.check(L.load().gsl_attention_bwd(_p(qkv), _p(o), _p(d_o), _p(lse), _p(dqkv), _p(delta), B, T, H, float(scale),
                                       code(qkv.dtype), int(layout), _stream()), "gsl_attention_bwd")
    return dqkv


def attention_fwd_cls(qkv, B, T, H, scale, layout=0, q_cls=None):
    """Attention output of the cls query alone (the last block under pool='cls'): o_cls [B, H*64], lse_cls [B, H].
    layout 2: `qkv` is kv [B*T, 2*H*64] (k | v, token-major) and q_cls [B, H*64] holds the queries."""
    _need(qkv, q_cls)
    o = torch.empty(B, H * 64, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(B, H, device=qkv.device, dtype=torch.float32)
    L.check(L.load().gsl_attention_fwd_cls(_p(qkv), _p(q_cls), _p(o), _p(lse), B, T, H, float(scale), code(qkv.dtype), int(layout),
                                           _stream()), "gsl_attention_fwd_cls")
    return o, lse


def attention_bwd_cls(qkv, o, d_o_cls, lse, B, T, H, scale, layout=0, q_cls=None):
    """o / lse: either the full forward tensors ([B*T, H*64] / [B, H, T]) or the compact ones of attention_fwd_cls ([B, H*64] / [B, H]).
    layout 0 / 1 -> dqkv [B*T, 3*H*64]; layout 2 (kv + q_cls) -> (dkv [B*T, 2*H*64], dq_cls [B, H*64])."""
    _need(qkv, o, d_o_cls, lse, q_cls)
    nw = 2 if layout == 2 else 3
    dqkv = torch.empty(B * T, nw * H * 64, device=qkv.device, dtype=qkv.dtype)
    dq = torch.empty(B, H * 64, device=qkv.device, dtype=qkv.dtype) if layout == 2 else None
    compact = o.shape[0] == B and lse.dim() == 2
    L.check(L.load().gsl_attention_bwd_cls(_p(qkv), _p(q_cls), _p(o), _p(d_o_cls), _p(lse), _p(dqkv), _p(dq), B, T, H, float(scale),
                                           code(qkv.dtype), int(layout), 1 if compact else 0, _stream()), "gsl_attention_bwd_cls")
    return (dqkv, dq) if layout == 2 else dqkv


_ws_cache = {}
_ws_retired = []


def lora_grad(Y, U, G, gsn, gsj, r, accumulate=True, gscale=None):
    """G[n*gsn + j*gsj] (+)= sum_m Y[m,n] U[m,j]; G is a view into the flat f32 gradient bucket. Y / U may be column blocks of wider
    row-major tensors (unit column stride). gscale: device {S, 1/S} of a loss-scaled (fp16) backward: the sum is multiplied by 1/S."""
    if not (Y.is_cuda and U.is_cuda and Y.stride(1) == 1 and U.stride(1) == 1):
        raise RuntimeError("lora_grad: operands must be CUDA tensors with unit column stride")
    M, N = Y.shape
    lib = L.load()
    need = lib.gsl_lora_grad_ws_elems(M, N, r)
    key = (Y.device.index,)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _ws_retired.append(ws)      # a captured HIP graph may still launch with the old workspace: never free it
        ws = torch.empty(max(need, 1 << 22), device=Y.device, dtype=torch.float32)
        _ws_cache[key] = ws
    L.check(lib.gsl_lora_grad(_p(Y), Y.stride(0), _p(U), U.stride(0), G.data_ptr(), gsn, gsj, M, N, r, code(Y.dtype),
                              1 if accumulate else 0, _p(ws), _p(gscale), _stream()), "gsl_lora_grad")


class _LgradDesc(ctypes.Structure):      # mirrors struct gsl_lgrad_desc (include/gslora_hip.h), 72 bytes
    _fields_ = [("Y", ctypes.c_void_p), ("ldy", ctypes.c_long), ("U", ctypes.c_void_p), ("ldu", ctypes.c_int), ("M", ctypes.c_int),
                ("N", ctypes.c_int), ("r", ctypes.c_int), ("accumulate", ctypes.c_int), ("pad_", ctypes.c_int), ("G", ctypes.c_void_p),
                ("gsn", ctypes.c_long), ("gsj", ctypes.c_long)]


def lora_grad_batchable(Y, U, r):
    """Can this reduction ride in gsl_lora_grad_batch (16-bit MFMA form: 256-column blocks, 16-byte rows)?"""
    return (Y.dtype in (torch.bfloat16, torch.float16) and U.dtype == Y.dtype and Y.is_cuda and Y.stride(1) == 1 and U.stride(1) == 1
            and Y.shape[1] % 256 == 0 and Y.stride(0) % 8 == 0 and U.stride(0) % 8 == 0 and U.stride(0) >= 16 and 1 <= r <= 16
            and Y.data_ptr() % 16 == 0 and U.data_ptr() % 16 == 0)


def lora_grad_batch(entries, gscale=None):
    """entries: [(Y, U, G, gsn, gsj, r, accumulate)] as for lora_grad — all of them in two launches (gsl_lora_grad_batch). The
    descriptors travel in the kernel arguments: nothing to keep alive on the host, HIP-graph capture friendly."""
    if not entries:
        return
    assert ctypes.sizeof(_LgradDesc) == 72
    arr = (_LgradDesc * len(entries))()
    for k, (Y, U, G, gsn, gsj, r, acc) in enumerate(entries):
        arr[k] = _LgradDesc(Y.data_ptr(), Y.stride(0), U.data_ptr(), U.stride(0), Y.shape[0], Y.shape[1], r, 1 if acc else 0, 0,
                            G.data_ptr(), gsn, gsj)
    lib = L.load()
    need = lib.gsl_lora_grad_batch_ws_elems(arr, len(entries))
    if need < 0:
        L.check(int(need), "gsl_lora_grad_batch_ws_elems")
    dev = entries[0][0].device
    key = (dev.index,)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _ws_retired.append(ws)      # a captured HIP graph may still launch with the old workspace: never free it
        ws = torch.empty(max(need, 1 << 22), device=dev, dtype=torch.float32)
        _ws_cache[key] = ws
    L.check(lib.gsl_lora_grad_batch(arr, len(entries), _p(ws), code(entries[0][0].dtype), _p(gscale), _stream()), "gsl_lora_grad_batch")


def loss_combine(ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r, hit_f, n_r, n_f, beta, BND, alpha, w_f, w_r, BND_pro):
    """-> (total [0-dim], meters [8], coefs [5]) — see gsl_loss_combine."""
    _need(ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r, hit_f)
    dev = ce_r_sum.device
    out = torch.empty(14, device=dev, dtype=torch.float32)
    L.check(L.load().gsl_loss_combine(_p(ce_r_sum), _p(ce_f_sum), _p(kl_f_sum), _p(kl_r_sum), _p(structure), _p(hit_r), _p(hit_f),
                                      float(n_r), float(n_f), float(beta), float(BND), float(alpha), float(w_f), float(w_r),
                                      float(BND_pro), out.data_ptr(), out.data_ptr() + 4, out.data_ptr() + 36, _stream()),
            "gsl_loss_combine")
    return out[0], out[1:9], out[9:14]


def loss_tail_max_rows():
    return int(L.load().gsl_loss_tail_max_rows())


def loss_tail(logits, labels, nr, emb, proto, structure, beta, BND, alpha, w_f, w_r, BND_pro):
    """The loss section of a single-process step in one launch (gsl_loss_tail): -> (total, meters [8], coefs [5], dlogits, demb or None)."""
    _need(logits, labels, emb, proto, structure)
    N, C = logits.shape
    out = torch.empty(14, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits)
    demb = None if emb is None else torch.empty_like(emb)
    L.check(L.load().gsl_loss_tail(_p(logits), _p(labels), N, int(nr), C, _p(emb), _p(proto), 0 if emb is None else emb.shape[1],
                                   0 if proto is None else proto.shape[0], _p(structure), float(beta), float(BND), float(alpha), float(w_f),
                                   float(w_r), float(BND_pro), _p(out), _p(dlogits), _p(demb), _stream()), "gsl_loss_tail")
    return out[0], out[1:9], out[9:14], dlogits, demb


def loss_combine_pack(pack8, structure, has_proto, beta, BND, alpha, w_f, w_r, BND_pro):
    """Data-parallel scalar tail from the all-reduced 8-float pack -> (total [0-dim], meters [8], coefs [5]) — see gsl_loss_combine_pack."""
    _need(pack8, structure)
    out = torch.empty(14, device=pack8.device, dtype=torch.float32)
    L.check(L.load().gsl_loss_combine_pack(_p(pack8), _p(structure), 1 if has_proto else 0, float(beta), float(BND), float(alpha),
                                           float(w_f), float(w_r), float(BND_pro), out.data_ptr(), out.data_ptr() + 4,
                                           out.data_ptr() + 36, _stream()), "gsl_loss_combine_pack")
    return out[0], out[1:9], out[9:14]


def cosface_prep(W):
    _need(W)
    Wn = torch.empty_like(W)
    L.check(L.load().gsl_cosface_prep(_p(W), _p(Wn), W.shape[0], W.shape[1], _stream()), "gsl_cosface_prep")
    return Wn


def head_fwd(x, B, T, D, gamma, beta, eps, Wn, label, cos_s, cos_m, head_bias=None, linear=False, pool_mean=False):
    _need(x, gamma, beta, Wn, label, head_bias)
    dev = x.device
    emb = torch.empty(B, D, device=dev, dtype=torch.float32)
    mean = torch.empty(B, device=dev, dtype=torch.float32)
    rstd = torch.empty(B, device=dev, dtype=torch.float32)
    C = Wn.shape[0] if Wn is not None else 0
    logits = torch.empty(B, C, device=dev, dtype=torch.float32) if (label is not None or linear) else None
    L.check(L.load().gsl_head_fwd(_p(x), code(x.dtype), T, _p(gamma), _p(beta), float(eps), _p(Wn), _p(label), _p(emb), _p(mean), _p(rstd),
                                  _p(logits), B, D, C, float(cos_s), float(cos_m), _p(head_bias), 1 if linear else 0,
                                  1 if pool_mean else 0, _stream()), "gsl_head_fwd")
    return logits, emb, mean, rstd


def head_bwd(dlogits, demb, x, B, T, D, gamma, mean, rstd, emb, Wn, cos_s, dtype, p_drop=0.0, seed=0, site=0, linear=False,
             pool_mean=False, stream_dtype=torch.float32, compact=False, gscale=None, target_exp=0):
    """compact (pool='cls' only): dx / dxb are [B, D] — the cls rows alone, nothing zero-filled.
    gscale: f32 [4] device tensor that persists across steps (zeroed once) -> loss-scaled gradients (fp16 operands): dx / dxb come out
    multiplied by the power of two S the kernel picks from their largest magnitude; gscale receives {S, 1/S} for the LoRA-gradient
    reductions, [2] is cleared for the overflow guard of this backward (layernorm_bwd(gmax=gscale[2:])) and [3] carries the exponent in use
    (gsl_head_bwd). target_exp: 0 = the default 11."""
    _need(dlogits, demb, x, gamma, mean, rstd, emb, Wn, gscale)
    if gscale is not None and (gscale.numel() < 4 or gscale.dtype != torch.float32):
        raise RuntimeError("head_bwd: gscale must be a float32 tensor of 4 elements {S, 1/S, seen maximum, exponent}")
    amax_ws = torch.empty(B, device=x.device, dtype=torch.float32) if gscale is not None else None
    rows = B if compact else B * T
    dx = torch.empty(rows, D, device=x.device, dtype=stream_dtype)
    dxb = torch.empty(rows, D, device=x.device, dtype=dtype)
    C = Wn.shape[0] if Wn is not None else 0
    L.check(L.load().gsl_head_bwd(_p(dlogits), _p(demb), _p(x), code(x.dtype), T, _p(gamma), _p(mean), _p(rstd), _p(emb), _p(Wn), _p(dx),
                                  _p(dxb), B, D, C, float(cos_s), code(dtype), code(stream_dtype), float(p_drop), int(seed), int(site),
                                  1 if linear else 0, 1 if pool_mean else 0, 1 if compact else 0, _p(gscale), _p(amax_ws), int(target_exp), _stream()),
            "gsl_head_bwd")
    return dx, dxb


def ce_fwd(logits, labels):
    _need(logits, labels)
    out = torch.empty(2, device=logits.device, dtype=torch.float32)
    ws = torch.empty(2 * logits.shape[0], device=logits.device, dtype=torch.float32)
    L.check(L.load().gsl_ce_fwd(_p(logits), _p(labels), _p(out), _p(ws), logits.shape[0], logits.shape[1], _stream()), "gsl_ce_fwd")
    return out


def ce_bwd(logits, labels, coef, scale, dlogits=None, accumulate=None):
    _need(logits, labels, coef, dlogits)
    acc = (dlogits is not None) if accumulate is None else bool(accumulate)
    if dlogits is None:
        dlogits = torch.empty_like(logits)
    L.check(L.load().gsl_ce_bwd(_p(logits), _p(labels), _p(coef), float(scale), _p(dlogits), logits.shape[0], logits.shape[1],
                                1 if acc else 0, _stream()), "gsl_ce_bwd")
    return dlogits


def proto_kl_fwd(emb, labels, proto):
    _need(emb, labels, proto)
    out = torch.empty(1, device=emb.device, dtype=torch.float32)
    ws = torch.empty(emb.shape[0], device=emb.device, dtype=torch.float32)
    L.check(L.load().gsl_proto_kl_fwd(_p(emb), _p(labels), _p(proto), _p(out), _p(ws), emb.shape[0], emb.shape[1], proto.shape[0],
                                      _stream()), "gsl_proto_kl_fwd")
    return out


def proto_kl_bwd(emb, labels, proto, coef, scale, demb=None, accumulate=None):
    _need(emb, labels, proto, coef, demb)
    acc = (demb is not None) if accumulate is None else bool(accumulate)
    if demb is None:
        demb = torch.empty_like(emb)
    L.check(L.load().gsl_proto_kl_bwd(_p(emb), _p(labels), _p(proto), _p(coef), float(scale), _p(demb), emb.shape[0],
                                      emb.shape[1], proto.shape[0], 1 if acc else 0, _stream()), "gsl_proto_kl_bwd")
    return demb


def group_norms_fwd(flat, toff, tnumel, tgroup, ngroups, tau=0.0):
    _need(flat, toff, tnumel, tgroup)
    dev = flat.device
    nt = toff.numel()
    ws = torch.empty(nt * L.NORM_SPLIT, device=dev, dtype=torch.float32)
    tss = torch.empty(nt, device=dev, dtype=torch.float32)
    gn = torch.empty(ngroups, device=dev, dtype=torch.float32)
    cn = torch.empty(ngroups, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    mask = torch.empty(ngroups, device=dev, dtype=torch.uint8)
    L.check(L.load().gsl_group_norms_fwd(_p(flat), _p(toff), _p(tnumel), _p(tgroup), nt, ngroups, float(tau), _p(ws), _p(tss),
                                         _p(gn), _p(cn), _p(loss), _p(mask), _stream()), "gsl_group_norms_fwd")
    return dict(tensor_sumsq=tss, group_norm=gn, cal_norm=cn, loss=loss, mask=mask)


def group_norms_bwd(flat, toff, tnumel, tgroup, group_norm, coef, scale, gradflat):
    _need(flat, toff, tnumel, tgroup, group_norm, coef, gradflat)
    L.check(L.load().gsl_group_norms_bwd(_p(flat), _p(toff), _p(tnumel), _p(tgroup), toff.numel(), _p(group_norm), _p(coef),
                                         float(scale), _p(gradflat), _stream()), "gsl_group_norms_bwd")


def adamw_flat(p, g, m, v, lr, beta1, beta2, eps, wd, step, guard=None):
    """guard: f32 device element (gscale[2:] of the backward that produced g): the update is skipped when it holds >= 65504 or a non-finite value."""
    _need(p, g, m, v, guard)
    L.check(L.load().gsl_adamw_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                    float(wd), int(step), _p(guard), _stream()), "gsl_adamw_flat")


def adamw_flat_dev(p, g, m, v, lr_dev, b1, b2, eps, wd, step_dev, guard=None):
    """Same update with the step count (int64) and learning rate (f32) read from device memory (HIP-graph replays)."""
    _need(p, g, m, v, lr_dev, step_dev, guard)
    L.check(L.load().gsl_adamw_flat_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr_dev), float(b1), float(b2), float(eps), float(wd),
                                        _p(step_dev), _p(guard), _stream()), "gsl_adamw_flat_dev")


def cast(x, dtype):
    _need(x)
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    L.check(L.load().gsl_cast(_p(x), _p(out), x.numel(), code(dtype), _stream()), "gsl_cast")
    return out


def transpose_cast(W, dtype):
    _need(W)
    R, Cc = W.shape
    out = torch.empty(Cc, R, device=W.device, dtype=dtype)
    L.check(L.load().gsl_transpose_cast(_p(W), _p(out), R, Cc, code(dtype), _stream()), "gsl_transpose_cast")
    return out


def pack_pad(src, si, sj, rows, cols, rows_out, ld_out, dtype, scale=1.0):
    _need(src)
    out = torch.empty(rows_out, ld_out, device=src.device, dtype=dtype)
    L.check(L.load().gsl_pack_pad(_p(src), si, sj, rows, cols, float(scale), _p(out), rows_out, ld_out, code(dtype), _stream()),
            "gsl_pack_pad")
    return out


PACK_DESC = None


def pack_desc_table(entries, device):
    """entries: list of (src f32 tensor, si, sj, rows, cols, scale, out tensor). -> (uint8 device tensor holding gsl_pack_desc[n], max_elems)"""
    import numpy as np
    global PACK_DESC
    if PACK_DESC is None:   # mirrors struct gsl_pack_desc (include/gslora_hip.h), 56 bytes
        PACK_DESC = np.dtype([("in", "<u8"), ("si", "<i8"), ("sj", "<i8"), ("rows", "<i4"), ("cols", "<i4"), ("scale", "<f4"),
                              ("pad_", "<i4"), ("out", "<u8"), ("rows_out", "<i4"), ("ld_out", "<i4")], align=False)
        assert PACK_DESC.itemsize == 56
    arr = np.zeros(len(entries), dtype=PACK_DESC)
    mx = 0
    for k, (src, si, sj, rows, cols, scale, out) in enumerate(entries):
        arr[k] = (src.data_ptr(), si, sj, rows, cols, scale, 0, out.data_ptr(), out.shape[0], out.shape[1])
        mx = max(mx, out.numel())
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device), mx


def pack_pad_batch(table, n, max_elems, dtype):
    _need(table)
    L.check(L.load().gsl_pack_pad_batch(_p(table), int(n), int(max_elems), code(dtype), _stream()), "gsl_pack_pad_batch")


def dropout_mask(n, p_drop, seed, site, device):
    keep = torch.empty(n, device=device, dtype=torch.uint8)
    L.check(L.load().gsl_dropout_mask(_p(keep), n, float(p_drop), int(seed), int(site), _stream()), "gsl_dropout_mask")
    return keep
