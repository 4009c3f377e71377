#!/usr/bin/env python
"""Workgroup timeline of the 8-phase GEMM (dev build, GSL_P8_STAMPS_ALL=1): every workgroup records start / prologue / K loop / end
cycle stamps plus the CU it ran on. Per CU: the chain of tiles, the gaps between one tile's end and the next tile's start, the
span from the CU's first start to its last end; the shader clock under this load = span cycles / kernel wall time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch, numpy as np
from gslora_hip import _lib as L, ops
M = int(os.environ.get("M", 201728))
torch.manual_seed(0)
NB = 32768
dbg = torch.zeros(NB * 4, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
os.environ["GSL_P8_STAMPS_ALL"] = "1"
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()


def run(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = dbg.cpu().numpy().view(np.uint64).reshape(-1, 4)
    st = st[st[:, 3] != 0]
    cu = (st[:, 0] >> np.uint64(48)).astype(np.int64)
    t0 = (st[:, 0] & np.uint64(0xffffffffffff)).astype(np.int64)
    t1, t2, t3 = (st[:, i].astype(np.int64) for i in (1, 2, 3))
    dur = t3 - t0
    gaps, spans, busy, ntile = [], [], [], []
    for c in np.unique(cu):
        m = cu == c
        o = np.argsort(t0[m]); a, b = t0[m][o], t3[m][o]
        gaps += list(a[1:] - b[:-1]); spans.append(b.max() - a.min()); busy.append((b - a).sum()); ntile.append(m.sum())
    spans, busy, gaps = np.array(spans), np.array(busy), np.array(gaps)
    print(f"| {name} | {ms*1e3:.0f} | {len(st)} | {len(np.unique(cu))} | {np.median(ntile):.0f} ({min(ntile)}-{max(ntile)}) | {np.median(dur):.0f} | "
          f"{np.median(t1-t0):.0f} / {np.median(t2-t1):.0f} / {np.median(t3-t2):.0f} | {np.median(gaps):.0f} (p90 {np.percentile(gaps,90):.0f}) | "
          f"{np.median(spans):.0f} | {np.median(busy/spans):.3f} | {np.median(spans) / (ms * 1e3) / 1e3:.2f} |", flush=True)


print("| GEMM | wall us | tiles | CUs seen | tiles per CU | tile cycles (median) | prologue / K loop / epilogue | gap between tiles on one CU | CU span cycles | busy / span | GHz = CU span / wall |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
r, mlp, d = 8, 2048, 512
xn2, w1 = bf(M, d), bf(mlp, d, sc=d ** -0.5)
u1 = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16); u1[:, :r] = bf(M, r)
b1p = torch.zeros(mlp, 64, device="cuda", dtype=torch.bfloat16); b1p[:, :r] = bf(mlp, r, sc=0.1)
b1 = torch.randn(mlp, device="cuda")
h = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16)
gp = torch.empty(M, mlp, device="cuda", dtype=torch.uint8)
run("FFN1 fused (8-bit GELU')", lambda: ops.gemm_nt(xn2, w1, h, epilogue=L.EPI_BIAS_GELU_G8, A2=u1, W2=b1p, bias=b1, out2=gp, p_drop=0.1, seed=7, site=5))
dy, w2T = bf(M, d), bf(mlp, d, sc=d ** -0.5)
P = torch.zeros(16, d, device="cuda", dtype=torch.bfloat16); P[:r] = bf(r, d, sc=0.1)
Q = torch.zeros(mlp, 32, device="cuda", dtype=torch.bfloat16); Q[:, :r] = bf(mlp, r, sc=0.1)
v2 = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
da = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16)
G1, G2 = torch.zeros(mlp * r, device="cuda"), torch.zeros(r * mlp, device="cuda")
gpr = torch.randint(0, 253, (M, mlp), device="cuda", dtype=torch.uint8)
run("FFN2-dX x GELU' + LoRA-gradient reductions", lambda: ops.gemm_nt_lora_mulgrad(dy, w2T, P, Q, 1.0, v2, da, gpr, u1, G1, (r, 1), h, G2, (1, mlp), r, p_drop=0.1))
o, wo, bo = bf(M, d), bf(d, d, sc=d ** -0.5), torch.randn(d, device="cuda")
res = torch.randn(M, d, device="cuda").bfloat16(); out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
run("out-proj (bf16 stream)", lambda: ops.gemm_nt(o, wo, out, epilogue=L.EPI_BIAS_RES_BF16, bias=bo, res=res, p_drop=0.1, seed=7, site=5))
wq = bf(3 * d, d, sc=d ** -0.5); qkv = torch.empty(M, 3 * d, device="cuda", dtype=torch.bfloat16)
run("QKV forward (plain store)", lambda: ops.gemm_nt(xn2, wq, qkv, epilogue=L.EPI_STORE))
wqT = bf(d, 3 * d, sc=(3 * d) ** -0.5); dx = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
run("QKV dX (K = 1536)", lambda: ops.gemm_nt(qkv, wqT, dx, epilogue=L.EPI_STORE))
P2 = torch.zeros(16, mlp, device="cuda", dtype=torch.bfloat16); P2[:r] = bf(r, mlp, sc=0.05)
Q2 = torch.zeros(d, 32, device="cuda", dtype=torch.bfloat16); Q2[:, :r] = bf(d, r, sc=0.1)
w2 = bf(d, mlp, sc=mlp ** -0.5)
run("FFN2 forward (K = 2048, LoRA in kernel)", lambda: ops.gemm_nt_lora(h, w2, P2, Q2, 1.0, v2, out, epilogue=L.EPI_BIAS_RES_BF16, bias=bo, res=res, p_drop=0.1, seed=3, site=9))
