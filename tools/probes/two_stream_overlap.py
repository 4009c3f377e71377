#!/usr/bin/env python
"""Can a memory-bound kernel of one half-batch run UNDER a GEMM of the other half? Same process, two HIP streams:
the fused FFN1 GEMM (one 512-thread / 144 KB-LDS workgroup per CU, 222 VGPRs) on stream 1, LayerNorm forward / backward or the attention
backward (no or little LDS) on stream 2. Reported: each alone, both back to back on one stream, both concurrently on two streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
torch.manual_seed(0)
Bh, T, D, mlp, H = 512, 197, 512, 2048, 8          # one HALF of the 512 + 512 step
M = Bh * T
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
xn2, w1, b1 = bf(M, D), bf(mlp, D, sc=D ** -0.5), torch.randn(mlp, device="cuda")
u1 = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16); u1[:, :8] = bf(M, 8)
b1p = torch.zeros(mlp, 64, device="cuda", dtype=torch.bfloat16); b1p[:, :8] = bf(mlp, 8, sc=0.1)
h = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16); gp = torch.empty(M, mlp, device="cuda", dtype=torch.uint8)
x = bf(M, D); g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
dy, dres = bf(M, D), bf(M, D)
_, mean, rstd = ops.layernorm_fwd(x, D, M, D, g, b, 1e-5, torch.bfloat16)
qkv = bf(M, 3 * H * 64); d_o = bf(M, H * 64)
o, lse = ops.attention_fwd(qkv, Bh, T, H, D ** -0.5)
gemm = lambda: ops.gemm_nt(xn2, w1, h, epilogue=L.EPI_BIAS_GELU_G8, A2=u1, W2=b1p, bias=b1, out2=gp, p_drop=0.1, seed=7, site=5)
others = {"LayerNorm forward x4": lambda: [ops.layernorm_fwd(x, D, M, D, g, b, 1e-5, torch.bfloat16) for _ in range(4)],
          "LayerNorm backward x2": lambda: [ops.layernorm_bwd(dy, x, D, g, mean, rstd, dres, p_drop=0.1, seed=5, site=3) for _ in range(2)],
          "attention backward": lambda: ops.attention_bwd(qkv, o, d_o, lse, Bh, T, H, D ** -0.5)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


def both_serial(other):
    gemm(); other()


def both_concurrent(other):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        gemm()
    with torch.cuda.stream(s2):
        other()
    cur.wait_stream(s1); cur.wait_stream(s2)


tg = timed(gemm)
print(f"half-batch M = {M} rows; fused FFN1 GEMM alone: {tg:.0f} us\n")
print("| memory-bound partner | partner alone us | GEMM + partner, one stream us | two streams us | saved us | saved % of the pair |")
print("|---|---|---|---|---|---|")
for name, fn in others.items():
    to = timed(fn)
    ts = timed(lambda: both_serial(fn))
    tc = timed(lambda: both_concurrent(fn))
    print(f"| {name} | {to:.0f} | {ts:.0f} | {tc:.0f} | {ts - tc:.0f} | {100 * (ts - tc) / ts:.1f} |", flush=True)
