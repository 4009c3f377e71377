#!/usr/bin/env python
"""Round-3 cycle stamps of the 8-phase GEMM forms the step runs now (dev build: GSLORA_HIP_LIB=.../libgslora_hip_dev.so; GSL_P8_STAMPS =
device address of a 256 x 4 u64 buffer): every 64th workgroup records kernel start -> prologue landed -> K loop done -> epilogue done.
M = 201 728 rows. One launch at a time (idle chip apart from the kernel itself)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = 201728
torch.manual_seed(0)
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()


def report(name):
    torch.cuda.synchronize()
    st = dbg.cpu().view(-1, 4)
    st = st[(st != 0).all(1)]
    d = (st[:, 1:] - st[:, :-1]).double()
    tot = (st[:, 3] - st[:, 0]).double()
    print(f"| {name} | {d[:,0].median():.0f} | {d[:,1].median():.0f} | {d[:,2].median():.0f} | {tot.median():.0f} |", flush=True)
    dbg.zero_()


r, mlp, d = 8, 2048, 512
print("| GEMM of the step (round 3 form) | prologue | K loop | epilogue | total |\n|---|---|---|---|---|")
# fused FFN1: K segment, bias + GELU + 8-bit GELU' + dropout
xn2, w1 = bf(M, d), bf(mlp, d, sc=d ** -0.5)
u1 = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16); u1[:, :r] = bf(M, r)
b1p = torch.zeros(mlp, 64, device="cuda", dtype=torch.bfloat16); b1p[:, :r] = bf(mlp, r, sc=0.1)
b1 = torch.randn(mlp, device="cuda")
h = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16)
for g8 in (False, True):
    gp = torch.empty(M, mlp, device="cuda", dtype=torch.uint8 if g8 else torch.bfloat16)
    for _ in range(3):
        dbg.zero_()
        ops.gemm_nt(xn2, w1, h, epilogue=L.EPI_BIAS_GELU_G8 if g8 else L.EPI_BIAS_GELU, A2=u1, W2=b1p, bias=b1, out2=gp, p_drop=0.1, seed=7, site=5)
    report("FFN1 fused, GELU' as " + ("8-bit code" if g8 else "bf16"))
# gradient-fused FFN2-dX
dy, w2T = bf(M, d), bf(mlp, d, sc=d ** -0.5)
P = torch.zeros(16, d, device="cuda", dtype=torch.bfloat16); P[:r] = bf(r, d, sc=0.1)
Q = torch.zeros(mlp, 32, device="cuda", dtype=torch.bfloat16); Q[:, :r] = bf(mlp, r, sc=0.1)
v2 = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
da = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16)
G1, G2 = torch.zeros(mlp * r, device="cuda"), torch.zeros(r * mlp, device="cuda")
for g8 in (False, True):
    gp = torch.randint(0, 253, (M, mlp), device="cuda", dtype=torch.uint8) if g8 else bf(M, mlp)
    for _ in range(3):
        dbg.zero_()
        ops.gemm_nt_lora_mulgrad(dy, w2T, P, Q, 1.0, v2, da, gp, u1, G1, (r, 1), h, G2, (1, mlp), r, p_drop=0.1)
    report("FFN2-dX x GELU' + fused LoRA-gradient reductions, GELU' as " + ("8-bit code" if g8 else "bf16"))
# out-proj and FFN2 forward with the bf16 residual stream
o, wo, bo = bf(M, d), bf(d, d, sc=d ** -0.5), torch.randn(d, device="cuda")
for bfs in (False, True):
    res = torch.randn(M, d, device="cuda"); res = res.bfloat16() if bfs else res
    out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16 if bfs else torch.float32)
    for _ in range(3):
        dbg.zero_()
        ops.gemm_nt(o, wo, out, epilogue=L.EPI_BIAS_RES_BF16 if bfs else L.EPI_BIAS_RES_F32, bias=bo, res=res, p_drop=0.1, seed=7, site=5)
    report("out-proj, residual stream " + ("bf16" if bfs else "f32"))
P2 = torch.zeros(16, mlp, device="cuda", dtype=torch.bfloat16); P2[:r] = bf(r, mlp, sc=0.05)
Q2 = torch.zeros(d, 32, device="cuda", dtype=torch.bfloat16); Q2[:, :r] = bf(d, r, sc=0.1)
w2 = bf(d, mlp, sc=mlp ** -0.5)
for bfs in (False, True):
    res = torch.randn(M, d, device="cuda"); res = res.bfloat16() if bfs else res
    out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16 if bfs else torch.float32)
    for _ in range(3):
        dbg.zero_()
        ops.gemm_nt_lora(h, w2, P2, Q2, 1.0, v2, out, epilogue=L.EPI_BIAS_RES_BF16 if bfs else L.EPI_BIAS_RES_F32, bias=bo, res=res, p_drop=0.1, seed=3, site=9)
    report("FFN2 forward (LoRA in kernel), residual stream " + ("bf16" if bfs else "f32"))
