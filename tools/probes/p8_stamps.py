#!/usr/bin/env python
"""Cycle stamps of the 8-phase GEMM (GSL_P8_STAMPS = device address of a 256 x 4 u64 buffer): every 64th workgroup records kernel
start -> prologue landed -> K loop done -> epilogue done. Shapes of the bench step (M = 201 728 rows)."""
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
    print(f"{name:34s}: {st.shape[0]:3d} wgs; median cycles: prologue {d[:,0].median():6.0f}  K loop {d[:,1].median():6.0f}  "
          f"epilogue {d[:,2].median():6.0f}  total {tot.median():6.0f}")
    dbg.zero_()


def plain(name, N, K1, K2, epi, f32out=False):
    A1, W1 = bf(M, K1), bf(N, K1, sc=K1 ** -0.5)
    A2 = W2 = None
    if K2:
        A2, W2 = bf(M, K2), bf(N, K2, sc=0.1); A2[:, 8:] = 0
    kw = {}
    if epi in (L.EPI_BIAS_GELU, L.EPI_BIAS_RES_F32):
        kw["bias"] = torch.randn(N, device="cuda")
    if epi == L.EPI_BIAS_GELU:
        kw.update(out2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), p_drop=0.1, seed=7, site=5)
    if epi == L.EPI_BIAS_RES_F32:
        kw.update(res=torch.randn(M, N, device="cuda"), p_drop=0.1, seed=7, site=5)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32out else torch.bfloat16)
    for _ in range(3):
        dbg.zero_()
        ops.gemm_nt(A1, W1, out, epilogue=epi, A2=A2, W2=W2, **kw)
    report(name)


plain("FFN1 fused (K 512+64, N 2048)", 2048, 512, 64, L.EPI_BIAS_GELU)
plain("QKV store (K 512, N 1536)", 1536, 512, 0, L.EPI_STORE)
plain("FFN1-dX store (K 2048, N 512)", 512, 2048, 0, L.EPI_STORE)
plain("proj bias+res f32 (K 512, N 512)", 512, 512, 0, L.EPI_BIAS_RES_F32, f32out=True)
plain("FFN2 bias+res f32 (K 2048+64)", 512, 2048, 64, L.EPI_BIAS_RES_F32, f32out=True)

# gradient-fused FFN2-dX (MUL epilogue + in-kernel LoRA + the two LoRA-gradient partial reductions), as vit_runner calls it
try:
    import inspect
    print(inspect.signature(ops.gemm_nt_lora_mulgrad))
except Exception as ex:
    print(ex)

# in-kernel LoRA forms, as vit_runner calls them: P [16, K] (rows >= r zero), Q [N, 32] (columns >= r zero)
r, mlp, d = 8, 2048, 512
dy, w2T = bf(M, d), bf(mlp, d, sc=d ** -0.5)
P = torch.zeros(16, d, device="cuda", dtype=torch.bfloat16); P[:r] = bf(r, d, sc=0.1)
Q = torch.zeros(mlp, 32, device="cuda", dtype=torch.bfloat16); Q[:, :r] = bf(mlp, r, sc=0.1)
v2 = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
da, gp, h = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16), bf(M, mlp), bf(M, mlp)
u1 = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16); u1[:, :r] = bf(M, r)
G1, G2 = torch.zeros(mlp * r, device="cuda"), torch.zeros(r * mlp, device="cuda")
for _ in range(3):
    dbg.zero_()
    ops.gemm_nt_lora_mulgrad(dy, w2T, P, Q, 1.0, v2, da, gp, u1, G1, (r, 1), h, G2, (1, mlp), r)
report("FFN2-dX grad-fused MUL (K 512)")
for _ in range(3):
    dbg.zero_()
    ops.gemm_nt_lora(dy, w2T, P, Q, 1.0, v2, da, epilogue=L.EPI_MUL, aux=gp)
report("FFN2-dX MUL, LoRA in kernel")
# FFN2 forward with the LoRA term in the kernel: K = 2048, bias + residual f32
x1 = torch.randn(M, d, device="cuda"); b2 = torch.randn(d, device="cuda")
P2 = torch.zeros(16, mlp, device="cuda", dtype=torch.bfloat16); P2[:r] = bf(r, mlp, sc=0.05)
Q2 = torch.zeros(d, 32, device="cuda", dtype=torch.bfloat16); Q2[:, :r] = bf(d, r, sc=0.1)
y = torch.empty(M, d, device="cuda")
w2 = bf(d, mlp, sc=mlp ** -0.5)
for _ in range(3):
    dbg.zero_()
    ops.gemm_nt_lora(h, w2, P2, Q2, 1.0, v2, y, epilogue=L.EPI_BIAS_RES_F32, bias=b2, res=x1, p_drop=0.1, seed=3, site=9)
report("FFN2 fwd bias+res, LoRA in kernel")
