#!/usr/bin/env python
"""Is the fused FFN1 epilogue bound inside the CU or by the chip-wide store stream? Epilogue cycles (stamps of workgroup 0, 64, ...) of the
same kernel at growing M: a few workgroups on an idle chip ... all 256 CUs busy for many rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
torch.manual_seed(0)
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
os.environ["GSL_GEMM_VARIANT"] = "8"
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
N, K1, K2 = 2048, 512, 64
W1, W2, bias = bf(N, K1, sc=K1 ** -0.5), bf(N, K2, sc=0.1), torch.randn(N, device="cuda")
for M in (256, 2048, 8192, 32768, 201728):
    A1, A2 = bf(M, K1), bf(M, K2); A2[:, 8:] = 0
    out, out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for p in (0.1,):
        for _ in range(3):
            dbg.zero_()
            ops.gemm_nt(A1, W1, out, epilogue=L.EPI_BIAS_GELU, A2=A2, W2=W2, bias=bias, out2=out2, p_drop=p, seed=7, site=5)
        torch.cuda.synchronize()
        st = dbg.cpu().view(-1, 4); st = st[(st != 0).all(1)]
        d = (st[:, 1:] - st[:, :-1]).double()
        print(f"M={M:7d} ({(M // 256) * 8:5d} workgroups): prologue {d[:,0].median():6.0f}  K loop {d[:,1].median():6.0f}  epilogue {d[:,2].median():6.0f}  ({st.shape[0]} stamped)")

# the gradient-fused FFN2-dX (MUL epilogue + in-kernel LoRA + two LoRA-gradient partial reductions)
print("gradient-fused FFN2-dX:")
r, mlp, d = 8, 2048, 512
w2T = bf(mlp, d, sc=d ** -0.5)
P = torch.zeros(16, d, device="cuda", dtype=torch.bfloat16); P[:r] = bf(r, d, sc=0.1)
Q = torch.zeros(mlp, 32, device="cuda", dtype=torch.bfloat16); Q[:, :r] = bf(mlp, r, sc=0.1)
for M in (256, 2048, 8192, 32768, 201728):
    dy = bf(M, d)
    v2 = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
    da, gp, h = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16), bf(M, mlp), bf(M, mlp)
    u1 = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16); u1[:, :r] = bf(M, r)
    G1, G2 = torch.zeros(mlp * r, device="cuda"), torch.zeros(r * mlp, device="cuda")
    for _ in range(3):
        dbg.zero_()
        ops.gemm_nt_lora_mulgrad(dy, w2T, P, Q, 1.0, v2, da, gp, u1, G1, (r, 1), h, G2, (1, mlp), r)
    torch.cuda.synchronize()
    st = dbg.cpu().view(-1, 4); st = st[(st != 0).all(1)]
    dd = (st[:, 1:] - st[:, :-1]).double()
    print(f"M={M:7d}: prologue {dd[:,0].median():6.0f}  K loop {dd[:,1].median():6.0f}  epilogue {dd[:,2].median():6.0f}  ({st.shape[0]} stamped)")
