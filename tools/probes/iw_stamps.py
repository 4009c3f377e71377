#!/usr/bin/env python
"""Per-step cycle split of the in-wave pipelined GEMM (variant 11, GSL_P8_STAMPS): MFMA block / rest of the step / barrier wait,
for wave row 0 (MFMAs last) and wave row 1 (MFMAs first), workgroup 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = 201728
torch.manual_seed(0)
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
os.environ["GSL_GEMM_VARIANT"] = "11"
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
for name, N, K1, K2, epi in (("qkv store", 1536, 512, 0, L.EPI_STORE), ("ffn1 fused", 2048, 512, 64, L.EPI_BIAS_GELU)):
    A1, W1 = bf(M, K1), bf(N, K1, sc=K1 ** -0.5)
    A2 = W2 = None
    kw = {}
    if K2:
        A2, W2 = bf(M, K2), bf(N, K2, sc=0.1); A2[:, 8:] = 0
    if epi == L.EPI_BIAS_GELU:
        kw.update(bias=torch.randn(N, device="cuda"), out2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), p_drop=0.1, seed=7, site=5)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for abl in os.environ.get("ABLS", "0,512,3").split(","):
        os.environ["GSL_PP_ABL"] = abl
        for _ in range(2):
            dbg.zero_()
            ops.gemm_nt(A1, W1, out, epilogue=epi, A2=A2, W2=W2, **kw)
        torch.cuda.synchronize()
        d = dbg.cpu()[:8].view(2, 4).double()
        for r in range(2):
            n = max(d[r, 3].item(), 1)
            print(f"{name:11s} abl={abl:>3s} row {r}: per step  MFMA block {d[r,0]/n:6.0f}  rest {d[r,1]/n:6.0f}  barrier {d[r,2]/n:6.0f}  = {d[r,:3].sum()/n:6.0f} cycles  ({int(n)} steps)")
