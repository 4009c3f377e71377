#!/usr/bin/env python
"""Cycle stamps of the ping-pong GEMM (GSL_PP_ABL bit 256): per-phase durations of workgroup 0, waves 0 (group 0) and 4 (group 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M, N, K1, K2 = 201728, 2048, 512, 64
torch.manual_seed(0)
A1 = torch.randn(M, K1, device="cuda").bfloat16(); W1 = (torch.randn(N, K1, device="cuda") * K1 ** -0.5).bfloat16()
A2 = torch.randn(M, K2, device="cuda").bfloat16(); A2[:, 8:] = 0; W2 = (torch.randn(N, K2, device="cuda") * 0.1).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); out2 = torch.empty_like(out)
base = int(os.environ.get("ABL", "0"))
os.environ["GSL_GEMM_VARIANT"] = "10"
os.environ["GSL_PP_ABL"] = str(base | 256)
dbg = torch.zeros(4096, device="cuda", dtype=torch.float32)
for _ in range(3):
    ops.gemm_nt(A1, W1, out, epilogue=L.EPI_BIAS_GELU, A2=A2, W2=W2, bias=bias, out2=out2, p_drop=0.1, seed=7, site=5, res=dbg)
torch.cuda.synchronize()
st = dbg.view(torch.int64).cpu()
NB = 2 * ((K1 + K2) // 64)
for g in range(2):
    t = st[g * 1024:(g + 1) * 1024]
    t = t[t != 0]
    d = (t[1:] - t[:-1]).tolist()
    print(f"group {g}: {len(t)} stamps; slot durations (cycles):", [int(t[min(len(t) - 1, (i + 1) * NB)] - t[i * NB]) for i in range(min(6, len(t) // NB))])
    for sl in range(min(4, len(d) // NB)):
        print(f"  slot {sl}: phases", d[sl * NB:(sl + 1) * NB])
