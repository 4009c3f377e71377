#!/usr/bin/env python
"""Launch the fused FFN1 GEMM (and the FFN2 forward / FFN2-dX shapes) a few times: the subject of PMC sweeps over the launcher knobs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = 201728
torch.manual_seed(0)
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
A1, W1 = bf(M, 512), bf(2048, 512, sc=512 ** -0.5)
A2, W2 = bf(M, 64), bf(2048, 64, sc=0.1); A2[:, 8:] = 0
bias = torch.randn(2048, device="cuda")
h, gp = torch.empty(M, 2048, device="cuda", dtype=torch.bfloat16), torch.empty(M, 2048, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm_nt(A1, W1, h, epilogue=L.EPI_BIAS_GELU, A2=A2, W2=W2, bias=bias, out2=gp, p_drop=0.1, seed=7, site=5)
torch.cuda.synchronize()
