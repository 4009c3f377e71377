#!/usr/bin/env python
"""Ablation of the ring3 GEMM main loop (STORE epilogue): which of DMA / LDS reads / MFMA / barrier bounds it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
os.environ["GSL_GEMM_VARIANT"] = "3"
M = 100864
for N, K in [(512, 2048), (1536, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for abl, name in [(0, "full"), (1, "noDMA"), (2, "noLDSread"), (4, "noMFMA"), (3, "noDMA+noLDS (MFMA+barrier only)"),
                          (5, "noDMA+noMFMA (LDS reads only)"), (6, "noLDS+noMFMA (DMA only)"), (9, "noDMA+nobarrier"), (11, "MFMA only")]:
            os.environ["GSL_GEMM_ABL"] = str(abl)
            ops.gemm_nt(A, W, out); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.gemm_nt(A, W, out)
            e.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(s.elapsed_time(e) / 10)
    print(f"N={N} K={K}: " + " | ".join(f"{k}: {min(v)*1e3:.0f} us" for k, v in res.items()), flush=True)
