#!/usr/bin/env python
"""The skinny LoRA down-projection GEMMs of the step (N = 64 padded columns): time against their bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
M = 201728
torch.manual_seed(0)


def t(fn, n=20):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


for K in (512, 2048):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(64, K, device="cuda").bfloat16()
    out = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: ops.gemm_nt(A, W, out, alpha=0.125))
    print(f"u = s A P^T, K = {K}: {us:6.1f} us  ({(M * K * 2 + M * 128) / us / 1e6:.2f} TB/s)")
