#!/usr/bin/env python
"""QKV projection GEMM (M = 201 728, N = 1536, K = 512): plain row-major STORE against the head-major permuting store."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
B, T, H = 1024, 197, 8
M, N, K = B * T, 3 * H * 64, 512
torch.manual_seed(0)
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)


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


for _ in range(3):
    print(f"STORE {t(lambda: ops.gemm_nt(A, W, out)):7.1f} us   STORE_QKV_HM {t(lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_STORE_QKV_HM, T=T)):7.1f} us", flush=True)
