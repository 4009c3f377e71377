#!/usr/bin/env python
"""In-kernel LoRA on the 64x64 ring kernel against the two launches it replaces, at few-shot row counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
torch.manual_seed(0)
dt = torch.bfloat16


def t(fn, n=50):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


for M in (1576, 8):
    for N, K in ((512, 2048), (2048, 512), (512, 512)):
        A = torch.randn(M, K, device="cuda").to(dt); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        P16 = torch.zeros(16, K, device="cuda", dtype=dt); P16[:8] = (torch.randn(8, K, device="cuda") * K ** -0.5).to(dt)
        P64 = torch.zeros(64, K, device="cuda", dtype=dt); P64[:16] = P16
        Q32 = torch.zeros(N, 32, device="cuda", dtype=dt); Q32[:, :8] = (torch.randn(N, 8, device="cuda") * 0.3).to(dt)
        Q64 = torch.zeros(N, 64, device="cuda", dtype=dt); Q64[:, :32] = Q32
        tout = torch.empty(M, 64, device="cuda", dtype=dt); out = torch.empty(M, N, device="cuda", dtype=dt)
        two = lambda: (ops.gemm_nt(A, P64, tout, alpha=0.125), ops.gemm_nt(A, W, out, A2=tout, W2=Q64))
        one = lambda: ops.gemm_nt_lora(A, W, P16, Q32, 0.125, tout, out)
        plain = lambda: ops.gemm_nt(A, W, out)
        skinny = lambda: ops.gemm_nt(A, P64, tout, alpha=0.125)
        print(f"M {M} N {N} K {K}: plain {t(plain):.1f} us, skinny {t(skinny):.1f}, two launches {t(two):.1f}, in-kernel {t(one):.1f}", flush=True)
