#!/usr/bin/env python
"""The STORE-epilogue GEMMs of the step at M = 201 728 (QKV forward, out-proj dX, QKV dX, FFN1-dX) against hipBLASLt on the same shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = int(os.environ.get("M", 201728))
torch.manual_seed(0)
for name, N, K in (("qkv fwd 512->1536", 1536, 512), ("out-proj dX 512->512", 512, 512), ("qkv dX 1536->512", 512, 1536), ("qkv dX (K,V only) 1024->512", 512, 1024),
                   ("ffn1 dX 2048->512", 512, 2048), ("kv fwd 512->1024", 1024, 512)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def t(fn, n=10):
        for _ in range(2): fn()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / n * 1e3)
        return best
    ours = t(lambda: ops.gemm_nt(A, W, out))
    bl = t(lambda: torch.matmul(A, W.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"{name:30s} ours {ours:7.1f} us {fl / ours / 1e6:7.1f} TF/s | hipBLASLt {bl:7.1f} us {fl / bl / 1e6:7.1f} TF/s", flush=True)
