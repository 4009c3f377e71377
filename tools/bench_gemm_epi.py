#!/usr/bin/env python
"""Epilogue store-pattern probe: same main loop, bf16 (8-byte stores) vs f32 (16-byte stores) output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = 100864
for N, K in [(1536, 512), (2048, 512), (512, 2048)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); of = torch.empty(M, N, device="cuda", dtype=torch.float32)
    res = {}
    for rnd in range(3):
        for v in os.environ.get("VARIANTS", "4,9").split(","):
            os.environ["GSL_GEMM_VARIANT"] = v
            for name, fn in [("bf16", lambda: ops.gemm_nt(A, W, ob)), ("f32", lambda: ops.gemm_nt(A, W, of, epilogue=L.EPI_STORE_F32))]:
                fn(); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                res.setdefault(f"v{v}/{name}", []).append(s.elapsed_time(e) / 10 * 1e3)
    print(f"N={N} K={K}: " + "  ".join(f"{k}: {min(v):6.1f} us" for k, v in res.items()), flush=True)
