#!/usr/bin/env python
"""Is the GEMM K loop bound by the L2 -> LDS DMA stream, and does fetching the W fragments straight into registers relieve it?
Dev build (GSLORA_HIP_LIB=.../libgslora_hip_dev.so). 256x128x64 ring kernel (variant 3) and its ablations against the probe kernel
gemm_bf16_ring3w_kernel (variant 13 full, 14 = the A DMA stream alone, 15 = A DMA + W vector loads alone)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = int(os.environ.get("M", 100864))
for N, K in [(512, 2048), (1536, 512), (512, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = None
    res = {}
    cases = [("v3 full", 3, 0), ("v3 DMA only", 3, 6), ("v3 MFMA only", 3, 11), ("v13 W from L2 (full)", 13, 0), ("v14 A-DMA only", 14, 0), ("v15 A-DMA + W loads only", 15, 0),
             ("v8 8-phase", 8, 0)]
    for rnd in range(3):
        for name, v, abl in cases:
            os.environ["GSL_GEMM_VARIANT"] = str(v); os.environ["GSL_GEMM_ABL"] = str(abl)
            ops.gemm_nt(A, W, out); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.gemm_nt(A, W, out)
            e.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(s.elapsed_time(e) / 10)
            if rnd == 0 and name in ("v3 full", "v13 W from L2 (full)"):
                if ref is None: ref = out.clone()
                else: print("   v13 == v3 bitwise:", bool(torch.equal(ref, out)), " max diff", (ref.float() - out.float()).abs().max().item())
    print(f"N={N} K={K}: " + " | ".join(f"{k}: {min(v)*1e3:.0f} us" for k, v in res.items()), flush=True)
