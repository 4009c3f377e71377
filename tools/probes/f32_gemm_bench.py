#!/usr/bin/env python
"""f32 parity GEMM (gemm_f32_mfma_kernel) on the step's shapes: time and TF/s (python tools/probes/f32_gemm_bench.py; GSLORA_HIP_LIB selects a build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
torch.manual_seed(0)
M = 201728
for name, K, N, epi in (("QKV", 512, 1536, "store"), ("FFN1 (bias + exact-erf GELU, 2 outputs)", 512, 2048, "gelu"), ("FFN2 (bias + residual)", 2048, 512, "res"), ("FFN1 dX", 2048, 512, "store")):
    A, W = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5
    out, out2, bias, res = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda"), torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    fn = {"store": lambda: ops.gemm_nt(A, W, out), "gelu": lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_BIAS_GELU, bias=bias, out2=out2),
          "res": lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_BIAS_RES_F32, bias=bias, res=res)}[epi]
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 200
    print(f"{name:42s} K={K} N={N}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
