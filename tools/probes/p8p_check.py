#!/usr/bin/env python
"""Persistent 8-phase GEMM (gemm_bf16_p8p_kernel) against the per-tile kernel: run under two builds (GSLORA_HIP_LIB), outputs saved and
compared bit for bit by the second run; timings of the step's plain-store shapes.   python tools/probes/p8p_check.py OUT.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
torch.manual_seed(0)
M = 201728
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
shapes = {"QKV forward (head-major store)": (512, 1536, True), "out-proj dX": (512, 512, False), "QKV dX": (1536, 512, False), "FFN1 dX (no LoRA)": (2048, 512, False),
          "ragged M = 33490": (512, 1536, False)}
outs = {}
for name, (K, N, hm) in shapes.items():
    m = 33490 if "ragged" in name else M
    A, W = bf(m, K), bf(N, K, sc=K ** -0.5)
    out = torch.empty(m, N, device="cuda", dtype=torch.bfloat16)
    fn = (lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_STORE_QKV_HM, T=197)) if hm else (lambda: ops.gemm_nt(A, W, out))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    print(f"{name:34s} M={m} K={K} N={N}: {us:7.1f} us  {2.0 * m * N * K / us / 1e6:6.1f} TF/s", flush=True)
    outs[name] = out.cpu()
path = sys.argv[1]
if os.path.exists(path):
    ref = torch.load(path)
    for k in outs:
        print(f"  bit-identical to the other build: {k}: {torch.equal(outs[k], ref[k])}")
else:
    torch.save(outs, path)
