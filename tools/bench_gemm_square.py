#!/usr/bin/env python
"""Calibrate the GEMM variants on square shapes (guide reference points) with random [-1,1) operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
for n in (4096, 8192):
    A = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16(); W = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for v in (1, 3, 4):
            os.environ["GSL_GEMM_VARIANT"] = str(v)
            ops.gemm_nt(A, W, out); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): ops.gemm_nt(A, W, out)
            e.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(s.elapsed_time(e) / 5)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        (A @ W.t()); torch.cuda.synchronize(); s.record()
        for _ in range(5): (A @ W.t())
        e.record(); torch.cuda.synchronize()
        res.setdefault("hipblaslt", []).append(s.elapsed_time(e) / 5)
    print(f"{n}^3: " + "  ".join(f"{k}: {2.0 * n ** 3 / min(v) / 1e9:7.1f} TF" for k, v in res.items()), flush=True)
