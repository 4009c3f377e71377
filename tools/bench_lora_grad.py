#!/usr/bin/env python
"""lora_grad micro-benchmark: HBM rate of the skinny TN reduction on the step's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
M = int(os.environ.get("M", 201728))
for N in (2048, 512):
    Y = torch.randn(M, N, device="cuda").bfloat16(); U = torch.randn(M, 64, device="cuda").bfloat16(); U[:, 8:] = 0
    G = torch.zeros(N * 8, device="cuda")
    ops.lora_grad(Y, U, G, 8, 1, 8, accumulate=False); torch.cuda.synchronize()
    ref = (Y[:20000].float().t() @ U[:20000, :8].float())
    G2 = torch.zeros(N * 8, device="cuda"); ops.lora_grad(Y[:20000].contiguous(), U[:20000].contiguous(), G2, 8, 1, 8, accumulate=False)
    err = ((G2.view(N, 8) - ref).abs().max() / ref.abs().max()).item()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.lora_grad(Y, U, G, 8, 1, 8, accumulate=True)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e3
    print(f"N={N}: {t:7.1f} us  {M * N * 2 / t / 1e6:5.2f} TB/s (err {err:.1e})", flush=True)
