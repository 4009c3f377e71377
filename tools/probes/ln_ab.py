#!/usr/bin/env python
"""LayerNorm forward / backward at the bench shape (M = 201 728, D = 512): time and achieved bandwidth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
M, D = 201728, 512
torch.manual_seed(0)
x = torch.randn(M, D, device="cuda"); g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
dy = torch.randn(M, D, device="cuda").bfloat16(); dres = torch.randn(M, D, device="cuda").bfloat16()
y, mean, rstd = ops.layernorm_fwd(x, D, M, D, g, b, 1e-5, torch.bfloat16)


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


tf = t(lambda: ops.layernorm_fwd(x, D, M, D, g, b, 1e-5, torch.bfloat16))
tb = t(lambda: ops.layernorm_bwd(dy, x, D, g, mean, rstd, dres, p_drop=0.1, seed=3, site=5))
print(f"ln_fwd {tf:6.1f} us ({(M * D * 6 + M * 8) / tf / 1e6:.2f} TB/s)   ln_bwd {tb:6.1f} us ({M * D * (2 + 4 + 2 + 2 + 2) / tb / 1e6:.2f} TB/s)")
