import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/gs-lora_amd"]
import torch
from gslora_hip import ops
M, D = 201728, 512
dt = torch.float16
x = torch.randn(M, D, device="cuda").to(dt); dy = torch.randn(M, D, device="cuda").to(dt); dres = torch.randn(M, D, device="cuda").to(dt)
gam = torch.ones(D, device="cuda")
_, mean, rstd = ops.layernorm_fwd(x, D, M, D, gam, torch.zeros(D, device="cuda"), 1e-5, dt)
gmax = torch.zeros(1, device="cuda")
res = {0: [], 1: []}
for rnd in range(5):
    for g in (0, 1):
        for _ in range(3):
            ops.layernorm_bwd(dy, x, D, gam, mean, rstd, dres, p_drop=0.1, seed=1, site=2, gmax=gmax if g else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.layernorm_bwd(dy, x, D, gam, mean, rstd, dres, p_drop=0.1, seed=1, site=2, gmax=gmax if g else None)
        e1.record(); torch.cuda.synchronize()
        res[g].append(e0.elapsed_time(e1) / 20 * 1e3)
print("ln_bwd us without guard", [round(v, 1) for v in res[0]], "with guard", [round(v, 1) for v in res[1]])
