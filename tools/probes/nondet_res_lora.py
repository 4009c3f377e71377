#!/usr/bin/env python
"""Repeat the in-kernel-LoRA residual GEMM of tests/test_hip_ops.py::test_gemm_tail_split_* and report run-to-run differences (a race shows up as rows that change)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M, N, r = 201728, 512, 8
K = int(os.environ.get("K", "512"))
dt = torch.float16 if os.environ.get("DT", "f16") == "f16" else torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(5)
A = torch.randn(M, K, device="cuda", generator=g).to(dt)
W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
res = torch.randn(M, N, device="cuda", generator=g).to(torch.float16)
bias = torch.randn(N, device="cuda", generator=g)
P = torch.zeros(16, K, device="cuda"); P[:r] = torch.randn(r, K, device="cuda", generator=g) * K ** -0.5
Q = torch.zeros(N, 32, device="cuda"); Q[:, :r] = torch.randn(N, r, device="cuda", generator=g) * 0.3
P, Q = P.to(dt), Q.to(dt)
ref = None
bad = 0
for it in range(int(os.environ.get("REPS", "40"))):
    o = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    t = torch.full((M, 64), float("nan"), device="cuda", dtype=dt)
    o1 = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
    t1 = torch.full((M, 64), float("nan"), device="cuda", dtype=dt)
    ops.gemm_nt_lora(A, W, P, Q, 1.0 / r, t1, o1)
    ops.gemm_nt_lora(A, W, P, Q, 1.0 / r, t, o, epilogue=L.EPI_BIAS_RES_F16, bias=bias, res=res, p_drop=0.1, seed=77, site=6)
    torch.cuda.synchronize()
    cur = (o, t, o1, t1)
    if ref is None:
        ref = cur
        continue
    for name, a, b in zip(("res-out", "res-t", "store-out", "store-t"), cur, ref):
        if not torch.equal(a, b):
            d = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
            rows = d.any(1).nonzero().flatten()
            cols = d.any(0).nonzero().flatten()
            bad += 1
            print(f"iter {it}: {name} differs: {int(d.sum())} elements, rows {rows[:6].tolist()} .. {rows[-3:].tolist()} ({len(rows)}), cols {cols[:4].tolist()} .. {cols[-2:].tolist()} ({len(cols)}), "
                  f"nan in cur {int(torch.isnan(a.float()).sum())} ref {int(torch.isnan(b.float()).sum())}", flush=True)
print("lib", os.environ.get("GSLORA_HIP_LIB", "product"), "K", K, "mismatching (iteration, tensor) pairs:", bad)
