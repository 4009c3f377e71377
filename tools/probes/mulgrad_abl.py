#!/usr/bin/env python
"""The gradient-fused FFN2-dX GEMM (gsl_gemm_nt_lora_mulgrad) with and without its HBM writes (dev build, GSL_STORE_MODE=3): kernel time and stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
os.environ.setdefault("GSLORA_HIP_LIB", os.path.join(ROOT, "gs-lora_amd", "gslora_hip", "libgslora_hip_dev.so"))
import torch
from gslora_hip import _lib as L, ops
M, N, K, r = 201728, 2048, 512, 8
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
dt = torch.float16
A = torch.randn(M, K, device="cuda").to(dt); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
P = torch.zeros(16, K, device="cuda"); P[:r] = torch.randn(r, K, device="cuda") * K ** -0.5; P = P.to(dt)
Q = torch.zeros(N, 32, device="cuda"); Q[:, :r] = torch.randn(N, r, device="cuda") * 0.3; Q = Q.to(dt)
tout = torch.empty(M, 64, device="cuda", dtype=dt); out = torch.empty(M, N, device="cuda", dtype=dt)
aux = torch.randint(0, 250, (M, N), device="cuda", dtype=torch.uint8)
U1 = torch.randn(M, 16, device="cuda").to(dt); Y2 = torch.randn(M, N, device="cuda").to(dt)
G1 = torch.zeros(N, r, device="cuda"); G2 = torch.zeros(r, N, device="cuda")
call = lambda: ops.gemm_nt_lora_mulgrad(A, W, P, Q, 1.0 / r, tout, out, aux, U1, G1, (r, 1), Y2, G2, (1, N), r, p_drop=0.1)
for mode in ("1", "3", "1", "3"):
    os.environ["GSL_STORE_MODE"] = mode
    for _ in range(2):
        dbg.zero_(); call()
    torch.cuda.synchronize()
    st = dbg.cpu().view(-1, 4); st = st[(st != 0).all(1)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record(); torch.cuda.synchronize()
    d_ = (st[:, 1:] - st[:, :-1]).double(); tot = (st[:, 3] - st[:, 0]).double()
    print(f"| FFN2-dX gradient-fused | store mode {mode} | {e0.elapsed_time(e1) / 5 * 1e3:.0f} us (incl. the two reduce launches) | prologue {d_[:, 0].median():.0f}, K loop + rank-r tail {d_[:, 1].median():.0f}, epilogue {d_[:, 2].median():.0f}, total {tot.median():.0f} |", flush=True)
