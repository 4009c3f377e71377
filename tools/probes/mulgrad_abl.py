#!/usr/bin/env python
"""The gradient-fused FFN2-dX GEMM (gsl_gemm_nt_lora_mulgrad), kernel time and stamps, with parts of its epilogue removed: the HBM writes
(dev build, GSL_STORE_MODE=3) and — with the ablation variant (python -m gslora_hip.build --dev --variant mgabl -DGSL_MG_ABL=1, GSLORA_HIP_LIB
pointing at it, MG_ABL=1) — the operand loads (2), the transposed reads + reduction MFMAs (4), the dZ / h / U1 LDS hand-over (8) and the f32
staging round trip (16), the rank-r tail's Q loads + MFMAs (32), the t store of the first N tile (64), the final hand-over and partial stores (128). Ablated results are wrong by design; only the times mean something."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
os.environ.setdefault("GSLORA_HIP_LIB", os.path.join(ROOT, "gs-lora_amd", "gslora_hip", "libgslora_hip_dev.so"))
import torch
from gslora_hip import _lib as L, ops
M, N, K, r = 201728, 2048, 512, 8
dbg = torch.zeros(1024 + 2048, device="cuda", dtype=torch.int64)
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
cases = [("1", 0), ("3", 0), ("1", 0), ("3", 0)]
if os.environ.get("MG_ABL"):
    cases = [("1", 0), ("1", 0), ("1", 32), ("1", 64), ("1", 128), ("1", 2), ("1", 16), ("1", 12), ("3", 30), ("3", 30 + 32), ("3", 30 + 32 + 64), ("3", 254), ("1", 0)]
if os.environ.get("MG_ABL") == "phases":
    cases = [("1", 0), ("1", 0), ("1", 2), ("3", 0), ("1", 12), ("1", 0)]
if os.environ.get("MG_ABL") == "stagger":
    cases = [("1", 0), ("1", 0), ("1", 0x100), ("1", 0x200), ("1", 0x300), ("1", 0), ("1", 0x100), ("1", 0x200), ("1", 0x300), ("1", 0)]
for mode, mask in cases:
    os.environ["GSL_STORE_MODE"] = mode
    os.environ["GSL_O4_DELAY"] = str(mask)
    for _ in range(2):
        dbg.zero_(); call()
    torch.cuda.synchronize()
    st = dbg[:1024].cpu().view(-1, 4); ph = dbg[1024:].cpu().view(-1, 8)[(st != 0).all(1)][:, :5].double(); st = st[(st != 0).all(1)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    d_ = (st[:, 1:] - st[:, :-1]).double(); tot = (st[:, 3] - st[:, 0]).double()
    print(f"| FFN2-dX gradient-fused | store mode {mode}, ablation mask {mask} | {e0.elapsed_time(e1) / 10 * 1e3:.0f} us (incl. the two reduce launches) | prologue {d_[:, 0].median():.0f}, K loop + rank-r tail {d_[:, 1].median():.0f}, epilogue {d_[:, 2].median():.0f}, total {tot.median():.0f} |" + (f" wave 0 phases: staging {ph[:, 0].median():.0f}, rows {ph[:, 1].median():.0f}, requests {ph[:, 2].median():.0f}, reductions {ph[:, 3].median():.0f}, final {ph[:, 4].median():.0f}" if ph.any() else ""), flush=True)
