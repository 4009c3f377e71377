#!/usr/bin/env python
"""Cycle stamps of the plain-store GEMMs on the step's shapes (M = 201 728): the 4-wave 32x32x16 kernel (csrc/gemm_w4.inc) against the 8-wave
8-phase kernel (GSL_W4=0), dev build. Every 64th workgroup records kernel start -> prologue landed -> K loop done -> epilogue done."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
os.environ["GSLORA_HIP_LIB"] = os.path.join(ROOT, "gs-lora_amd", "gslora_hip", "libgslora_hip_dev.so")
import torch
from gslora_hip import _lib as L, ops
M = 201728
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
dt = torch.float16
for name, N, K, T in (("QKV (N 1536, K 512, head-major)", 1536, 512, 197), ("out-proj dX (N 512, K 512)", 512, 512, 0), ("QKV dX (N 512, K 1536)", 512, 1536, 0)):
    A = torch.randn(M, K, device="cuda").to(dt); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    for w4 in ("1", "0"):
        os.environ["GSL_W4"] = w4
        for _ in range(3):
            dbg.zero_()
            ops.gemm_nt(A, W, out, epilogue=L.EPI_STORE_QKV_HM if T else L.EPI_STORE, T=T)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_nt(A, W, out, epilogue=L.EPI_STORE_QKV_HM if T else L.EPI_STORE, T=T)
        e1.record(); torch.cuda.synchronize()
        st = dbg.cpu().view(-1, 4); st = st[(st != 0).all(1)]
        d = (st[:, 1:] - st[:, :-1]).double(); tot = (st[:, 3] - st[:, 0]).double()
        print(f"| {name} | {'4-wave 32x32x16' if w4 == '1' else '8-wave 8-phase'} | {e0.elapsed_time(e1) / 5 * 1e3:.0f} us | {st.shape[0]} wgs: prologue {d[:, 0].median():.0f}, "
              f"K loop {d[:, 1].median():.0f} ({d[:, 1].median() / (K / 64):.0f} per K tile), epilogue {d[:, 2].median():.0f}, total {tot.median():.0f} cycles |", flush=True)
