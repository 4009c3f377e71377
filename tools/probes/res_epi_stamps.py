#!/usr/bin/env python
"""Cycle stamps (dev build, GSL_P8_STAMPS) of the two residual-epilogue GEMMs of the forward in the step's fp16 shapes: out-proj (K 512, N 512) and
FFN2 with the LoRA term in the kernel (K 2048, N 512); x_out = fp16(dropout(acc + bias) + x_in). GSL_STORE_MODE=3 discards the output stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
os.environ.setdefault("GSLORA_HIP_LIB", os.path.join(ROOT, "gs-lora_amd", "gslora_hip", "libgslora_hip_dev.so"))
import torch
from gslora_hip import _lib as L, ops
M, D, MLP, r = 201728, 512, 2048, 8
dt = torch.float16
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(dt)
x_in, out, bias = rn(M, D), torch.empty(M, D, device="cuda", dtype=dt), torch.randn(D, device="cuda")
a_proj, w_proj = rn(M, D), rn(D, D, sc=D ** -0.5)
h, w2 = rn(M, MLP), rn(D, MLP, sc=MLP ** -0.5)
P2 = torch.zeros(16, MLP, device="cuda", dtype=dt); P2[:r] = rn(r, MLP, sc=0.05)
Q2 = torch.zeros(D, 32, device="cuda", dtype=dt); Q2[:, :r] = rn(D, r, sc=0.1)
tout = torch.empty(M, 64, device="cuda", dtype=dt)
calls = {
    "out-proj fwd (K 512)": lambda: ops.gemm_nt(a_proj, w_proj, out, epilogue=L.EPI_BIAS_RES_F16, bias=bias, res=x_in, p_drop=0.1, seed=3, site=9),
    "FFN2 fwd, LoRA in kernel (K 2048)": lambda: ops.gemm_nt_lora(h, w2, P2, Q2, 1.0 / r, tout, out, epilogue=L.EPI_BIAS_RES_F16, bias=bias, res=x_in, p_drop=0.1, seed=3, site=9),
}
for rep in range(2):
    for name, call in calls.items():
        for mode in ("1", "3"):
            os.environ["GSL_STORE_MODE"] = mode
            for _ in range(2):
                dbg.zero_(); call()
            torch.cuda.synchronize()
            st = dbg.cpu().view(-1, 4); st = st[(st != 0).all(1)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record(); torch.cuda.synchronize()
            d_ = (st[:, 1:] - st[:, :-1]).double(); tot = (st[:, 3] - st[:, 0]).double()
            print(f"| {name} | store mode {mode} | {e0.elapsed_time(e1) / 10 * 1e3:.0f} us | prologue {d_[:, 0].median():.0f}, K loop {d_[:, 1].median():.0f}, "
                  f"epilogue {d_[:, 2].median():.0f}, total {tot.median():.0f} |", flush=True)
