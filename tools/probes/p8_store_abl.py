#!/usr/bin/env python
"""What the 8-phase kernel's epilogues cost without their HBM writes (dev build, GSL_STORE_MODE=3 discards every staged store; 0 plain, 1 non-temporal
= the product's, 2 sc1): kernel time and cycle stamps (prologue / K loop / epilogue) on the step's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
os.environ.setdefault("GSLORA_HIP_LIB", os.path.join(ROOT, "gs-lora_amd", "gslora_hip", "libgslora_hip_dev.so"))
import torch
from gslora_hip import _lib as L, ops
M = 201728
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
dt = torch.float16
for name, N, K, kind in (("fused FFN1 (N 2048, K 576, G8)", 2048, 512, "ffn1"), ("QKV (N 1536, K 512, HM + LN)", 1536, 512, "qkv"), ("out-proj dX (N 512)", 512, 512, "store")):
    A = torch.randn(M, K, device="cuda").to(dt); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    if kind == "qkv":
        mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda"); c = W.float().sum(1).contiguous(); d = torch.zeros(N, device="cuda")
        call = lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_STORE_QKV_HM_LN, T=197, pos=mean, cls=rstd, aux=c, bias=d)
    elif kind == "ffn1":
        a2 = torch.randn(M, 64, device="cuda"); a2[:, 8:] = 0; A2 = a2.to(dt); W2 = (torch.randn(N, 64, device="cuda") * 0.1).to(dt)
        bias = torch.randn(N, device="cuda"); q = torch.empty(M, N, device="cuda", dtype=torch.uint8)
        call = lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_BIAS_GELU_G8, A2=A2, W2=W2, bias=bias, out2=q, p_drop=0.1, seed=7, site=5)
    else:
        call = lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_STORE)
    for mode in ("1", "3", "0", "1", "3"):
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
        print(f"| {name} | store mode {mode} | {e0.elapsed_time(e1) / 5 * 1e3:.0f} us | prologue {d_[:, 0].median():.0f}, K loop {d_[:, 1].median():.0f}, epilogue {d_[:, 2].median():.0f}, total {tot.median():.0f} |", flush=True)
