#!/usr/bin/env python
"""Ablations of the overlap GEMM (csrc/gemm_o4.inc; ablation build: `cd gs-lora_amd && python -m gslora_hip.build --dev --variant o4abl -DGSL_O4_ABL=1`
-> build_variants/libgslora_hip_o4abl.so): GSL_PF bits 1 no global stores, 2 no epilogue, 4 no operand requests, 8 no MFMAs,
16 no fragment reads; GSL_O4_ONE_PER_CU=1 = one workgroup per CU (16 KB of dynamic LDS on top). Kernel time + stamps, step shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
os.environ.setdefault("GSLORA_HIP_LIB", os.path.join(ROOT, "build_variants", "libgslora_hip_o4abl.so"))
import torch
from gslora_hip import _lib as L, ops
M = int(os.environ.get("M", 201728))
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
os.environ["GSL_O4"] = "1"
dt = torch.float16
SHAPES = (("QKV (N 1536, K 512, HM + LN)", 1536, 512, 0, "qkv"), ("fused FFN1 (N 2048, K 576, G8)", 2048, 512, 64, "ffn1"))
VAR = [("full", 0, 0), ("no stores", 1, 0), ("no epilogue", 2, 0), ("no epilogue, no requests", 6, 0), ("no epilogue, no MFMA", 10, 0), ("no epilogue, no reads", 18, 0),
       ("no epilogue, MFMA only", 22, 0), ("no requests", 4, 0), ("no MFMA", 8, 0),
       ("full, 1 WG/CU", 0, 1), ("no epilogue, 1 WG/CU", 2, 1), ("no epilogue, MFMA only, 1 WG/CU", 22, 1), ("no requests, 1 WG/CU", 4, 1)]
for name, N, K, K2, kind in SHAPES:
    A = torch.randn(M, K, device="cuda").to(dt); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    if kind == "qkv":
        mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda"); c = W.float().sum(1).contiguous(); d = torch.zeros(N, device="cuda")
        call = lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_STORE_QKV_HM_LN, T=197, pos=mean, cls=rstd, aux=c, bias=d)
    else:
        a2 = torch.randn(M, 64, device="cuda"); a2[:, 8:] = 0; A2 = a2.to(dt); W2 = (torch.randn(N, 64, device="cuda") * 0.1).to(dt)
        bias = torch.randn(N, device="cuda"); q = torch.empty(M, N, device="cuda", dtype=torch.uint8)
        call = lambda: ops.gemm_nt(A, W, out, epilogue=L.EPI_BIAS_GELU_G8, A2=A2, W2=W2, bias=bias, out2=q, p_drop=0.1, seed=7, site=5)
    for vname, pf, one in VAR:
        os.environ["GSL_PF"] = str(pf); os.environ["GSL_O4_ONE_PER_CU"] = str(one); os.environ["GSL_O4_DELAY"] = "0"
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
        print(f"| {name} | {vname} | {e0.elapsed_time(e1) / 5 * 1e3:.0f} us | prologue {d_[:, 0].median():.0f}, K loop {d_[:, 1].median():.0f}, epilogue {d_[:, 2].median():.0f}, total {tot.median():.0f} |", flush=True)
