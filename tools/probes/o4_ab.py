#!/usr/bin/env python
"""The overlap GEMM (csrc/gemm_o4.inc, GSL_O4=1) against the 8-wave 8-phase kernel (GSL_O4=0) on the step's shapes (M = 201 728), dev build:
kernel times (HIP events, interleaved rounds) and cycle stamps of every 64th workgroup (kernel start -> prologue landed -> K loop done ->
epilogue done). GSL_O4_DELAY=<cycles> sets the out-of-phase start of the second workgroup per CU.  usage: o4_ab.py [delay ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = int(os.environ.get("M", 201728))
dbg = torch.zeros(1024, device="cuda", dtype=torch.int64)
os.environ["GSL_P8_STAMPS"] = hex(dbg.data_ptr())
dt = torch.float16
delays = [int(a) for a in sys.argv[1:]] or [6000]
SHAPES = (("QKV (N 1536, K 512, head-major + LN fold)", 1536, 512, 0, "qkv"), ("out-proj dX (N 512, K 512)", 512, 512, 0, "store"),
          ("QKV dX (N 512, K 1536)", 512, 1536, 0, "store"), ("fused FFN1 (N 2048, K 512 + 64, G8)", 2048, 512, 64, "ffn1"))
for name, N, K, K2, kind in SHAPES:
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
    flops = 2.0 * M * N * (K + (8 if K2 else 0))
    # the 8-phase kernel and the stamped overlap kernel run from the dev build; "product" = the overlap kernel of the product library (the dev
    # build's ablation branches sit inside its pinned MFMA groups and cost it ~15 %)
    variants = [("8-wave 8-phase", "0", 0)] + [(f"overlap 4-wave x2, delay {dl} (dev build)", "1", dl) for dl in delays] 
    res = {v[0]: [] for v in variants}
    stamps = {}
    for rnd_ in range(3):
        for vname, o4, dl in variants:
            os.environ["GSL_O4"] = o4; os.environ["GSL_O4_DELAY"] = str(dl)
            import contextlib
            with (contextlib.nullcontext() if o4 == "P" else L.use_dev()):
                for _ in range(2):
                    dbg.zero_(); call()
                torch.cuda.synchronize()
                st = dbg.cpu().view(-1, 4); st = st[(st != 0).all(1)]
                stamps[vname] = st
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    call()
                e1.record(); torch.cuda.synchronize()
            res[vname].append(e0.elapsed_time(e1) / 5 * 1e3)
    for vname, _, _ in variants:
        st = stamps[vname]
        t = sorted(res[vname])[1]
        if st.shape[0] == 0:
            print(f"| {name} | {vname} | {t:.0f} us ({min(res[vname]):.0f} - {max(res[vname]):.0f}) | {flops / t / 1e6:.0f} TF/s = {flops / t / 1e6 / 2500:.3f} | (no stamps) |", flush=True)
            continue
        d = (st[:, 1:] - st[:, :-1]).double(); tot = (st[:, 3] - st[:, 0]).double()
        print(f"| {name} | {vname} | {t:.0f} us ({min(res[vname]):.0f} - {max(res[vname]):.0f}) | {flops / t / 1e6:.0f} TF/s = {flops / t / 1e6 / 2500:.3f} | {st.shape[0]} wgs: prologue {d[:, 0].median():.0f}, "
              f"K loop {d[:, 1].median():.0f}, epilogue {d[:, 2].median():.0f}, total {tot.median():.0f} cycles |", flush=True)
