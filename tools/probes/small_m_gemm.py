#!/usr/bin/env python
"""Small-M GEMMs of the launch-bound regime (few-shot 4+4: M = 1576 rows; batch 48+48: M = 18912) timed as HIP-graph replays of 20
launches (no Python launch overhead): the library's tile choice against forced variants (dev build: GSLORA_HIP_LIB=..._dev.so,
GSL_GEMM_VARIANT) and hipBLASLt (torch.matmul) on the same shape.   M=1576 VARIANTS=0,1,8,12 python tools/probes/small_m_gemm.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch  # noqa: E402
from gslora_hip import _lib as L, ops  # noqa: E402

M = int(os.environ.get("M", 1576))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]      # 0 = the library's own choice
SHAPES = [("qkv 512->1536", 1536, 512, 0), ("out-proj 512->512", 512, 512, 0), ("ffn1 512->2048 (+64)", 2048, 512, 64),
          ("ffn2 2048->512 (+64)", 512, 2048, 64), ("qkv-dX 1536->512", 512, 1536, 0), ("ffn2-dX 512->2048 (+64)", 2048, 512, 64)]
torch.manual_seed(0)
N_IT = 20


def graph_time(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N_IT):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / N_IT * 1e3)
    return best


for name, N, K1, K2 in SHAPES:
    A1 = torch.randn(M, K1, device="cuda").bfloat16(); W1 = (torch.randn(N, K1, device="cuda") * K1 ** -0.5).bfloat16()
    A2 = torch.randn(M, 64, device="cuda").bfloat16() if K2 else None
    W2 = (torch.randn(N, 64, device="cuda") * 0.1).bfloat16() if K2 else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = A1.float() @ W1.float().t() + (A2.float() @ W2.float().t() if K2 else 0)
    row = []
    for v in VARIANTS:
        if v:
            os.environ["GSL_GEMM_VARIANT"] = str(v)
        else:
            os.environ.pop("GSL_GEMM_VARIANT", None)
        fn = lambda: ops.gemm_nt(A1, W1, out, A2=A2, W2=W2)
        t = graph_time(fn)
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        row.append(f"v{v}: {t:6.1f} us (err {err:.0e})")
    tb = graph_time(lambda: torch.matmul(A1, W1.t(), out=out))
    print(f"{name:26s} M={M} N={N} K={K1}+{K2}: " + "  ".join(row) + f"  | hipBLASLt {tb:6.1f} us", flush=True)
