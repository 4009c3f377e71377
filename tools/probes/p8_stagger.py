#!/usr/bin/env python
"""Does de-synchronising the workgroups of the 8-phase GEMM (GSL_P8_STAGGER) overlap the HBM-heavy epilogues with other CUs' K loops?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = 201728
torch.manual_seed(0)
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()


def timeit(fn, n=10):
    for _ in range(2): fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


def mk(N, K1, K2, epi):
    A1, W1 = bf(M, K1), bf(N, K1, sc=K1 ** -0.5)
    A2 = W2 = None
    if K2:
        A2, W2 = bf(M, K2), bf(N, K2, sc=0.1); A2[:, 8:] = 0
    kw = {}
    if epi in (L.EPI_BIAS_GELU, L.EPI_BIAS_RES_F32): kw["bias"] = torch.randn(N, device="cuda")
    if epi == L.EPI_BIAS_GELU: kw.update(out2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), p_drop=0.1, seed=7, site=5)
    if epi == L.EPI_BIAS_RES_F32: kw.update(res=torch.randn(M, N, device="cuda"), p_drop=0.1, seed=7, site=5)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == L.EPI_BIAS_RES_F32 else torch.bfloat16)
    return lambda: ops.gemm_nt(A1, W1, out, epilogue=epi, A2=A2, W2=W2, **kw)


cases = {"FFN1 fused": mk(2048, 512, 64, L.EPI_BIAS_GELU), "QKV": mk(1536, 512, 0, L.EPI_STORE), "FFN1-dX K2048": mk(512, 2048, 0, L.EPI_STORE),
         "proj bias+res": mk(512, 512, 0, L.EPI_BIAS_RES_F32), "FFN2 bias+res K2048+64": mk(512, 2048, 64, L.EPI_BIAS_RES_F32)}
stags = [0, 1, 2, 3, 4, 6, 8]
print(f"{'shape':26s}" + "".join(f"{'stag ' + str(s):>10s}" for s in stags) + "   (us; stag x 8128 cycles)")
for name, fn in cases.items():
    row = []
    for sg in stags:
        os.environ["GSL_P8_STAGGER"] = str(sg)
        row.append(timeit(fn))
    print(f"{name:26s}" + "".join(f"{t:10.1f}" for t in row))
