#!/usr/bin/env python
"""Does the last partial round of 256x256 tiles cost a whole round?  N = 512 GEMMs of the step have 1 576 tiles on 256 CUs (6.16 rounds).
Times gsl_gemm_nt (STORE) at M = 196 608 (exactly 6 rounds), 201 728 (the step), and the 5 120-row remainder alone (small-tile kernel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch  # noqa: E402
from gslora_hip import _lib as L, ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
MS = [int(m) for m in os.environ.get("MS", "196608,201728,5120,10240,20480,32768").split(",")]
print("| N | K | M | tiles 256x256 | rounds on 256 CUs | us | us per full round |\n|---|---|---|---|---|---|---|")
for N, K in ((512, 2048), (512, 512), (512, 1536), (2048, 512)):
    W = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    A = torch.randn(max(MS), K, device=dev).bfloat16()
    out = torch.empty(max(MS), N, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for M in MS:
            a, o = A[:M], out[:M]
            ops.gemm_nt(a, W, o, epilogue=L.EPI_STORE); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.gemm_nt(a, W, o, epilogue=L.EPI_STORE)
            e.record(); torch.cuda.synchronize()
            res.setdefault(M, []).append(s.elapsed_time(e) / 20 * 1e3)
    for M in MS:
        t = min(res[M]); tiles = ((M + 255) // 256) * (N // 256)
        print(f"| {N} | {K} | {M} | {tiles} | {tiles / 256:.2f} | {t:.1f} | {t / max(1, tiles // 256):.1f} |")
