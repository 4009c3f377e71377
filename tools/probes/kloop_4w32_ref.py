"""Production 8-phase GEMM (gsl_gemm_nt, STORE epilogue) on the shapes of tools/probes/kloop_4w32.hip: whole-kernel time for comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
for dt in (torch.bfloat16, torch.float16):
    for M, N, K in ((8192, 8192, 512), (8192, 8192, 2048), (65536, 2048, 512), (65536, 512, 2048), (65536, 1536, 512), (65536, 512, 512), (65536, 512, 1536)):
        A = (torch.rand(M, K, device="cuda") - 0.5).to(dt); W = (torch.rand(N, K, device="cuda") - 0.5).to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        for _ in range(3): ops.gemm_nt(A, W, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_nt(A, W, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"| production 8-phase (8 waves, 16x16x32), {dt} | {M} x {N} x {K} | {ms:.3f} | {2.0 * M * N * K / ms / 1e9:.0f} |", flush=True)
