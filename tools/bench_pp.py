#!/usr/bin/env python
"""Ping-pong GEMM kernel (GSL_GEMM_VARIANT=10) against the 256x256 8-phase kernel (8) on the step's shapes: results compared
(different K summation order: tolerance of one bf16 rounding), HIP-event timed in interleaved rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch  # noqa: E402
from gslora_hip import _lib as L, ops  # noqa: E402

M = int(os.environ.get("M", 201728))
SHAPES = [("ffn1 gelu+lora", 2048, 512, 64, L.EPI_BIAS_GELU, 0.1), ("qkv store", 1536, 512, 0, L.EPI_STORE, 0.0),
          ("ffn1 gelu no-drop", 2048, 512, 64, L.EPI_BIAS_GELU, 0.0), ("dx 2048->512 store", 512, 2048, 0, L.EPI_STORE, 0.0)]
if os.environ.get("SHAPE"):
    SHAPES = [s_ for s_ in SHAPES if s_[0].startswith(os.environ["SHAPE"])]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "8,10").split(",")]
ROUNDS, ITERS = int(os.environ.get("ROUNDS", 3)), int(os.environ.get("ITERS", 10))
torch.manual_seed(0)
dev = "cuda"
for name, N, K1, K2, epi, p in SHAPES:
    A1 = torch.randn(M, K1, device=dev).bfloat16(); W1 = (torch.randn(N, K1, device=dev) * K1 ** -0.5).bfloat16()
    A2 = W2 = None
    if K2:
        A2 = torch.randn(M, K2, device=dev).bfloat16(); A2[:, 8:] = 0
        W2 = (torch.randn(N, K2, device=dev) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev) if epi == L.EPI_BIAS_GELU else None
    outs = {}
    times = {v: [] for v in VARIANTS}
    for v in VARIANTS:
        outs[v] = (torch.zeros(M, N, device=dev, dtype=torch.bfloat16), torch.zeros(M, N, device=dev, dtype=torch.bfloat16) if epi == L.EPI_BIAS_GELU else None)

    def run(v):
        os.environ["GSL_GEMM_VARIANT"] = str(v)
        ops.gemm_nt(A1, W1, outs[v][0], epilogue=epi, A2=A2, W2=W2, bias=bias, out2=outs[v][1], p_drop=p, seed=7, site=5)
    for rnd in range(ROUNDS):
        for v in VARIANTS:
            run(v); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(ITERS):
                run(v)
            e.record(); torch.cuda.synchronize()
            times[v].append(s.elapsed_time(e) / ITERS)
    ref = outs[VARIANTS[0]]
    msg = []
    for v in VARIANTS[1:]:
        d = (outs[v][0].float() - ref[0].float()).abs()
        tol = 2.0 ** -7 * ref[0].float().abs() + 2e-3
        bad = int((d > tol).sum())
        zeros_match = bool(((outs[v][0] == 0) == (ref[0] == 0)).all()) if p > 0 else True
        d2 = 0.0 if outs[v][1] is None else (outs[v][1].float() - ref[1].float()).abs().max().item()
        msg.append(f"v{v} vs v{VARIANTS[0]}: max|d| {d.max().item():.4f} bad {bad} zeros_match {zeros_match} out2 max|d| {d2:.4f}")
    flops = 2.0 * M * N * (K1 + (8 if K2 else 0))
    print(f"{name:22s} M={M} N={N} K={K1}+{K2}: " + "  ".join(f"v{v}: {min(times[v]) * 1e3:7.1f} us {flops / min(times[v]) / 1e9:6.0f} TF" for v in VARIANTS)
          + "  | " + "; ".join(msg), flush=True)
