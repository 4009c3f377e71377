#!/usr/bin/env python
"""GEMM micro-benchmark on the GPU box: every kernel variant (GSL_GEMM_VARIANT) on the step's real shapes,
HIP-event timed, interleaved rounds, checked against torch (hipBLASLt) which also serves as the
same-hardware reference throughput."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch  # noqa: E402
from gslora_hip import _lib as L, ops  # noqa: E402

M = int(os.environ.get("M", 100864))
SHAPES = [("ffn1 512->2048 +lora gelu", 2048, 512, 64, L.EPI_BIAS_GELU), ("ffn2 2048->512 +lora res", 512, 2048, 64, L.EPI_BIAS_RES_F32),
          ("qkv 512->1536", 1536, 512, 0, L.EPI_STORE), ("out 512->512 res", 512, 512, 0, L.EPI_BIAS_RES_F32),
          ("dX 2048->512 mul", 2048, 512, 64, L.EPI_MUL), ("lora-down N=64 K=2048", 64, 2048, 0, L.EPI_STORE)]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1,2").split(",")]
if os.environ.get("SHAPE"):
    SHAPES = [s_ for s_ in SHAPES if s_[0].startswith(os.environ["SHAPE"])]
ROUNDS, ITERS = int(os.environ.get("ROUNDS", 3)), int(os.environ.get("ITERS", 10))
dev = "cuda"
torch.manual_seed(0)
for name, N, K1, K2, epi in SHAPES:
    A1 = torch.randn(M, K1, device=dev).bfloat16(); W1 = (torch.randn(N, K1, device=dev) * K1 ** -0.5).bfloat16()
    A2 = torch.randn(M, K2, device=dev).bfloat16() if K2 else None
    W2 = (torch.randn(N, K2, device=dev) * 0.1).bfloat16() if K2 else None
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); aux = torch.randn(M, N, device=dev).bfloat16()
    f32 = epi in (L.EPI_BIAS_RES_F32,)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == L.EPI_BIAS_GELU else None
    flops = 2.0 * M * N * (K1 + K2)

    def run():
        ops.gemm_nt(A1, W1, out, epilogue=epi, A2=A2, W2=W2, bias=bias if epi != L.EPI_MUL else None, res=res if f32 else None,
                    aux=aux if epi == L.EPI_MUL else None, out2=out2, p_drop=0.1 if epi in (L.EPI_BIAS_GELU, L.EPI_BIAS_RES_F32) else 0.0,
                    seed=1, site=1)
    res_t = {}
    for rnd in range(ROUNDS):
        for v in VARIANTS:
            os.environ["GSL_GEMM_VARIANT"] = str(v)
            run(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(ITERS):
                run()
            e.record(); torch.cuda.synchronize()
            res_t.setdefault(v, []).append(s.elapsed_time(e) / ITERS)
    # correctness of the last variant vs variant 0 on a plain STORE_F32 pass
    chk = {}
    for v in VARIANTS:
        os.environ["GSL_GEMM_VARIANT"] = str(v)
        o = torch.empty(M, N, device=dev, dtype=torch.float32)
        ops.gemm_nt(A1, W1, o, epilogue=L.EPI_STORE_F32, A2=A2, W2=W2)
        chk[v] = o
    ref = A1.float() @ W1.float().t() + (A2.float() @ W2.float().t() if K2 else 0)
    errs = {v: ((chk[v] - ref).abs().max() / ref.abs().max()).item() for v in VARIANTS}
    # hipBLASLt reference (plain bf16 matmul, no epilogue)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    (A1 @ W1.t()); torch.cuda.synchronize(); t0.record()
    for _ in range(10):
        (A1 @ W1.t())
    t1.record(); torch.cuda.synchronize()
    bl = t0.elapsed_time(t1) / 10
    line = f"{name:28s} M={M} N={N} K={K1}+{K2}: " + "  ".join(
        f"v{v}: {min(res_t[v]) * 1e3:7.1f} us {flops / min(res_t[v]) / 1e9:7.1f} TF (err {errs[v]:.1e})" for v in VARIANTS)
    print(line + f"  | torch.matmul {bl * 1e3:7.1f} us {2.0 * M * N * K1 / bl / 1e9:7.1f} TF", flush=True)
