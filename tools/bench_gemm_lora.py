#!/usr/bin/env python
"""In-kernel-LoRA GEMM (gsl_gemm_nt_lora) A/B: single-phase 256x256 kernel (GSL_GEMM_VARIANT=4) vs the 8-phase schedule (default),
on the step's shapes; checked against an fp32 torch reference of A W^T + s (A P^T) Q^T."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch  # noqa: E402
from gslora_hip import _lib as L, ops  # noqa: E402

M = int(os.environ.get("M", 201728))
SHAPES = [("ffn2 fwd 2048->512 res", 512, 2048, L.EPI_BIAS_RES_F32), ("ffn2 dX 512->2048 mul", 2048, 512, L.EPI_MUL),
          ("ffn1 dX 2048->512 store", 512, 2048, L.EPI_STORE)]
dev = "cuda"
torch.manual_seed(0)
for name, N, K, epi in SHAPES:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    P = torch.zeros(16, K, device=dev).bfloat16(); P[:8] = (torch.randn(8, K, device=dev) * K ** -0.5).bfloat16()
    Q = torch.zeros(N, 32, device=dev).bfloat16(); Q[:, :8] = (torch.randn(N, 8, device=dev) * 0.3).bfloat16()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); aux = torch.randn(M, N, device=dev).bfloat16()
    f32 = epi == L.EPI_BIAS_RES_F32
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    tout = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)

    def run():
        ops.gemm_nt_lora(A, W, P, Q, 0.125, tout, out, epilogue=epi, bias=bias if f32 else None, res=res if f32 else None,
                         aux=aux if epi == L.EPI_MUL else None, p_drop=0.0)
    tm = {}
    outs = {}
    for rnd in range(3):
        for v in ("4", "8"):
            os.environ["GSL_GEMM_VARIANT"] = v
            run(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run()
            e.record(); torch.cuda.synchronize()
            tm.setdefault(v, []).append(s.elapsed_time(e) / 10)
            outs[v] = (out.float().clone(), tout.float().clone())
    rows = slice(0, 4096)
    t_ref = 0.125 * (A[rows].float() @ P.float().t())
    ref = A[rows].float() @ W.float().t() + t_ref.bfloat16().float()[:, :16] @ Q.float()[:, :16].t()
    if f32:
        ref = ref + bias + res[rows]
    elif epi == L.EPI_MUL:
        ref = ref * aux[rows].float()
    errs = {v: ((outs[v][0][rows] - ref).abs().max() / ref.abs().max()).item() for v in tm}
    terr = {v: (outs[v][1][rows, :16] - t_ref).abs().max().item() for v in tm}
    same = torch.equal(outs["4"][0], outs["8"][0]) and torch.equal(outs["4"][1], outs["8"][1])
    fl = 2.0 * M * N * K
    print(f"{name:26s} M={M}: " + "  ".join(f"v{v}: {min(t) * 1e3:7.1f} us {fl / min(t) / 1e9:6.1f} TF (err {errs[v]:.1e}, t err {terr[v]:.1e})"
                                            for v, t in tm.items()) + f"  bitwise-equal={same}", flush=True)
os.environ.pop("GSL_GEMM_VARIANT", None)
