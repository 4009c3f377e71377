#!/usr/bin/env python
"""Wave quantisation of the N = 512 GEMMs: 788 x 2 = 1576 tiles of 256x256 on 256 CUs are 6.16 rounds -> 7. Does running the rows of the
partial round (20 M-tiles = 5120 rows) as a second, concurrent launch on a side stream (128x128 tiles) shorten the GEMM?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import _lib as L, ops
M = 201728
torch.manual_seed(0)
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
side = torch.cuda.Stream()


def timeit(fn, n=20):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


for name, N, K, epi in (("out-proj bias+res", 512, 512, L.EPI_BIAS_RES_F32), ("proj-dX store", 512, 512, L.EPI_STORE), ("QKV-dX store", 512, 1536, L.EPI_STORE),
                        ("FFN1-dX store K=2048", 512, 2048, L.EPI_STORE)):
    A, W = bf(M, K), bf(N, K, sc=K ** -0.5)
    kw = {}
    if epi == L.EPI_BIAS_RES_F32:
        kw = dict(bias=torch.randn(N, device="cuda"), res=torch.randn(M, N, device="cuda"), p_drop=0.1, seed=7, site=5)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == L.EPI_BIAS_RES_F32 else torch.bfloat16)
    tiles = (M // 256) * (N // 256)
    full_rounds = tiles // 256
    m_main = (full_rounds * 256 // (N // 256)) * 256
    ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()

    def whole():
        ops.gemm_nt(A, W, out, epilogue=epi, **kw)

    def split():
        ev_f.record()
        with torch.cuda.stream(side):
            side.wait_event(ev_f)
            kt = dict(kw)
            if "res" in kt: kt["res"] = kt["res"][m_main:]
            ops.gemm_nt(A[m_main:], W, out[m_main:], epilogue=epi, **kt)
            ev_j.record()
        km = dict(kw)
        if "res" in km: km["res"] = km["res"][:m_main]
        ops.gemm_nt(A[:m_main], W, out[:m_main], epilogue=epi, **km)
        torch.cuda.current_stream().wait_event(ev_j)

    def split_seq():
        km = dict(kw)
        if "res" in km: km["res"] = km["res"][:m_main]
        ops.gemm_nt(A[:m_main], W, out[:m_main], epilogue=epi, **km)
        kt = dict(kw)
        if "res" in kt: kt["res"] = kt["res"][m_main:]
        ops.gemm_nt(A[m_main:], W, out[m_main:], epilogue=epi, **kt)
    print(f"{name:22s}: {tiles} tiles = {tiles / 256:.2f} rounds; one launch {timeit(whole):7.1f} us   main {m_main} rows + tail {M - m_main} rows: "
          f"concurrent {timeit(split):7.1f} us   back to back {timeit(split_seq):7.1f} us")
