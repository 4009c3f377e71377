#!/usr/bin/env python
"""Attention micro-benchmark (B=512, T=197, H=8): fwd / bwd timings, optional staging ablation (GSL_ATTN_ABL)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
B, T, H = int(os.environ.get("B", 512)), 197, 8
qkv = (torch.randn(B * T, 3 * H * 64, device="cuda")).bfloat16()
d_o = torch.randn(B * T, H * 64, device="cuda").bfloat16()
scale = 512 ** -0.5
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for abl in os.environ.get("ABLS", "0,1,2").split(","):
    os.environ["GSL_ATTN_ABL"] = abl
    o, lse = ops.attention_fwd(qkv, B, T, H, scale)
    tf = t(lambda: ops.attention_fwd(qkv, B, T, H, scale))
    tb = t(lambda: ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale))
    fl = 4.0 * B * H * T * T * 64
    print(f"abl={abl}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF)   bwd(dq+dkv) {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.1f} TF)", flush=True)
