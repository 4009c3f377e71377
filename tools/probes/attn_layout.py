#!/usr/bin/env python
"""Is the attention bound by the 128-byte-segment access pattern of the token-major [B*T, 3*H*64] qkv layout? Same work, two layouts:
H = 8 heads interleaved per token row (3072-byte row stride) vs H = 1 (each item's q|k|v rows contiguous, 384-byte stride)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
T = 197
scale = 512 ** -0.5


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


for B, H in ((1024, 8), (8192, 1), (1024, 8), (8192, 1)):
    torch.manual_seed(0)
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda").bfloat16()
    d_o = torch.randn(B * T, H * 64, device="cuda").bfloat16()
    o, lse = ops.attention_fwd(qkv, B, T, H, scale)
    tf = t(lambda: ops.attention_fwd(qkv, B, T, H, scale))
    tb = t(lambda: ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale))
    print(f"B={B:5d} H={H}: fwd {tf:7.1f} us   bwd {tb:7.1f} us", flush=True)
