#!/usr/bin/env python
"""What do the fragment-layout (8 bytes per lane) output stores of the fused attention backward cost? GSL_ATTN_ABL=4 skips them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
B, T, H = 1024, 197, 8
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").bfloat16()
d_o = torch.randn(B * T, H * 64, device="cuda").bfloat16()
scale = 512 ** -0.5
hm = qkv.view(B, T, 3, H, 64).permute(0, 3, 2, 1, 4).contiguous().view(B * T, 3 * H * 64)
o, lse = ops.attention_fwd(qkv, B, T, H, scale)


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


for abl in ("0", "4", "0", "4"):
    os.environ["GSL_ATTN_ABL"] = abl
    print(f"abl={abl}: bwd token-major {t(lambda: ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale)):7.1f} us   head-major input {t(lambda: ops.attention_bwd(hm, o, d_o, lse, B, T, H, scale, layout=1)):7.1f} us", flush=True)
