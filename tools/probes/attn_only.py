#!/usr/bin/env python
"""Launch the attention forward / backward (B = 1024, T = 197, H = 8) a few times: the subject of PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops
B, T, H = int(os.environ.get("B", 1024)), 197, 8
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").bfloat16()
d_o = torch.randn(B * T, H * 64, device="cuda").bfloat16()
scale = 512 ** -0.5
o, lse = ops.attention_fwd(qkv, B, T, H, scale)
for _ in range(4):
    ops.attention_fwd(qkv, B, T, H, scale)
    ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale)
torch.cuda.synchronize()
