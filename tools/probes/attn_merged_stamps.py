#!/usr/bin/env python
"""Cycle stamps of the merged attention backward (dev build, GSL_ATTN_STAMPS): third item of every 32nd workgroup, every wave:
item start | P1a | barrier | P2a | barrier | P1b pairs | P1b last pair | stores + next keys | barrier | P2b | barrier (waves 13..15 = loaders:
req dO,o | barrier | - | barrier | delta + lse | req q | - | barrier | deposit | barrier)."""
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
dbg = torch.zeros(2048, device="cuda", dtype=torch.int64)
os.environ["GSL_ATTN_STAMPS"] = hex(dbg.data_ptr())
for _ in range(3):
    dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale); b.record()
torch.cuda.synchronize()
st = dbg.cpu().view(8, 16, 16)[:, :, :11].double()      # [workgroup][wave][stamp]
d = st[:, :, 1:] - st[:, :, :-1]
print(f"kernel {a.elapsed_time(b) * 1e3:.0f} us; median over 8 workgroups, cycles per section")
print("wave   P1a    b1    P2a    b2   P1b-pairs  last   st+ld    b3    P2b    b4   | item")
for w in range(16):
    m = d[:, w].median(0).values
    print(f"{w:3d} " + " ".join(f"{v:6.0f}" for v in m.tolist()) + f"  | {(st[:, w, 10] - st[:, w, 0]).median():.0f}")
