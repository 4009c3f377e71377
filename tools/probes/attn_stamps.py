#!/usr/bin/env python
"""Cycle stamps of the fused attention backward (GSL_ATTN_STAMPS): per workgroup start -> K/V staged -> wave 0 done with phase A ->
barrier -> Q/dO panels staged -> wave 0 round 1 of phase B -> round 2 (wave 0 owns key tiles 0 and 8)."""
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
st = dbg.cpu().view(-1, 8)[:, :7]
st = st[(st != 0).all(1)]
d = (st[:, 1:] - st[:, :-1]).double()
names = ["stage K/V", "phase A (wave 0: 2 tiles)", "barrier wait", "restage Q/dO", "phase B round 1", "phase B round 2"]
print(f"{st.shape[0]} workgroups, kernel {a.elapsed_time(b) * 1e3:.0f} us; median cycles per section:")
for i, n in enumerate(names):
    print(f"  {n:28s} {d[:, i].median():8.0f}   (mean {d[:, i].mean():8.0f})")
tot = (st[:, 6] - st[:, 0]).double()
print(f"  {'total per item':28s} {tot.median():8.0f}; kernel span {(st[:, 6].max() - st[:, 0].min())} cycles")
