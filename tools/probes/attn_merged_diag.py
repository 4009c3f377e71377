import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops, _lib as L
B, T, H = 70, 197, 8
scale = 64 ** -0.5
g = torch.Generator().manual_seed(1)
for dt in (torch.bfloat16, torch.float16):
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 1.3).cuda().to(dt)
    d_o = torch.randn(B * T, H * 64, generator=g).cuda().to(dt)
    o, lse = ops.attention_fwd(qkv, B, T, H, scale)
    m = ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale)
    os.environ["GSL_ATTN_BWD_MERGED"] = "0"
    with L.use_dev():
        f = ops.attention_bwd(qkv, o, d_o, lse, B, T, H, scale)
    os.environ.pop("GSL_ATTN_BWD_MERGED")
    m3 = m.view(B, T, 3, H, 64).float(); f3 = f.view(B, T, 3, H, 64).float()
    for i, n in enumerate("qkv"):
        d = (m3[:, :, i] - f3[:, :, i]).abs()
        nz = (d > 0)
        print(dt, "d" + n, "mismatches", int(nz.sum()), "of", d.numel(), "max", float(d.max()), "max|val|", float(f3[:, :, i].abs().max()))
        if nz.any():
            idx = nz.nonzero()[:5].tolist()
            print("   first:", idx, [ (float(m3[:, :, i][tuple(j)]), float(f3[:, :, i][tuple(j)])) for j in idx])
            print("   by token tile:", [int(nz[:, t*16:(t+1)*16].sum()) for t in range(13)])
