import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
from gslora_hip import ops, _lib as L
torch.manual_seed(0)
M, D, N = 4096, 512, 1536
for dt in (torch.float16, torch.bfloat16):
    for mu_scale, outl in ((0.0, 0.0), (0.5, 0.0), (2.0, 0.0), (0.5, 8.0)):
        x32 = torch.randn(M, D) + mu_scale * torch.randn(M, 1)
        if outl: x32[:, 7] += outl; x32[:, 300] -= outl
        x = x32.to(dt).cuda()
        g = (1 + 0.2 * torch.randn(D)).cuda(); b = (0.1 * torch.randn(D)).cuda()
        W = (torch.randn(N, D) * D ** -0.5).cuda()
        xd = x.double(); mu = xd.mean(1, keepdim=True); var = ((xd - mu) ** 2).mean(1, keepdim=True)
        ref = (((xd - mu) / (var + 1e-5).sqrt()) * g.double() + b.double()) @ W.double().t()
        # unfused
        xn, mean, rstd = ops.layernorm_fwd(x, D, M, D, g, b, 1e-5, dt)
        y0 = torch.empty(M, N, device="cuda", dtype=dt); ops.gemm_nt(xn, W.to(dt), y0)
        # fused
        mean2, rstd2 = ops.layernorm_stats(x, D, M, D, g, b, 1e-5, dt)
        wf = (W * g[None, :]).to(dt); c = wf.float().sum(1).contiguous(); d = (W @ b).contiguous()
        y1 = torch.empty(M, N, device="cuda", dtype=dt); ops.gemm_nt(x, wf, y1, epilogue=L.EPI_STORE_LN, pos=mean2, cls=rstd2, aux=c, bias=d)
        # f32 output variants isolate the output rounding
        e0 = (y0.double() - ref); e1 = (y1.double() - ref)
        print(f"{str(dt):15s} mean {mu_scale} outlier {outl}: unfused rms {e0.pow(2).mean().sqrt():.3e} max {e0.abs().max():.3e} | folded rms {e1.pow(2).mean().sqrt():.3e} max {e1.abs().max():.3e} | stats equal {torch.equal(mean, mean2) and torch.equal(rstd, rstd2)}")
