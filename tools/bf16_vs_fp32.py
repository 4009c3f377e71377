#!/usr/bin/env python
"""Speed-mode (bf16) vs parity-mode (fp32) of the HIP path on the FULL ViT-P8S8 model: logits / embedding / LoRA-gradient agreement.
Both modes run the same weights, batch and loss (dropout off); reports max-abs and relative Frobenius errors."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch
import loralib as lora
from vit_pytorch_face import ViT_face
from gslora_hip import losses
torch.manual_seed(0)
B = int(os.environ.get("B", 64))
m = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=100, image_size=112, patch_size=8, dim=512, depth=6, heads=8, mlp_dim=2048,
             dropout=0.0, emb_dropout=0.0, lora_rank=8)
with torch.no_grad():
    for n, p in m.named_parameters():
        if "lora_B" in n:
            p.normal_(0, 0.02)
lora.mark_only_lora_as_trainable(m)
m = m.cuda().train()
x = torch.rand(2 * B, 3, 112, 112, device="cuda"); y = torch.randint(0, 100, (2 * B,), device="cuda")
proto = torch.randn(100, 512, device="cuda")
res = {}
for mode in ("fp32", "bf16"):
    mm = copy.deepcopy(m).set_compute_dtype(mode)
    lo, em = mm(x, y)
    ce_r = losses.ce_sum_top1(lo[:B], y[:B])[0] / B
    ce_f = losses.ce_sum_top1(lo[B:], y[B:])[0] / B
    kl = losses.proto_kl_sum(em[:B], y[:B], proto) / B
    total = 0.15 * torch.relu(105.0 - ce_f) + ce_r + 1e-4 * losses.structure_loss(mm, "block") + 0.05 * kl
    total.backward()
    res[mode] = (lo.detach().float(), em.detach().float(), torch.cat([p.grad.reshape(-1) for p in mm.parameters() if p.requires_grad]), total.item())
a, b = res["fp32"], res["bf16"]
g32, g16 = a[2], b[2]
print(f"B={B}+{B}: logits max|d| {float((a[0]-b[0]).abs().max()):.4f} (scale 64), emb max|d| {float((a[1]-b[1]).abs().max()):.4f}, "
      f"loss fp32 {a[3]:.5f} bf16 {b[3]:.5f}, LoRA grad rel Frobenius err {float((g32-g16).norm()/g32.norm()):.4f}, "
      f"cosine {float(torch.dot(g32, g16)/(g32.norm()*g16.norm())):.6f}")
