#!/usr/bin/env python
"""Which bf16 operand of the speed mode costs how much accuracy (VERDICT r03 next #1b): FULL ViT-P8S8, batch B+B, the step's loss; every
configuration against the path's own f32 parity mode on the same weights / batch (dropout off): logits, embedding, LoRA-gradient
relative Frobenius error and cosine, per adapter matrix kind. In-process: the knobs are module attributes of gslora_hip.vit_runner.
Usage (GPU box): B=64 python tools/precision_ablation.py > gpurun_out/precision_ablation.md"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch  # noqa: E402
import loralib as lora  # noqa: E402
from gslora_hip import losses, vit_runner as R  # noqa: E402
from vit_pytorch_face import ViT_face  # noqa: E402

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
x = torch.rand(2 * B, 3, 112, 112, device="cuda")
y = torch.randint(0, 100, (2 * B,), device="cuda")
proto = torch.randn(100, 512, device="cuda")


def run(mode, attrs):
    saved = {k: getattr(R, k) for k in attrs}
    for k, v in attrs.items():
        setattr(R, k, v)
    try:
        mm = copy.deepcopy(m).set_compute_dtype(mode)
        lo, em = mm(x, y)
        ce_r = losses.ce_sum_top1(lo[:B], y[:B])[0] / B
        ce_f = losses.ce_sum_top1(lo[B:], y[B:])[0] / B
        kl = losses.proto_kl_sum(em[:B], y[:B], proto) / B
        total = 0.15 * torch.relu(105.0 - ce_f) + ce_r + 1e-4 * losses.structure_loss(mm, "block") + 0.05 * kl
        total.backward()
        grads = {n: p.grad.detach().clone().reshape(-1) for n, p in mm.named_parameters() if p.requires_grad}
        return lo.detach().float(), em.detach().float(), grads, total.item()
    finally:
        for k, v in saved.items():
            setattr(R, k, v)


ref = run("fp32", {})
CONFIGS = [("fp16 operands (round 5 default: fp16 streams, loss-scaled backward, 8-bit GELU')", {"_mode": "fp16"}),
           ("fp16 operands, fp16 GELU' instead of the 8-bit code", {"_mode": "fp16", "GP8": False}),
           ("fp16 operands, f32 forward + gradient streams", {"_mode": "fp16", "FWD_STREAM": "f32", "GRAD_STREAM_BF16": False}),
           ("fp16 operands, all wide (f32 streams, fp16 GELU')", {"_mode": "fp16", "FWD_STREAM": "f32", "GRAD_STREAM_BF16": False, "GP8": False}),
           ("bf16 operands (round 4 default: fp16 forward stream, bf16 gradient stream, 8-bit GELU')", {}),
           ("bf16 forward residual stream (round 3)", {"FWD_STREAM": "bf16"}),
           ("f32 forward residual stream", {"FWD_STREAM": "f32"}),
           ("f32 gradient residual stream", {"GRAD_STREAM_BF16": False}),
           ("bf16 GELU' instead of the 8-bit code", {"GP8": False}),
           ("all three wide (round-2 precision)", {"FWD_STREAM": "f32", "GRAD_STREAM_BF16": False, "GP8": False}),
           ("dense last block (TAIL_CLS off)", {"TAIL_CLS": False})]
print(f"FULL ViT-P8S8, batch {B}+{B}, against the f32 parity mode of the same path (loss {ref[3]:.5f})\n")
print("| configuration | logits max abs (scale 64) | emb max abs | loss | LoRA grad rel. Frobenius | cosine | worst tensor rel. | dA1 / dB1 / dA2 / dB2 rel. |")
print("|---|---|---|---|---|---|---|---|")
for name, attrs in CONFIGS:
    attrs = dict(attrs)
    lo, em, g, tot = run(attrs.pop("_mode", "bf16"), attrs)
    g32 = torch.cat([ref[2][k] for k in g])
    g16 = torch.cat([g[k] for k in g])
    worst = max(float((g[k] - ref[2][k]).norm() / ref[2][k].norm()) for k in g if ref[2][k].norm() > 0)
    kinds = []
    for suffix in ("net.0.lora_A", "net.0.lora_B", "net.3.lora_A", "net.3.lora_B"):
        a = torch.cat([g[k] for k in g if k.endswith(suffix)])
        b = torch.cat([ref[2][k] for k in g if k.endswith(suffix)])
        kinds.append(f"{float((a - b).norm() / b.norm()):.4f}")
    print(f"| {name} | {float((lo - ref[0]).abs().max()):.4f} | {float((em - ref[1]).abs().max()):.4f} | {tot:.5f} | "
          f"{float((g32 - g16).norm() / g32.norm()):.5f} | {float(torch.dot(g32, g16) / (g32.norm() * g16.norm())):.6f} | {worst:.4f} | {' / '.join(kinds)} |", flush=True)
