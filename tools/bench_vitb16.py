#!/usr/bin/env python
"""ViT-B/16 (ModifiedViT, LoRA r=16 on the FFN linears, 100-way head) forgetting step on one MI355X — BASELINE config 3 shape.
Reference setting: per-GPU batch 48 remain + 48 forget (scripts/run_cl_forget_image.sh: -b 48), dropout 0 (torchvision default).
  python tools/bench_vitb16.py [--batch 48] [--steps 10] [--warmup 3] [--dtype bf16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-lora_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--graph", action="store_true", help="replay the step as a captured HIP graph")
    a = ap.parse_args()
    import loralib as lora
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import gs_lora_step
    from util.utils import replace_ffn_with_lora
    from vit_pytorch_face import ModifiedViT
    from vit_pytorch_face.modified_VIT import vit_b_16
    torch.manual_seed(0)
    m = replace_ffn_with_lora(ModifiedViT(vit_b_16(num_classes=100)), rank=a.rank)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("lora_B"):
                p.normal_(0, 0.02)
        m.heads.head.weight.normal_(0, 0.02)
    lora.mark_only_lora_as_trainable(m)
    m = m.to("cuda").set_compute_dtype(a.dtype).train()
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    B = a.batch
    x = torch.rand(2 * B, 3, 224, 224, device="cuda")
    y = torch.cat([torch.randint(0, 80, (B,)), torch.randint(80, 100, (B,))]).cuda()
    proto = torch.randn(100, 768, device="cuda")
    crit = torch.nn.CrossEntropyLoss()
    from gslora_hip.step import GraphedStep
    stepper = GraphedStep(m, opt, crit) if a.graph else (lambda *t, **k: gs_lora_step(m, opt, crit, *t, **k))
    step = lambda: stepper(x[:B], y[:B], x[B:], y[B:], beta=0.15, alpha=1e-4, BND=105.0, use_prototype=True,
                                proto_table=proto, w_f=0.05, w_r=0.05, BND_pro=18.0)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pack = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # model flops per image (fwd 1x + bwd: dX for 11 of 12 blocks' dense GEMMs, no dense dW): reported for orientation only
    T, D, F, L = 197, 768, 3072, 12
    fwd = L * (2 * T * D * 3 * D + 2 * T * D * D + 4 * T * D * F + 4 * T * T * D) + 2 * T * D * 768
    print(json.dumps({"workload": f"ViT-B/16 224px r={a.rank} forget step, batch {B}+{B}, {a.dtype}" + (", HIP graph" if a.graph else ""), "images_per_s": round(2 * B * a.steps / el, 2),
                      "ms_per_step": round(1e3 * el / a.steps, 3), "fwd_gflop_per_image": round(fwd / 1e9, 2),
                      "meters": [round(v, 4) for v in pack.tolist()]}))


if __name__ == "__main__":
    main()
