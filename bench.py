#!/usr/bin/env python
"""bench.py — forgetting-step images/sec of the GS-LoRA step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (config.workload): BASELINE.json configs[1] — ViT-P8S8 depth 6, 112 px, LoRA r=8 on both
FFN linears, CosFace-100 head, per-GPU batch 512 remain + 512 forget images resident in HBM,
bf16 speed mode, dropout 0.1 / emb-dropout 0.1 (the reference's training setting), prototype term on,
FusedAdamW (lr 1e-2, wd 0.05). One "step" = the engine_cl.train_one_epoch loop body
(2 forwards, 5 loss terms, backward, gradient all-reduce when N>1, AdamW). Weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gs-lora_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E, same guide
FULL = dict(image_size=112, patch_size=8, dim=512, depth=6, heads=8, mlp_dim=2048, num_class=100, lora_rank=8)
HYPER = dict(lr=1e-2, wd=0.05, beta=0.15, alpha=1e-4, BND=105.0, BND_pro=18.0, pro_f=0.01, pro_r=0.01)


def build_model(dtype, dropout, device):
    import loralib as lora
    from vit_pytorch_face import ViT_face
    torch.manual_seed(1337)
    m = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=FULL["num_class"], image_size=FULL["image_size"],
                 patch_size=FULL["patch_size"], dim=FULL["dim"], depth=FULL["depth"], heads=FULL["heads"],
                 mlp_dim=FULL["mlp_dim"], dropout=dropout, emb_dropout=dropout, lora_rank=FULL["lora_rank"])
    lora.mark_only_lora_as_trainable(m)
    with torch.no_grad():   # non-trivial adapters (B != 0) so the LoRA paths do real work
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.normal_(0.0, 0.02)
    return m.to(device).set_compute_dtype(dtype).train()


def cpu_baseline(batch=16, steps=2):
    """The CPU oracle (a port of the reference step; the reference's Python cannot travel to the GPU
    box) timed on the host cores: fp32, B=16+16, full-size model, dropout omitted (the reference spends
    ~26 % of its CPU time in bernoulli_, so this baseline is FASTER than the reference itself)."""
    from oracle import gslora_oracle as O
    from oracle import recipe
    cfg = recipe.cfg_full()
    cores = min(32, os.cpu_count() or 1)     # more threads than this only adds fork/join overhead at B=16
    torch.set_num_threads(cores)
    st = recipe.make_state(cfg)
    xr = torch.tensor(recipe.make_images(cfg, batch, seed=1)); yr = torch.tensor(recipe.make_labels(cfg, batch, seed=1, hi=80))
    xf = torch.tensor(recipe.make_images(cfg, batch, seed=2)); yf = torch.tensor(recipe.make_labels(cfg, batch, seed=2, lo=80))
    proto = torch.tensor(recipe.make_prototypes(cfg))
    hy = dict(beta=HYPER["beta"], alpha=HYPER["alpha"], BND=HYPER["BND"], BND_pro=HYPER["BND_pro"], pro_f_weight=HYPER["pro_f"],
              pro_r_weight=HYPER["pro_r"], wd=HYPER["wd"])
    opt = None
    times = []
    for s in range(steps + 1):
        t0 = time.perf_counter()
        _, _, new_st, opt = O.train_step(st, cfg, xr, yr, xf, yf, hy, opt_state=opt, step=s + 1, lr=HYPER["lr"], proto=proto)
        st = {k: v.numpy() for k, v in new_st.items()}
        if s:
            times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    return {"value": round(2 * batch / t, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle train_step (fp32 torch CPU, no dropout), ViT-P8S8 d6 r8, B={batch}+{batch}, median of {steps} steps "
                      f"after 1 warm-up, {t:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU images per forward (remain and forget each)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as a captured HIP graph (the engines' default for launch-bound batches)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the GS-LoRA step has no CPU fallback)")
    # development knobs for exercising the world > 1 code path on a 1-GPU box: GSL_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # GSL_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device). Never set by the driver.
    if os.environ.get("GSL_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GSL_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from gslora_hip import ops
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import gs_lora_step
    model = build_model(args.dtype, args.dropout, dev)
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=HYPER["lr"], weight_decay=HYPER["wd"], eps=1e-8)
    crit = torch.nn.CrossEntropyLoss()
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1337 + rank)
    import random
    order = list(range(100)); random.seed(1337); random.shuffle(order)
    mk_img = lambda: (torch.randint(0, 256, (B, 3, 112, 112), generator=g, dtype=torch.uint8).float() / 255.0).to(dev)
    x_r, x_f = mk_img(), mk_img()
    y_r = torch.tensor(order[:80])[torch.randint(0, 80, (B,), generator=g)].to(dev)
    y_f = torch.tensor(order[80:])[torch.randint(0, 20, (B,), generator=g)].to(dev)
    proto = torch.randn(100, FULL["dim"], generator=g).to(dev)

    from gslora_hip.step import GraphedStep
    stepper = GraphedStep(model, opt, crit) if (args.graph and world == 1) else (lambda *a, **k: gs_lora_step(model, opt, crit, *a, **k))

    def step():
        return stepper(x_r, y_r, x_f, y_f, beta=HYPER["beta"], alpha=HYPER["alpha"], BND=HYPER["BND"],
                            use_structure=True, group_type="block", use_prototype=True, proto_table=proto, w_f=HYPER["pro_f"],
                            w_r=HYPER["pro_r"], BND_pro=HYPER["BND_pro"])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ops.PROFILE = {"ffn1": []}
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step HIP events (no sync inside the timed region)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        pack = step()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    meters = pack.tolist()

    if rank == 0:
        r = FULL["lora_rank"]
        durs = [a.elapsed_time(b) for a, b, *_ in prof["ffn1"]]       # ms per launch of the fused FFN1+LoRA+GELU GEMM
        avg_ms = sum(durs) / max(1, len(durs))
        # algorithmic flops of one launch: 2*M*N*K for the dense part + 2*M*N*r for the LoRA up-projection K segment
        fl = [2.0 * m_ * n_ * k1 + 2.0 * m_ * n_ * r for _, _, m_, n_, k1, _ in prof["ffn1"]]
        flops = sum(fl) / max(1, len(fl))
        M = prof["ffn1"][0][2] if prof["ffn1"] else 2 * B * 197
        ach = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        alg_bytes = int(M * (FULL["dim"] + 64) * 2 + 2 * M * FULL["mlp_dim"] * 2)     # A + LoRA segment read, h + GELU' written (bf16)
        traffic = None
        try:   # HBM bytes per launch of the roofline kernel, from the committed PMC passes (rocprofv3 cannot wrap itself)
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pj.get("rows_per_launch") == M:
                traffic = pj["hbm_bytes_per_launch"]
        except Exception:
            pass
        ips = world * 2 * B * args.steps / elapsed
        out = {
            "metric": "forgetting-step images/sec, ViT-P8S8 d6 112px r=8", "value": round(ips, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ViT-P8S8 depth-6 CASIA-100-shaped single-task forget step, LoRA r=8, per-GPU batch {B} remain + "
                                   f"{B} forget (112x112 synthetic), dropout {args.dropout}, prototype term on, FusedAdamW",
                       "global_batch": world * 2 * B, "tokens_per_image": 197, "parallelism": f"dp{world}"},
            # The fused FFN1 GEMM writes TWO [M, 2048] bf16 outputs (h and GELU'): 430 GFLOP over 1.885 GB of algorithmic bytes is
            # 228 FLOP/B, below the MI355X ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B) -> HBM is the roof that bounds it.
            "roofline": {"bound": "hbm", "kernel": "gsl_gemm_nt<BIAS_GELU> (fused FFN1 + LoRA-up K-segment + bias + GELU + GELU' + dropout, fwd)",
                         "achieved": round(alg_bytes / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(alg_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if avg_ms > 0 else 0.0,
                         "launches_timed": len(durs), "rows_per_launch": M, "avg_ms": round(avg_ms, 4), "traffic": traffic,
                         "algorithmic_bytes": alg_bytes, "arithmetic_intensity_flop_per_byte": round(flops / alg_bytes, 1),
                         "mfma_tflops": round(ach, 2), "mfma_frac_of_peak": round(ach / PEAK_BF16_TFLOPS, 4)},
            "step_flops_frac_of_peak": round((15.646e9 * 2 * B * args.steps / elapsed) / (PEAK_BF16_TFLOPS * 1e12), 4),
            "last_step_meters": {"beta*loss_forget": meters[0], "loss_remain": meters[1], "total": meters[2]},
            "hip_graph": bool(args.graph and world == 1),
            "ms_per_step_events": {"median": round(per_step[len(per_step) // 2], 3), "p10": round(per_step[int(0.1 * (len(per_step) - 1))], 3),
                                   "p90": round(per_step[int(round(0.9 * (len(per_step) - 1)))], 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
