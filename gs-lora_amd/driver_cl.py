#!/usr/bin/env python
"""driver_cl — the build's own counterpart of the reference's continual-forgetting driver
(train/train_own_forget_cl.py: task loop :515-536, loaders :696-750, prototypes :633-646, optimizer/scheduler :807-820,
epochs :999-1106, LoRA-norm report :1100-1106, task checkpoint :1696-1705), issuing the SAME call sequence against the
HIP-backed modules on device-resident synthetic data (the reference's ImageFolder / wandb / argparse plumbing is out of scope).

    python gs-lora_amd/driver_cl.py --num_tasks 4 --epochs 2 --batch_size 48

Sequence per task i:
  i > 0: load task-level/Backbone_task_{i-1}.pth (saved in eval()==merged form) and reinitialize_lora_parameters
  split classes: forget = order[en1:en2], remain = order[:en1]   (st1=0, en1=num_first - i*per_forget, en2=en1+per_forget)
  prototypes = calculate_prototypes(model, forget U remain subset)      (eval mode, leaves the model in eval())
  criterion = CrossEntropyLoss; optimizer = create_optimizer(args, model); scheduler = create_scheduler(args, optimizer)
  forget_acc_before = eval_data(...)
  for epoch: scheduler.step(epoch); alpha = 0 if epoch < warmup_alpha else alpha; engine_cl.train_one_epoch(**30 kwargs)
  get_norm_of_lora(model); model.eval(); torch.save(state_dict) ; model.train()
"""
import argparse
import copy
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch  # noqa: E402
from torch.utils.data import DataLoader, TensorDataset  # noqa: E402

import engine_cl  # noqa: E402
import loralib as lora  # noqa: E402
from gslora_hip.optim import create_optimizer, create_scheduler  # noqa: E402
from util.cal_norm import get_norm_of_lora  # noqa: E402
from util.utils import AverageMeter, calculate_prototypes, count_trainable_parameters, reinitialize_lora_parameters  # noqa: E402
from vit_pytorch_face import ViT_face  # noqa: E402


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--num_class", type=int, default=100)
    p.add_argument("--num_tasks", type=int, default=4)
    p.add_argument("--per_forget_cls", type=int, default=20)
    p.add_argument("--epochs", type=int, default=2)
    p.add_argument("--batch_size", type=int, default=48)
    p.add_argument("--samples_per_class", type=int, default=8)
    p.add_argument("--lr", type=float, default=1e-2)
    p.add_argument("--min_lr", type=float, default=1e-5)
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--opt", default="adamw")
    p.add_argument("--opt_eps", type=float, default=1e-8)
    p.add_argument("--opt_betas", default=None)
    p.add_argument("--sched", default="cosine")
    p.add_argument("--warmup_epochs", type=int, default=0)
    p.add_argument("--warmup_lr", type=float, default=1e-6)
    p.add_argument("--cooldown_epochs", type=int, default=10)
    p.add_argument("--alpha", type=float, default=1e-4)
    p.add_argument("--warmup_alpha", type=int, default=0, help="epochs with alpha = 0 (reference :1007-1011)")
    p.add_argument("--beta", type=float, default=0.15)
    p.add_argument("--BND", type=float, default=105.0)
    p.add_argument("--BND_pro", type=float, default=18.0)
    p.add_argument("--pro_f_weight", type=float, default=0.01)
    p.add_argument("--pro_r_weight", type=float, default=0.01)
    p.add_argument("--lora_rank", type=int, default=8)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--dtype", default="bf16")
    p.add_argument("--small", action="store_true", help="shrunken model (48 px, dim 128, depth 3) for tests")
    p.add_argument("--outdir", default=None)
    p.add_argument("--seed", type=int, default=1337)
    p.add_argument("--average_weight", action="store_true", help="EMA model of the reference (:502-507, :1058-1098)")
    p.add_argument("--ema_epoch", type=int, default=30)
    p.add_argument("--ema_decay", type=float, default=0.9)
    return p.parse_args(argv)


def synthetic_dataset(num_class, per_class, image_size, seed):
    """Class-conditional synthetic faces: a per-class low-frequency pattern plus noise, u8/255 like ToTensor()."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(num_class, 3, 7, 7, generator=g)
    base = torch.nn.functional.interpolate(base, size=image_size, mode="bilinear", align_corners=False)
    x = base.repeat_interleave(per_class, 0) * 0.7 + 0.3 * torch.rand(num_class * per_class, 3, image_size, image_size, generator=g)
    y = torch.arange(num_class).repeat_interleave(per_class)
    return (x * 255).round().clamp(0, 255) / 255.0, y


def subset(x, y, classes):
    m = torch.isin(y, torch.tensor(classes))
    return TensorDataset(x[m], y[m])


def main(argv=None):
    args = get_args(argv)
    torch.manual_seed(args.seed)
    dev = torch.device("cuda")
    geo = (dict(image_size=48, patch_size=8, dim=128, depth=3, heads=2, mlp_dim=256) if args.small else
           dict(image_size=112, patch_size=8, dim=512, depth=6, heads=8, mlp_dim=2048))
    out = args.outdir or tempfile.mkdtemp(prefix="gslora_cl_")
    os.makedirs(os.path.join(out, "task-level"), exist_ok=True)
    cfg = {"DATA_ROOT": "./data/synthetic/", "BND_pro": args.BND_pro, "MULTI_GPU": False, "WORK_PATH": out, "BACKBONE_NAME": "VIT"}

    order = list(range(args.num_class))                       # reference :198-204
    random.seed(args.seed)
    random.shuffle(order)
    model = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=args.num_class, dropout=args.dropout, emb_dropout=args.dropout,
                     lora_rank=args.lora_rank, **geo)
    lora.mark_only_lora_as_trainable(model)                   # :314-317
    print("trainable parameters:", count_trainable_parameters(model))
    model = model.to(dev).set_compute_dtype(args.dtype)
    x_all, y_all = synthetic_dataset(args.num_class, args.samples_per_class, geo["image_size"], args.seed)
    x_te, y_te = synthetic_dataset(args.num_class, 2, geo["image_size"], args.seed + 1)
    num_first = args.num_class - args.per_forget_cls           # classes not yet forgotten after task 0
    ema_model = None
    if args.average_weight:                                   # :502-507: a deep copy taken in eval() — its adapters are flagged MERGED
        model.eval()
        ema_model = copy.deepcopy(model).to(dev)
        model.train()
    report = []
    for task_i in range(args.num_tasks):                      # :515
        if task_i > 0:                                        # :524-536
            sd = torch.load(os.path.join(out, "task-level", f"Backbone_task_{task_i - 1}.pth"), map_location="cpu")
            model.load_state_dict(sd)
            reinitialize_lora_parameters(model)
        en1 = num_first - task_i * args.per_forget_cls + 0     # :539-545  (st1 = 0, st2 = en1)
        en2 = en1 + args.per_forget_cls
        remain_cls, forget_cls = order[:en1], order[en1:en2]
        gen = torch.Generator().manual_seed(args.seed + task_i)
        mk = lambda ds, bs, sh: DataLoader(ds, batch_size=bs, shuffle=sh, generator=gen if sh else None, drop_last=False)
        tr_f, tr_r = subset(x_all, y_all, forget_cls), subset(x_all, y_all, remain_cls)
        loader_f, loader_r = mk(tr_f, args.batch_size, True), mk(tr_r, args.batch_size, True)
        te_f, te_r = mk(subset(x_te, y_te, forget_cls), 5 * args.batch_size, False), mk(subset(x_te, y_te, remain_cls), 5 * args.batch_size, False)
        protos = calculate_prototypes(model, subset(x_all, y_all, forget_cls + remain_cls), batch_size=500, device=dev)   # :633-646
        model.train()
        criterion = torch.nn.CrossEntropyLoss()
        optimizer = create_optimizer(args, model)             # :811
        scheduler, _ = create_scheduler(args, optimizer)      # :818
        forget_before = engine_cl.eval_data(model, te_f, dev, f"forget-{task_i}-before")
        remain_before = engine_cl.eval_data(model, te_r, dev, f"remain-{task_i}-before")
        model.train()
        batch, best_h, lrs, ema_acc = 0, 0.0, [], None
        for epoch in range(args.epochs):                      # :1006
            scheduler.step(epoch)
            lrs.append(optimizer.param_groups[0]["lr"])
            alpha = 0.0 if epoch < args.warmup_alpha else args.alpha
            m = {k: AverageMeter() for k in ("losses_forget", "losses_remain", "losses_total", "losses_structure", "top1_forget",
                                             "top1_remain", "losses_prototype_forget", "losses_prototype_remain")}
            ret = engine_cl.train_one_epoch(
                model=model, dataloader_forget=loader_f, dataloader_remain=loader_r, device=dev, criterion=criterion,
                optimizer=optimizer, epoch=epoch, beta=args.beta, alpha=alpha, BND=args.BND, batch=batch, testloader_forget=te_f,
                testloader_remain=te_r, forget_acc_before=forget_before, highest_H_mean=best_h, cfg=cfg, task_i=str(task_i),
                use_prototype=True, prototype_dict=protos, prototype_weight_forget=args.pro_f_weight,
                prototype_weight_remain=args.pro_r_weight, **m)
            batch, best_h = ret[0], ret[1]
            if ema_model is not None:                         # :1058-1098. Reproduced as written: the TRAIN-mode (un-merged) parameters
                with torch.no_grad():                         # are copied / averaged into a model whose adapters stay flagged merged.
                    snap = copy.deepcopy(model)
                    ema_model.eval()
                    if epoch == args.ema_epoch:
                        for p_, e_ in zip(snap.parameters(), ema_model.parameters()):
                            e_.data = p_.data.detach()
                    elif epoch > args.ema_epoch:
                        for p_, e_ in zip(snap.parameters(), ema_model.parameters()):
                            e_.data = e_.data.detach() * args.ema_decay + p_.data.detach() * (1 - args.ema_decay)
                    if epoch >= args.ema_epoch:
                        ema_acc = (engine_cl.eval_data(ema_model, te_f, dev, f"forget-ema-{task_i}", batch),
                                   engine_cl.eval_data(ema_model, te_r, dev, f"remain-ema-{task_i}", batch))
                model.train()
        norms = [float(v) for v in get_norm_of_lora(model, type="L2", group_num=geo["depth"], group_type="block")]   # :1100-1106
        forget_after = engine_cl.eval_data(model, te_f, dev, f"forget-{task_i}-after")
        remain_after = engine_cl.eval_data(model, te_r, dev, f"remain-{task_i}-after")
        model.eval()                                          # :1696-1705: checkpoints hold MERGED weights
        torch.save(model.state_dict(), os.path.join(out, "task-level", f"Backbone_task_{task_i}.pth"))
        model.train()
        report.append(dict(task=task_i, forget_cls=forget_cls, steps=batch, lrs=lrs, norms=norms, total_loss=ret[6].avg if ret[6].count else None,
                           forget_before=forget_before, forget_after=forget_after, remain_before=remain_before, remain_after=remain_after,
                           ema_acc=ema_acc))
        print(f"[task {task_i}] steps={batch} lr={lrs} norms={[round(v, 3) for v in norms]} "
              f"forget {forget_before:.1f}->{forget_after:.1f}  remain {remain_before:.1f}->{remain_after:.1f}")
    return report, out, (model if ema_model is None else (model, ema_model))


if __name__ == "__main__":
    main()
