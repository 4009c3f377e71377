#!/usr/bin/env python
"""driver_cl — the build's own counterpart of the reference's continual-forgetting driver
(train/train_own_forget_cl.py: task loop :515-536, loaders :696-750, prototypes :633-646, optimizer/scheduler :807-820,
per-task hyper-parameters and epochs :999-1106, EMA model :502-507 / :1058-1098, LoRA-norm report :1100-1106, task checkpoint
:1696-1705), issuing the SAME call sequence against the HIP-backed modules (the reference's ImageFolder / wandb plumbing is out of scope).

    python gs-lora_amd/driver_cl.py --num_tasks 4 --epochs 2 --batch_size 48 --cl_beta_list 0.2 0.25 0.25 0.2 \
           --cl_prof_list 0.01 0.01 0.01 0.01 --average_weight --ema_epoch 30 --ema_decay 0.9      (scripts/run_cl_forget.sh:208-218)

Sequence per task i (run_tasks):
  i > 0: load task-level/Backbone_task_{i-1}.pth (saved in eval()==merged form) and reinitialize_lora_parameters
  split classes: forget = order[en1:en2], remain = order[:en1]   (st1=0, en1=num_first - i*per_forget, en2=en1+per_forget)
  prototypes = calculate_prototypes(model, forget U remain subset)      (eval mode, leaves the model in eval())
  criterion = CrossEntropyLoss; optimizer = create_optimizer(args, model); scheduler = create_scheduler(args, optimizer)
  forget_acc_before = eval_data(...)
  cl_beta = args.cl_beta_list[i]; pro_f_weight = args.cl_prof_list[i] if the list is given                               (:999-1002)
  for epoch: alpha = (0 if epoch < alpha_epoch else big_alpha) if warmup_alpha else args.alpha                            (:1007-1011)
             scheduler.step(epoch); engine_cl.train_one_epoch(**30 kwargs); EMA update / eval                            (:1013-1098)
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

METERS = ("losses_forget", "losses_remain", "losses_total", "losses_structure", "top1_forget", "top1_remain", "losses_prototype_forget",
          "losses_prototype_remain")


def get_args(argv=None):
    """The subset of the reference's util/args.py the continual driver reads, same names / types / defaults."""
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
    # util/args.py:366-376 — a FLAG; with it the structure weight is 0 before alpha_epoch and big_alpha from then on (:1007-1011)
    p.add_argument("--warmup_alpha", default=False, action="store_true")
    p.add_argument("--big_alpha", type=float, default=1e-4)
    p.add_argument("--alpha_epoch", type=int, default=20)
    p.add_argument("--beta", type=float, default=0.15, help="used for every task when --cl_beta_list is not given")
    p.add_argument("--cl_beta_list", nargs="*", default=[], type=float, help="per-task forget-loss weight (util/args.py:298; :1000)")
    p.add_argument("--BND", type=float, default=105.0)
    p.add_argument("--BND_pro", type=float, default=18.0)
    p.add_argument("--pro_f_weight", type=float, default=0.01)
    p.add_argument("--cl_prof_list", nargs="*", default=[], type=float, help="per-task pro_f_weight (util/args.py:347; :1001-1002)")
    p.add_argument("--pro_r_weight", type=float, default=0.01)
    p.add_argument("--lora_rank", type=int, default=8)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--dtype", default=os.environ.get("GSLORA_DTYPE", "fp16"), help="fp16 | bf16 (16-bit MFMA operands) | fp32 (parity mode)")
    p.add_argument("--small", action="store_true", help="shrunken model (48 px, dim 128, depth 3) for tests")
    p.add_argument("--outdir", default=None)
    p.add_argument("--seed", type=int, default=1337)
    p.add_argument("--average_weight", default=False, action="store_true", help="EMA model of the reference (:502-507, :1058-1098)")
    p.add_argument("--ema_epoch", type=int, default=50)
    p.add_argument("--ema_decay", type=float, default=0.99)
    return p.parse_args(argv)


def task_hyper(args, task_i, epoch):
    """(cl_beta, pro_f_weight, alpha) of one (task, epoch), as the reference derives them (train_own_forget_cl.py:999-1011). The
    reference indexes cl_beta_list unconditionally; here an absent list means `--beta` for every task."""
    if args.cl_beta_list and task_i >= len(args.cl_beta_list):
        raise IndexError(f"--cl_beta_list has {len(args.cl_beta_list)} entries, task {task_i} needs one (train_own_forget_cl.py:1000)")
    if args.cl_prof_list and task_i >= len(args.cl_prof_list):
        raise IndexError(f"--cl_prof_list has {len(args.cl_prof_list)} entries, task {task_i} needs one (train_own_forget_cl.py:1002)")
    beta = args.cl_beta_list[task_i] if args.cl_beta_list else args.beta
    pro_f = args.cl_prof_list[task_i] if len(args.cl_prof_list) != 0 else args.pro_f_weight
    alpha = (0.0 if epoch < args.alpha_epoch else args.big_alpha) if args.warmup_alpha else args.alpha
    return beta, pro_f, alpha


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


def run_tasks(model, args, task_data, dev, out, depth, cfg=None, after_reinit=None, after_task=None):
    """The reference's task loop on a model that is already LoRA-marked and on the device.
    task_data(task_i, model) -> dict(loader_f, loader_r, te_f, te_r, protos): the task's train / test loaders and prototype dict (called
    after the reload + re-initialisation of the task, so it may run calculate_prototypes on the model).
    after_reinit(model, task_i): hook behind reinitialize_lora_parameters (tests install seeded adapter matrices: kaiming_uniform_ draws
    from the device RNG). after_task(task_i, model, ema_model, record): hook behind the task's checkpoint (model in train mode)."""
    os.makedirs(os.path.join(out, "task-level"), exist_ok=True)
    cfg = cfg or {"DATA_ROOT": "./data/synthetic/", "BND_pro": args.BND_pro, "MULTI_GPU": False, "WORK_PATH": out, "BACKBONE_NAME": "VIT"}
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
            if after_reinit is not None:
                after_reinit(model, task_i)
        td = task_data(task_i, model)
        loader_f, loader_r, te_f, te_r, protos = td["loader_f"], td["loader_r"], td["te_f"], td["te_r"], td["protos"]
        model.train()
        criterion = torch.nn.CrossEntropyLoss()
        optimizer = create_optimizer(args, model)             # :811
        scheduler, _ = create_scheduler(args, optimizer)      # :818
        forget_before = engine_cl.eval_data(model, te_f, dev, f"forget-{task_i}-before")
        remain_before = engine_cl.eval_data(model, te_r, dev, f"remain-{task_i}-before")
        model.train()
        batch, best_h, lrs, hypers, ema_accs = 0, 0.0, [], [], []
        m = {k: AverageMeter() for k in METERS}               # :950-963: created once per task, re-bound by the engine's return value
        ret = None
        for epoch in range(args.epochs):                      # :1006
            cl_beta, pro_f, alpha = task_hyper(args, task_i, epoch)
            scheduler.step(epoch)                             # :1013
            lrs.append(optimizer.param_groups[0]["lr"])
            hypers.append((cl_beta, pro_f, alpha))
            ret = engine_cl.train_one_epoch(
                model=model, dataloader_forget=loader_f, dataloader_remain=loader_r, device=dev, criterion=criterion,
                optimizer=optimizer, epoch=epoch, beta=cl_beta, alpha=alpha, BND=args.BND, batch=batch, testloader_forget=te_f,
                testloader_remain=te_r, forget_acc_before=forget_before, highest_H_mean=best_h, cfg=cfg, task_i=task_i,
                use_prototype=True, prototype_dict=protos, prototype_weight_forget=pro_f,
                prototype_weight_remain=args.pro_r_weight, **m)
            batch, best_h = ret[0], ret[1]
            m = dict(losses_forget=ret[2], losses_remain=ret[3], top1_forget=ret[4], top1_remain=ret[5], losses_total=ret[6],
                     losses_structure=ret[7], losses_prototype_forget=ret[8], losses_prototype_remain=ret[9])
            if ema_model is not None:                         # :1058-1098. Reproduced as written: the TRAIN-mode (un-merged) parameters
                with torch.no_grad():                         # are copied / averaged into a model whose adapters stay flagged merged.
                    if epoch == args.ema_epoch:
                        snap = copy.deepcopy(model)
                        ema_model.eval()
                        for p_, e_ in zip(snap.parameters(), ema_model.parameters()):
                            e_.data = p_.data.detach()
                    elif epoch > args.ema_epoch:
                        snap = copy.deepcopy(model)
                        ema_model.eval()
                        for p_, e_ in zip(snap.parameters(), ema_model.parameters()):
                            e_.data = e_.data.detach() * args.ema_decay + p_.data.detach() * (1 - args.ema_decay)
                    if epoch >= args.ema_epoch:
                        ema_accs.append((engine_cl.eval_data(ema_model, te_f, dev, f"forget-ema-{task_i}", batch),
                                         engine_cl.eval_data(ema_model, te_r, dev, f"remain-ema-{task_i}", batch)))
                model.train()
        norms = [float(v) for v in get_norm_of_lora(model, type="L2", group_num=depth, group_type="block")]   # :1100-1106
        forget_after = engine_cl.eval_data(model, te_f, dev, f"forget-{task_i}-after")
        remain_after = engine_cl.eval_data(model, te_r, dev, f"remain-{task_i}-after")
        model.eval()                                          # :1696-1705: checkpoints hold MERGED weights
        torch.save(model.state_dict(), os.path.join(out, "task-level", f"Backbone_task_{task_i}.pth"))
        model.train()
        rec = dict(task=task_i, steps=batch, lrs=lrs, hypers=hypers, norms=norms,
                   total_loss=ret[6].avg if (ret is not None and ret[6].count) else None,
                   forget_before=forget_before, forget_after=forget_after, remain_before=remain_before, remain_after=remain_after,
                   ema_acc=ema_accs[-1] if ema_accs else None, ema_accs=ema_accs, **td.get("info", {}))
        if after_task is not None:
            after_task(task_i, model, ema_model, rec)
        report.append(rec)
        print(f"[task {task_i}] steps={batch} lr={lrs} (beta, pro_f, alpha)={hypers} norms={[round(v, 3) for v in norms]} "
              f"forget {forget_before:.1f}->{forget_after:.1f}  remain {remain_before:.1f}->{remain_after:.1f}")
    return report, ema_model


def main(argv=None):
    args = get_args(argv)
    torch.manual_seed(args.seed)
    dev = torch.device("cuda")
    geo = (dict(image_size=48, patch_size=8, dim=128, depth=3, heads=2, mlp_dim=256) if args.small else
           dict(image_size=112, patch_size=8, dim=512, depth=6, heads=8, mlp_dim=2048))
    out = args.outdir or tempfile.mkdtemp(prefix="gslora_cl_")
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

    def task_data(task_i, model):
        en1 = num_first - task_i * args.per_forget_cls        # :539-545  (st1 = 0, st2 = en1)
        en2 = en1 + args.per_forget_cls
        remain_cls, forget_cls = order[:en1], order[en1:en2]
        gen = torch.Generator().manual_seed(args.seed + task_i)
        mk = lambda ds, bs, sh: DataLoader(ds, batch_size=bs, shuffle=sh, generator=gen if sh else None, drop_last=False)
        tr_f, tr_r = subset(x_all, y_all, forget_cls), subset(x_all, y_all, remain_cls)
        protos = calculate_prototypes(model, subset(x_all, y_all, forget_cls + remain_cls), batch_size=500, device=dev)   # :633-646
        return dict(loader_f=mk(tr_f, args.batch_size, True), loader_r=mk(tr_r, args.batch_size, True),
                    te_f=mk(subset(x_te, y_te, forget_cls), 5 * args.batch_size, False),
                    te_r=mk(subset(x_te, y_te, remain_cls), 5 * args.batch_size, False), protos=protos, info=dict(forget_cls=forget_cls))

    report, ema_model = run_tasks(model, args, task_data, dev, out, geo["depth"])
    return report, out, (model if ema_model is None else (model, ema_model))


if __name__ == "__main__":
    main()
