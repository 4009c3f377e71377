"""Engine-level goldens from the REAL reference (imported unmodified from /root/reference; build container only):

  tests/golden/engine_cl_traj.npz   24 steps of engine_cl.train_one_epoch on the full ViT-P8S8 (4 epochs x 6 remain batches, forget
                                    loader cycled, cosine lr per epoch), per-step meter values, eval accuracies / H-mean before and
                                    after, final eval logits; then one more epoch in which engine_cl.evaluate runs inside the engine
                                    (batch counter 99): H-mean, best-checkpoint save + prune to two, training resumed after the merge
                                    round trip.
  tests/golden/engine_single.npz    engine.train_one_epoch: normal branch, few-shot loop inversion, epoch < ALPHA_EPOCH, the literal
                                    prototype bound 18, the three groupings; engine.evaluate / eval_data (deep-copied model).
  tests/golden/chain2.npz           two tasks chained the way train_own_forget_cl.py does it: train -> eval() -> save merged state ->
                                    load_state_dict -> reinitialize_lora_parameters -> train.

  tests/golden/engine_cl_acc.npz     the trajectory's model evaluated by the real eval_data on 1 000 + 1 000 held-out samples before / after
                                    the 24 steps: accuracies at 0.1 pp resolution + per-sample predictions and decision margins.

  tests/golden/chain4.npz            BASELINE config 3 as written: four tasks with the shipped per-task beta / prototype-weight lists, the
                                    alpha warm-up switch and the EMA model (its un-merged-into-merged quirk included).

Inputs come from oracle/scenarios.py (shared with tests/test_hip_engines.py); fixtures hold OUTPUTS only.
Usage: python oracle/make_golden_engines.py [traj] [single] [chain]
"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import recipe, scenarios as S  # noqa: E402
from oracle.make_golden import build_reference_model, install_shims  # noqa: E402

NAMES = ["losses_forget", "losses_remain", "losses_total", "losses_structure", "top1_forget", "top1_remain",
         "losses_prototype_forget", "losses_prototype_remain"]


class UpdateLog:
    """Records every AverageMeter.update(val, n) of the reference (the engine re-binds fresh meters after each display)."""

    def __init__(self, rutil):
        self.vals, self.rutil, self.orig = [], rutil, rutil.AverageMeter.update

    def __enter__(self):
        log, orig = self.vals, self.orig

        def update(meter, val, n=1):
            log.append(float(val))
            return orig(meter, val, n)
        self.rutil.AverageMeter.update = update
        return self

    def __exit__(self, *a):
        self.rutil.AverageMeter.update = self.orig

    def steps(self):
        return np.array(self.vals, dtype=np.float64).reshape(-1, 8)


def fresh_meters(rutil):
    return {k: rutil.AverageMeter() for k in NAMES}


def lora_state(model):
    return {n: p.detach().numpy().copy() for n, p in model.named_parameters() if p.requires_grad}


def gen_traj(out):
    import engine_cl
    from util import utils as rutil
    cfg, T = recipe.cfg_full(), S.TRAJ
    rem, forg, test_rem, test_forg = S.class_loaders(cfg, T["n_remain"], T["n_forget"], T["batch"])
    state = recipe.make_state(cfg)

    def emb_fn(st, x):
        m = build_reference_model(cfg, st).train()
        with torch.no_grad():
            return torch.cat([m(x[i:i + 12]) for i in range(0, x.shape[0], 12)])
    res = {}
    res["head_bias"], res["loss_weight"] = S.discriminative_head(emb_fn, state, cfg, (rem, forg))
    state["mlp_head.0.bias"], state["loss.weight"] = res["head_bias"], res["loss_weight"]
    model = build_reference_model(cfg, state)
    proto = S.prototypes(cfg)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=T["lr"], weight_decay=T["wd"], eps=1e-8)
    crit = torch.nn.CrossEntropyLoss()
    dev = torch.device("cpu")
    x_ev = torch.cat([test_rem.batches[0][0], test_forg.batches[0][0]])
    y_ev = torch.cat([test_rem.batches[0][1], test_forg.batches[0][1]])

    def snapshot(tag):
        with torch.no_grad():
            res[f"acc_forget_{tag}"] = np.float64(engine_cl.eval_data(model, test_forg, dev, "forget", 0))
            res[f"acc_remain_{tag}"] = np.float64(engine_cl.eval_data(model, test_rem, dev, "remain", 0))
            model.eval()
            lo, _ = model(x_ev, y_ev)
            res[f"eval_logits_{tag}"] = lo.numpy().copy()
            lo_nl = model(x_ev)           # embedding without label
            res[f"eval_emb_{tag}"] = lo_nl.numpy().copy()
        model.train()

    snapshot("before")
    cfgd = {"DATA_ROOT": "./data/casia100/", "BND_pro": T["BND_pro"], "MULTI_GPU": False, "WORK_PATH": "/tmp", "BACKBONE_NAME": "VIT"}
    meters = fresh_meters(rutil)
    batch_ctr, hmean = 0, 0.0
    epoch_avgs = []
    with UpdateLog(rutil) as log:
        for epoch in range(T["epochs"]):
            for g in opt.param_groups:
                g["lr"] = S.cosine_lr(epoch, T["epochs"], T["lr"], T["lr_min"])
            ret = engine_cl.train_one_epoch(
                model=model, dataloader_forget=forg, dataloader_remain=rem, device=dev, criterion=crit, optimizer=opt, epoch=epoch,
                beta=T["beta"], alpha=T["alpha"], BND=T["BND"], batch=batch_ctr, testloader_forget=None, testloader_remain=None,
                forget_acc_before=T["forget_acc_before"], highest_H_mean=hmean, cfg=cfgd, task_i="0", use_prototype=True,
                prototype_dict=proto, prototype_weight_forget=T["pro_f_weight"], prototype_weight_remain=T["pro_r_weight"], **meters)
            batch_ctr, hmean = ret[0], ret[1]
            meters = dict(losses_forget=ret[2], losses_remain=ret[3], top1_forget=ret[4], top1_remain=ret[5], losses_total=ret[6],
                          losses_structure=ret[7], losses_prototype_forget=ret[8], losses_prototype_remain=ret[9])
            epoch_avgs.append([meters[k].avg for k in NAMES])
    res["step_updates"] = log.steps()                  # [24, 8] in scenarios.REF_UPDATE_ORDER
    res["epoch_avgs"] = np.array(epoch_avgs, dtype=np.float64)
    res["batch_ctr"] = np.int64(batch_ctr)
    snapshot("after")
    st = lora_state(model)
    res["lora_norms_after"] = np.array([np.linalg.norm(v) for v in st.values()], dtype=np.float64)
    for k in ("transformer.layers.0.1.fn.fn.net.0.lora_A", "transformer.layers.5.1.fn.fn.net.3.lora_B",
              "transformer.layers.2.1.fn.fn.net.0.lora_B"):
        res[f"param_after::{k}"] = st[k]

    # ---- part 2: evaluate() inside the engine (batch counter 97 -> evaluation after the third step), save + prune -------------------
    work = tempfile.mkdtemp(prefix="gsl_golden_")
    open(os.path.join(work, "config.txt"), "w").write("cfg\n")
    for i, name in enumerate(["Backbone_VIT_Epoch_1_Batch_10_Time_old_checkpoint.pth", "Backbone_VIT_Epoch_1_Batch_20_Time_old_checkpoint.pth"]):
        p = os.path.join(work, name)
        torch.save({"dummy": torch.zeros(1)}, p)
        os.utime(p, (time.time() - 1000 + 10 * i, time.time() - 1000 + 10 * i))
    cfgd2 = dict(cfgd, WORK_PATH=work)
    meters = fresh_meters(rutil)
    with UpdateLog(rutil) as log:
        ret = engine_cl.train_one_epoch(
            model=model, dataloader_forget=forg, dataloader_remain=rem, device=dev, criterion=crit, optimizer=opt, epoch=4,
            beta=T["beta"], alpha=T["alpha"], BND=T["BND"], batch=S.EVAL["batch0"], testloader_forget=test_forg, testloader_remain=test_rem,
            forget_acc_before=S.EVAL["forget_acc_before"], highest_H_mean=0.0, cfg=cfgd2, task_i="0", use_prototype=True,
            prototype_dict=proto, prototype_weight_forget=T["pro_f_weight"], prototype_weight_remain=T["pro_r_weight"], **meters)
    res["eval_step_updates"] = log.steps()
    res["eval_hmean"] = np.float64(ret[1])
    res["eval_batch_ctr"] = np.int64(ret[0])
    files = sorted(os.listdir(work))
    res["eval_n_files"] = np.int64(len(files))
    res["eval_kept_old"] = np.array([int("Batch_10_" in " ".join(files)), int("Batch_20_" in " ".join(files))])
    new = [f for f in files if f.endswith(".pth") and "_old_" not in f]
    res["eval_new_is_batch100"] = np.int64(len(new) == 1 and "_Epoch_5_Batch_100_" in new[0])
    print("[golden] traj accuracies:", {k: float(v) for k, v in res.items() if k.startswith("acc_")}, "hmean", float(ret[1]), files)
    ck = torch.load(os.path.join(work, new[0]))
    res["ckpt_merged_w_l0_net0_sum"] = np.float64(ck["transformer.layers.0.1.fn.fn.net.0.weight"].double().sum().item())
    res["ckpt_merged_w_l0_net0_row7"] = ck["transformer.layers.0.1.fn.fn.net.0.weight"][7].numpy().copy()
    res["ckpt_n_keys"] = np.int64(len(ck))
    shutil.rmtree(work)
    # a second evaluate() with a LOWER H-mean must not save (forget_acc_before lowered)
    work2 = tempfile.mkdtemp(prefix="gsl_golden_")
    with torch.no_grad():
        h2 = engine_cl.evaluate(model, test_forg, test_rem, dev, batch=199, epoch=5, forget_acc_before=S.EVAL["forget_acc_before"] - 30.0,
                                highest_H_mean=float(ret[1]), cfg=dict(cfgd, WORK_PATH=work2), optimizer=opt, task_i="0")
    res["eval2_hmean_returned"] = np.float64(h2)
    res["eval2_n_files"] = np.int64(len(os.listdir(work2)))
    shutil.rmtree(work2)
    model.train()
    snapshot("final")
    np.savez_compressed(os.path.join(out, "engine_cl_traj.npz"), **res)
    print("[golden] engine_cl_traj:", {k: (v.shape if hasattr(v, "shape") and v.ndim else float(v)) for k, v in res.items()
                                       if not k.startswith("eval_logits") and not k.startswith("eval_emb") and "::" not in k and "row7" not in k})


def gen_acc(out):
    """tests/golden/engine_cl_acc.npz: the trajectory scenario (same model, head, loaders, 24 steps of the real engine_cl.train_one_epoch)
    evaluated with the REAL eval_data on 1 000 + 1 000 held-out samples before and after training: accuracies at 0.1 pp resolution and
    the per-sample predictions (argmax of the margin logits, as eval_data takes it, engine_cl.py:336-339)."""
    import engine_cl
    from util import utils as rutil
    cfg, T, A = recipe.cfg_full(), S.TRAJ, S.ACC
    rem, forg, _, _ = S.class_loaders(cfg, T["n_remain"], T["n_forget"], T["batch"])
    big_rem, big_forg = S.class_eval_loaders(cfg, A["n_per_split"], A["batch"])
    g = np.load(os.path.join(out, "engine_cl_traj.npz"))
    state = recipe.make_state(cfg)
    state["mlp_head.0.bias"], state["loss.weight"] = g["head_bias"], g["loss_weight"]
    model = build_reference_model(cfg, state)
    proto = S.prototypes(cfg)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=T["lr"], weight_decay=T["wd"], eps=1e-8)
    crit = torch.nn.CrossEntropyLoss()
    dev = torch.device("cpu")
    res = {}

    def snapshot(tag):
        t0 = time.time()
        with torch.no_grad():
            res[f"acc_forget_{tag}"] = np.float64(engine_cl.eval_data(model, big_forg, dev, "forget", 0))
            res[f"acc_remain_{tag}"] = np.float64(engine_cl.eval_data(model, big_rem, dev, "remain", 0))
            model.eval()
            for kind, ld in (("forget", big_forg), ("remain", big_rem)):
                lo = torch.cat([model(x, y)[0] for x, y in ld.batches])
                res[f"pred_{kind}_{tag}"] = lo.argmax(1).numpy().astype(np.int16)
                top2 = lo.topk(2, dim=1).values
                res[f"margin_{kind}_{tag}"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)     # how close each decision is
        model.train()
        print(f"[golden] acc snapshot {tag}: forget {res[f'acc_forget_{tag}']:.2f} remain {res[f'acc_remain_{tag}']:.2f} ({time.time() - t0:.0f} s)", flush=True)

    snapshot("before")
    cfgd = {"DATA_ROOT": "./data/casia100/", "BND_pro": T["BND_pro"], "MULTI_GPU": False, "WORK_PATH": "/tmp", "BACKBONE_NAME": "VIT"}
    meters = fresh_meters(rutil)
    batch_ctr = 0
    with UpdateLog(rutil) as log:
        for epoch in range(T["epochs"]):
            for gq in opt.param_groups:
                gq["lr"] = S.cosine_lr(epoch, T["epochs"], T["lr"], T["lr_min"])
            ret = engine_cl.train_one_epoch(
                model=model, dataloader_forget=forg, dataloader_remain=rem, device=dev, criterion=crit, optimizer=opt, epoch=epoch,
                beta=T["beta"], alpha=T["alpha"], BND=T["BND"], batch=batch_ctr, testloader_forget=None, testloader_remain=None,
                forget_acc_before=T["forget_acc_before"], highest_H_mean=0.0, cfg=cfgd, task_i="0", use_prototype=True,
                prototype_dict=proto, prototype_weight_forget=T["pro_f_weight"], prototype_weight_remain=T["pro_r_weight"], **meters)
            batch_ctr = ret[0]
            meters = dict(losses_forget=ret[2], losses_remain=ret[3], top1_forget=ret[4], top1_remain=ret[5], losses_total=ret[6],
                          losses_structure=ret[7], losses_prototype_forget=ret[8], losses_prototype_remain=ret[9])
    assert np.abs(log.steps() - g["step_updates"]).max() == 0.0, "the trajectory must be the one of engine_cl_traj.npz"
    snapshot("after")
    np.savez_compressed(os.path.join(out, "engine_cl_acc.npz"), **res)
    print("[golden] engine_cl_acc:", {k: float(v) for k, v in res.items() if k.startswith("acc_")})


def gen_acc_stat(out, only=None):
    """tests/golden/engine_cl_acc_stat.npz: the statistical accuracy evidence (scenarios.ACC_STAT x scenarios.acc_seeds(name)) — per scenario and
    data seed: the four accuracies of the REAL eval_data (engine_cl.py:318-346) before / after training with the REAL
    engine_cl.train_one_epoch, per-sample predictions and decision margins; per scenario the two frozen head tensors (data)."""
    import engine_cl
    from util import utils as rutil
    cfg = recipe.cfg_full()
    path = os.path.join(out, "engine_cl_acc_stat.npz")
    res = dict(np.load(path)) if os.path.exists(path) else {}
    for name in S.ACC_STAT:      # resume: cells already generated by an interrupted run (python ... acc_stat real s3 s4 adds the missing seeds)
        part = path + f".part_{name}.npz"
        if os.path.exists(part):
            res.update({k: v for k, v in np.load(part).items()})
    traj = np.load(os.path.join(out, "engine_cl_traj.npz"))
    dev = torch.device("cpu")
    for name, sc in S.ACC_STAT.items():
        if only and name not in only:
            continue
        base = recipe.make_state(cfg)
        if name == "harsh":
            hb, lw = traj["head_bias"], traj["loss_weight"]
        else:
            def emb_fn(st, x):
                m = build_reference_model(cfg, st).train()
                with torch.no_grad():
                    return torch.cat([m(x[i:i + 20]) for i in range(0, x.shape[0], 20)])
            hb, lw = S.discriminative_head(emb_fn, base, cfg, (S.acc_stat_head_set(cfg, sc["noise"]),), common=sc["common"])
            res[f"{name}::head_bias"], res[f"{name}::loss_weight"] = hb, lw
        for seed in S.acc_seeds(name):
            if only and f"s{seed}" not in only and any(o.startswith("s") for o in only):
                continue
            t0 = time.time()
            state = dict(base)
            state["mlp_head.0.bias"], state["loss.weight"] = hb, lw
            model = build_reference_model(cfg, state)
            rem, forg, big_rem, big_forg = S.acc_stat_loaders(cfg, name, seed)
            proto = S.prototypes(cfg)
            opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=sc["lr"], weight_decay=sc["wd"], eps=1e-8)
            crit = torch.nn.CrossEntropyLoss()
            key = f"{name}::s{seed}::"

            def snapshot(tag):
                # ONE pass per split: the real eval_data computes the accuracy, a forward hook records the logits it saw (the margin
                # logits, labels passed: engine_cl.py:336) for the per-sample predictions and decision margins
                seen = []
                hook = model.register_forward_hook(lambda mod, args, outp: seen.append(outp[0].detach().clone()))
                try:
                    with torch.no_grad():
                        for kind, ld in (("forget", big_forg), ("remain", big_rem)):
                            seen.clear()
                            res[key + f"acc_{kind}_{tag}"] = np.float64(engine_cl.eval_data(model, ld, dev, kind, 0))
                            lo = torch.cat(seen)
                            assert lo.shape[0] == sc["n_per_split"]
                            res[key + f"pred_{kind}_{tag}"] = lo.argmax(1).numpy().astype(np.int16)
                            top2 = lo.topk(2, dim=1).values
                            res[key + f"margin_{kind}_{tag}"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
                finally:
                    hook.remove()
                model.train()
            snapshot("before")
            cfgd = {"DATA_ROOT": "./data/casia100/", "BND_pro": sc["BND_pro"], "MULTI_GPU": False, "WORK_PATH": "/tmp", "BACKBONE_NAME": "VIT"}
            meters, batch_ctr = fresh_meters(rutil), 0
            for epoch in range(sc["epochs"]):
                for gq in opt.param_groups:
                    gq["lr"] = S.cosine_lr(epoch, sc["epochs"], sc["lr"], sc["lr_min"])
                ret = engine_cl.train_one_epoch(
                    model=model, dataloader_forget=forg, dataloader_remain=rem, device=dev, criterion=crit, optimizer=opt, epoch=epoch,
                    beta=sc["beta"], alpha=sc["alpha"], BND=sc["BND"], batch=batch_ctr, testloader_forget=None, testloader_remain=None,
                    forget_acc_before=sc["forget_acc_before"], highest_H_mean=0.0, cfg=cfgd, task_i="0", use_prototype=True,
                    prototype_dict=proto, prototype_weight_forget=sc["pro_f_weight"], prototype_weight_remain=sc["pro_r_weight"], **meters)
                batch_ctr = ret[0]
                meters = dict(losses_forget=ret[2], losses_remain=ret[3], top1_forget=ret[4], top1_remain=ret[5], losses_total=ret[6],
                              losses_structure=ret[7], losses_prototype_forget=ret[8], losses_prototype_remain=ret[9])
            snapshot("after")
            print(f"[golden] acc_stat {name} seed {seed}: " + " ".join(f"{k}={float(res[key + k]):.2f}" for k in
                  ("acc_forget_before", "acc_remain_before", "acc_forget_after", "acc_remain_after")) + f" ({time.time() - t0:.0f} s)", flush=True)
            np.savez_compressed(path + f".part_{name}.npz", **{k: v for k, v in res.items() if k.startswith(name + "::")})
    # (the two scenarios can be generated by two processes — `acc_stat harsh` / `acc_stat real` — each leaving its .part file; any call merges)
    for name in S.ACC_STAT:
        part = path + f".part_{name}.npz"
        if os.path.exists(part):
            res.update({k: v for k, v in np.load(part).items()})
    np.savez_compressed(path, **res)


def gen_single(out):
    import engine as eng
    from util import utils as rutil
    cfg = recipe.cfg_small2()
    H = S.SINGLE_HYPER
    res = {}
    dev = torch.device("cpu")
    for name, sc in S.SINGLE.items():
        model = build_reference_model(cfg, recipe.make_state(cfg))
        rem, forg = S.loaders(cfg, sc["n_remain"], sc["n_forget"], sc["batch"], seed=sc["seed"])
        proto = S.prototypes(cfg, sc["proto_scale"])
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=H["lr"], weight_decay=H["wd"], eps=1e-8)
        crit = torch.nn.CrossEntropyLoss()
        cfgd = {"few_shot": sc["few_shot"], "ALPHA_EPOCH": sc["ALPHA_EPOCH"], "NUM_LAYERS": cfg["depth"], "GROUP_TYPE": sc["GROUP_TYPE"],
                "GROUP_POS": "FFN", "MULTI_GPU": False, "WORK_PATH": "/tmp", "BACKBONE_NAME": "VIT"}
        meters = fresh_meters(rutil)
        with UpdateLog(rutil) as log:
            ret = eng.train_one_epoch(
                model=model, dataloader_forget=forg, dataloader_remain=rem, device=dev, criterion=crit, optimizer=opt, epoch=sc["epoch"],
                beta=H["beta"], alpha=H["alpha"], BND=H["BND"], batch=0, testloader_forget=None, testloader_remain=None,
                forget_acc_before=0.0, highest_H_mean=0.0, cfg=cfgd, prototype_weight_forget=H["pro_f_weight"],
                prototype_weight_remain=H["pro_r_weight"], use_prototype=sc["use_prototype"], prototype_dict=proto, **meters)
        res[f"{name}::step_updates"] = log.steps()
        res[f"{name}::batch_ctr"] = np.int64(ret[0])
        for n, p in model.named_parameters():
            if p.requires_grad:
                res[f"{name}::grad_last::{n}"] = p.grad.numpy().copy()
                res[f"{name}::param::{n}"] = p.detach().numpy().copy()
        if name == "normal":      # engine.evaluate / eval_data work on deep copies: the training weights are untouched (:449, :514)
            before = {n: p.detach().clone() for n, p in model.named_parameters()}
            work = tempfile.mkdtemp(prefix="gsl_golden_")
            with torch.no_grad():
                res["normal::acc_forget"] = np.float64(eng.eval_data(model, forg, dev, "forget", 0))
                res["normal::acc_remain"] = np.float64(eng.eval_data(model, rem, dev, "remain", 0))
                res["normal::hmean"] = np.float64(eng.evaluate(model, forg, rem, dev, batch=9, epoch=0, forget_acc_before=100.0,
                                                               highest_H_mean=0.0, cfg=dict(cfgd, WORK_PATH=work), optimizer=opt))
            res["normal::n_files"] = np.int64(len(os.listdir(work)))
            shutil.rmtree(work)
            assert model.training and all(torch.equal(before[n], p) for n, p in model.named_parameters())
    np.savez_compressed(os.path.join(out, "engine_single.npz"), **res)
    for name in S.SINGLE:
        print(f"[golden] engine_single/{name}: steps {res[f'{name}::step_updates'].shape[0]}, first step", np.round(res[f"{name}::step_updates"][0], 4))
    print("[golden] engine_single eval:", res["normal::acc_forget"], res["normal::acc_remain"], res["normal::hmean"])


def gen_chain(out):
    import engine_cl
    from util import utils as rutil
    from util.cal_norm import get_norm_of_lora
    cfg, C = recipe.cfg_full(), S.CHAIN
    model = build_reference_model(cfg, recipe.make_state(cfg))
    dev = torch.device("cpu")
    crit = torch.nn.CrossEntropyLoss()
    proto = S.prototypes(cfg)
    res = {}
    work = tempfile.mkdtemp(prefix="gsl_golden_")
    cfgd = {"DATA_ROOT": "./data/casia100/", "BND_pro": C["BND_pro"], "MULTI_GPU": False, "WORK_PATH": work, "BACKBONE_NAME": "VIT"}
    model.train()
    x_ev = torch.tensor(recipe.make_images(cfg, 3, seed=901, tag="xev"))
    y_ev = torch.tensor(recipe.make_labels(cfg, 3, seed=901, tag="yev"))
    for task in range(2):
        if task > 0:
            model.load_state_dict(torch.load(os.path.join(work, "task-level", f"Backbone_task_{task - 1}.pth")))
            rutil.reinitialize_lora_parameters(model)
            st = lora_state(model)
            assert all(np.all(v == 0) for k, v in st.items() if k.endswith("lora_B"))
            bound = {k: np.sqrt(6.0 / (51.0 * v.shape[1])) for k, v in st.items() if k.endswith("lora_A")}
            assert all(np.abs(st[k]).max() <= b and np.abs(st[k]).max() > 0.8 * b for k, b in bound.items())
            with torch.no_grad():
                for k, v in S.chain_lora_A(cfg, task).items():
                    model.get_parameter(k).copy_(v)
                lo, _ = model(x_ev, y_ev)
                res["logits_after_reload_reinit"] = lo.numpy().copy()
        rem, forg = S.loaders(cfg, C["n_remain"], C["n_forget"], C["batch"], seed=10 + task)
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=C["lr"], weight_decay=C["wd"], eps=1e-8)
        meters = fresh_meters(rutil)
        with UpdateLog(rutil) as log:
            engine_cl.train_one_epoch(
                model=model, dataloader_forget=forg, dataloader_remain=rem, device=dev, criterion=crit, optimizer=opt, epoch=0,
                beta=C["betas"][task], alpha=C["alpha"], BND=C["BND"], batch=0, testloader_forget=None, testloader_remain=None,
                forget_acc_before=0.0, highest_H_mean=0.0, cfg=cfgd, task_i=task, use_prototype=True, prototype_dict=proto,
                prototype_weight_forget=C["pro_f_weight"], prototype_weight_remain=C["pro_r_weight"], **meters)
        res[f"task{task}::step_updates"] = log.steps()
        res[f"task{task}::norm_list"] = np.array([float(v) for v in get_norm_of_lora(model, type="L2", group_num=cfg["depth"])])
        model.eval()
        os.makedirs(os.path.join(work, "task-level"), exist_ok=True)
        torch.save(model.state_dict(), os.path.join(work, "task-level", f"Backbone_task_{task}.pth"))
        with torch.no_grad():
            lo, _ = model(x_ev, y_ev)
            res[f"task{task}::eval_logits"] = lo.numpy().copy()
        sd = model.state_dict()
        res[f"task{task}::saved_w_l3_net3_row5"] = sd["transformer.layers.3.1.fn.fn.net.3.weight"][5].numpy().copy()
        res[f"task{task}::saved_lora_B_l3_net3"] = sd["transformer.layers.3.1.fn.fn.net.3.lora_B"].numpy().copy()
        model.train()
    shutil.rmtree(work)
    np.savez_compressed(os.path.join(out, "chain2.npz"), **res)
    print("[golden] chain2: |logits(after reload+reinit) - eval logits(task0)| =",
          float(np.abs(res["logits_after_reload_reinit"] - res["task0::eval_logits"]).max()))


def gen_chain4(out):
    """tests/golden/chain4.npz — BASELINE config 3 as the reference runs it: four tasks with per-task cl_beta_list / cl_prof_list, the
    alpha warm-up switch and the EMA model, issued by hand in the order of train/train_own_forget_cl.py (:502-507 EMA copy in eval(),
    :515-536 reload + re-init, :803-834 per-task counters and meters, :999-1011 per-task / per-epoch hyper-parameters, :1013 scheduler
    step, :1015-1056 train_one_epoch, :1058-1098 EMA update + eval, :1100-1106 norms, :1696-1705 merged save). The statements between
    the `# ref` marks restate those lines on an argparse-like namespace; everything they call is the REAL reference code."""
    import copy
    import types
    import engine_cl
    from util import utils as rutil
    from util.cal_norm import get_norm_of_lora
    cfg, C = recipe.cfg_small6(), S.CHAIN4
    args = types.SimpleNamespace(cl_beta_list=list(C["cl_beta_list"]), cl_prof_list=list(C["cl_prof_list"]), warmup_alpha=C["warmup_alpha"],
                                 alpha_epoch=C["alpha_epoch"], big_alpha=C["big_alpha"], alpha=C["alpha"], pro_f_weight=C["pro_f_weight"],
                                 pro_r_weight=C["pro_r_weight"], average_weight=True, ema_epoch=C["ema_epoch"], ema_decay=C["ema_decay"],
                                 BND=C["BND"], num_tasks=C["num_tasks"], prototype=True)
    BACKBONE = build_reference_model(cfg, recipe.make_state(cfg))
    DEVICE = torch.device("cpu")
    LOSS = torch.nn.CrossEntropyLoss()
    prototype = S.prototypes(cfg, C["proto_scale"])
    res = {}
    work = tempfile.mkdtemp(prefix="gsl_golden_")
    os.makedirs(os.path.join(work, "task-level"))
    cfgd = {"DATA_ROOT": "./data/casia100/", "BND_pro": C["BND_pro"], "MULTI_GPU": False, "WORK_PATH": work, "BACKBONE_NAME": "VIT"}
    x_ev = torch.tensor(recipe.make_images(cfg, 4, seed=902, tag="xev"))
    y_ev = torch.tensor(recipe.make_labels(cfg, 4, seed=902, tag="yev"))
    # ref :502-509
    BACKBONE.eval()
    ema_model = copy.deepcopy(BACKBONE)
    BACKBONE.train()
    for task_i in range(args.num_tasks):
        if task_i > 0:      # ref :524-536
            BACKBONE.load_state_dict(torch.load(os.path.join(work, "task-level", "Backbone_task_{}.pth".format(task_i - 1))))
            rutil.reinitialize_lora_parameters(BACKBONE)
            with torch.no_grad():      # seeded adapter matrices (the kaiming draws come from the device RNG)
                for k, v in S.chain_lora_A(cfg, task_i).items():
                    BACKBONE.get_parameter(k).copy_(v)
        train_loader_remain, train_loader_forget, testloader_remain, testloader_forget = S.chain4_task(cfg, task_i)
        highest_H_mean = 0.0      # ref :803
        OPTIMIZER = torch.optim.AdamW([p for p in BACKBONE.parameters() if p.requires_grad], lr=C["lr"], weight_decay=C["wd"], eps=1e-8)
        batch = 0                 # ref :824
        meters = fresh_meters(rutil)      # ref :827-834
        with torch.no_grad():
            forget_acc_before = engine_cl.eval_data(BACKBONE, testloader_forget, DEVICE, "forget-{}".format(task_i), batch)
            remain_acc_before = engine_cl.eval_data(BACKBONE, testloader_remain, DEVICE, "remain-{}".format(task_i), batch)
        res[f"task{task_i}::acc_before"] = np.array([forget_acc_before, remain_acc_before])
        # ref :999-1004
        cl_beta = args.cl_beta_list[task_i]
        if len(args.cl_prof_list) != 0:
            args.pro_f_weight = args.cl_prof_list[task_i]
        BACKBONE.train()
        hyper, ema_accs = [], []
        with UpdateLog(rutil) as log:
            for epoch in range(C["epochs"]):
                if args.warmup_alpha:      # ref :1007-1011
                    if epoch < args.alpha_epoch:
                        args.alpha = 0
                    else:
                        args.alpha = args.big_alpha
                for g in OPTIMIZER.param_groups:      # ref :1013 lr_scheduler.step(epoch) (timm cosine, restated in scenarios.cosine_lr)
                    g["lr"] = S.cosine_lr(epoch, C["epochs"], C["lr"], C["lr_min"])
                hyper.append([cl_beta, args.pro_f_weight, args.alpha, OPTIMIZER.param_groups[0]["lr"]])
                ret = engine_cl.train_one_epoch(      # ref :1015-1056
                    model=BACKBONE, dataloader_forget=train_loader_forget, dataloader_remain=train_loader_remain,
                    testloader_forget=testloader_forget, testloader_remain=testloader_remain, device=DEVICE, criterion=LOSS,
                    optimizer=OPTIMIZER, epoch=epoch, batch=batch, beta=cl_beta, BND=args.BND, forget_acc_before=forget_acc_before,
                    highest_H_mean=highest_H_mean, cfg=cfgd, alpha=args.alpha, task_i=task_i, use_prototype=args.prototype,
                    prototype_dict=prototype, prototype_weight_forget=args.pro_f_weight, prototype_weight_remain=args.pro_r_weight,
                    **meters)
                batch, highest_H_mean = ret[0], ret[1]
                meters = dict(losses_forget=ret[2], losses_remain=ret[3], top1_forget=ret[4], top1_remain=ret[5], losses_total=ret[6],
                              losses_structure=ret[7], losses_prototype_forget=ret[8], losses_prototype_remain=ret[9])
                if args.average_weight:      # ref :1058-1098
                    if epoch == args.ema_epoch:
                        with torch.no_grad():
                            BACKBONE_COPY = copy.deepcopy(BACKBONE)
                            ema_model.eval()
                            for param, ema_param in zip(BACKBONE_COPY.parameters(), ema_model.parameters()):
                                ema_param.data = param.data.detach()
                    elif epoch > args.ema_epoch:
                        with torch.no_grad():
                            BACKBONE_COPY = copy.deepcopy(BACKBONE)
                            ema_model.eval()
                            for param, ema_param in zip(BACKBONE_COPY.parameters(), ema_model.parameters()):
                                ema_param.data = ema_param.data.detach() * args.ema_decay + param.data.detach() * (1 - args.ema_decay)
                    if epoch < args.ema_epoch:
                        pass
                    else:
                        with torch.no_grad():
                            fa = engine_cl.eval_data(ema_model, testloader_forget, DEVICE, "forget-ema-{}".format(task_i), batch)
                            ra = engine_cl.eval_data(ema_model, testloader_remain, DEVICE, "remain-ema-{}".format(task_i), batch)
                        ema_accs.append([fa, ra])
        res[f"task{task_i}::step_updates"] = log.steps()
        res[f"task{task_i}::hyper"] = np.array(hyper, dtype=np.float64)
        res[f"task{task_i}::ema_accs"] = np.array(ema_accs, dtype=np.float64)
        res[f"task{task_i}::batch_ctr"] = np.int64(batch)
        res[f"task{task_i}::norm_list"] = np.array([float(v) for v in get_norm_of_lora(BACKBONE, type="L2", group_num=cfg["depth"])])
        with torch.no_grad():
            ema_model.eval()
            res[f"task{task_i}::ema_eval_logits"] = ema_model(x_ev, y_ev)[0].numpy().copy()
            res[f"task{task_i}::ema_lora_B_l1_net0"] = dict(ema_model.named_parameters())["transformer.layers.1.1.fn.fn.net.0.lora_B"].numpy().copy()
            res[f"task{task_i}::acc_after"] = np.array([engine_cl.eval_data(BACKBONE, testloader_forget, DEVICE, "forget", batch),
                                                        engine_cl.eval_data(BACKBONE, testloader_remain, DEVICE, "remain", batch)])
        BACKBONE.eval()      # ref :1696-1705
        torch.save(BACKBONE.state_dict(), os.path.join(work, "task-level", "Backbone_task_{}.pth".format(task_i)))
        with torch.no_grad():
            res[f"task{task_i}::eval_logits"] = BACKBONE(x_ev, y_ev)[0].numpy().copy()
        sd = BACKBONE.state_dict()
        res[f"task{task_i}::saved_w_l4_net3_row5"] = sd["transformer.layers.4.1.fn.fn.net.3.weight"][5].numpy().copy()
        BACKBONE.train()
    shutil.rmtree(work)
    np.savez_compressed(os.path.join(out, "chain4.npz"), **res)
    for t in range(args.num_tasks):
        print(f"[golden] chain4 task {t}: steps {res[f'task{t}::step_updates'].shape[0]} hyper {res[f'task{t}::hyper'].tolist()} "
              f"ema accs {res[f'task{t}::ema_accs'].tolist()} acc before/after {res[f'task{t}::acc_before']} {res[f'task{t}::acc_after']}")


def gen_poolmean(out):
    """ViT_face(pool='mean') (vit_face.py:540): forward, eval forward and the LoRA gradients of the three-term loss on the 3-layer model."""
    import engine as eng
    import engine_cl
    from vit_pytorch_face import ViT_face
    import loralib as lora
    cfg = recipe.cfg_small2()
    model = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=cfg["num_class"], image_size=cfg["image_size"], patch_size=cfg["patch_size"],
                     dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], mlp_dim=cfg["mlp_dim"], lora_rank=cfg["lora_rank"], pool="mean")
    model.load_state_dict({k: torch.tensor(v) for k, v in recipe.make_state(cfg).items()}, strict=True)
    lora.mark_only_lora_as_trainable(model)
    model.train()
    rem, forg = S.loaders(cfg, 1, 1, 3, seed=5)
    (xr, yr), (xf, yf) = rem.batches[0], forg.batches[0]
    proto = S.prototypes(cfg)
    H = S.SINGLE_HYPER
    crit = torch.nn.CrossEntropyLoss()
    lo_r, em_r = model(xr, yr)
    lo_f, em_f = model(xf, yf)
    sl = eng.get_structure_loss(model, num_layers=cfg["depth"], group_type="block", group_pos="FFN")
    kl_f = engine_cl.get_prototype_loss(em_f, yf, proto)
    kl_r = engine_cl.get_prototype_loss(em_r, yr, proto)
    total = (H["beta"] * torch.relu(H["BND"] - crit(lo_f, yf)) + crit(lo_r, yr) + H["alpha"] * sl
             + H["pro_f_weight"] * torch.relu(2.0 - kl_f) + H["pro_r_weight"] * kl_r)
    model.zero_grad()
    total.backward()
    res = {"logits_r": lo_r.detach().numpy(), "emb_r": em_r.detach().numpy(), "total": np.float64(total.item())}
    for n, p in model.named_parameters():
        if p.requires_grad:
            res[f"grad::{n}"] = p.grad.numpy().copy()
    model.eval()
    with torch.no_grad():
        res["eval_logits_f"] = model(xf, yf)[0].numpy()
    np.savez_compressed(os.path.join(out, "pool_mean_small2.npz"), **res)
    print("[golden] pool_mean_small2: total", res["total"])


def gen_protoaug(out):
    """calculate_prototypes(aug_num = 3) of the reference (util/utils.py:502-549) with the deterministic transforms stand-in."""
    from unittest import mock
    from util import utils as rutil
    cfg = recipe.cfg_small2()
    model = build_reference_model(cfg, recipe.make_state(cfg))
    x = torch.tensor(recipe.make_images(cfg, 7, seed=77, tag="xp"))
    y = torch.tensor(recipe.make_labels(cfg, 7, seed=77, tag="yp", lo=0, hi=4))
    ds = S.TransformDataset(x, y)
    with mock.patch.object(rutil, "transforms", S.StubTransforms), mock.patch.object(rutil, "DataLoader", torch.utils.data.DataLoader), \
            mock.patch.object(rutil, "ConcatDataset", torch.utils.data.ConcatDataset):
        protos = rutil.calculate_prototypes(model, ds, batch_size=5, device="cpu", aug_num=3)
    keys = sorted(protos)
    np.savez_compressed(os.path.join(out, "proto_aug_small2.npz"), keys=np.array(keys, dtype=np.int64),
                        vals=np.stack([protos[k].numpy() for k in keys]).astype(np.float32))
    print("[golden] proto_aug_small2:", keys)


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    out = os.path.join(ROOT, "tests", "golden")
    only = sys.argv[1:]
    if not only or "single" in only:
        gen_single(out)
    if not only or "chain" in only:
        gen_chain(out)
    if not only or "traj" in only:
        gen_traj(out)
    if not only or "acc" in only:
        gen_acc(out)
    if not only or "acc_stat" in only:
        gen_acc_stat(out, [a for a in only if a in S.ACC_STAT or (a[:1] == "s" and a[1:].isdigit())])
    if not only or "chain4" in only:
        gen_chain4(out)
    if not only or "poolmean" in only:
        gen_poolmean(out)
    if not only or "protoaug" in only:
        gen_protoaug(out)


if __name__ == "__main__":
    main()
