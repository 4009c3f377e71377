"""engine — single-task forgetting engine, MI355X-native drop-in for the reference `engine.py`
(train_one_epoch :13-433, evaluate :436-498, eval_data :501-529, get_structure_loss :532-687).
Differences from engine_cl that the reference has and that are kept: the structure term is gated
by `epoch < cfg["ALPHA_EPOCH"]` (:82-90), the prototype bound is the literal 18 (:105), grouping
comes from cfg["GROUP_TYPE"] in {block, lora, matrix} (:585-650), and with cfg["few_shot"] and a
longer forget loader the roles of the two loaders are swapped (:53-236)."""
import contextlib
import os

import torch
import torch.nn as nn

import loralib as _lora
import util.utils as util
from engine_cl import DISP_FREQ, VER_FREQ, _log, _unwrap  # noqa: F401
from engine_cl import eval_data as _eval_data_cl
from util.utils import get_time
from gslora_hip import losses as _losses
from gslora_hip.step import MeterQueue, gs_lora_step, pick_stepper  # noqa: F401
from util.data_prefetcher import data_prefetcher

PROTO_BND = 18   # hard-coded in the reference (engine.py:105)


def train_one_epoch(model: torch.nn.Module, dataloader_forget, dataloader_remain, device, criterion, optimizer,
                    epoch: int, losses_forget, losses_remain, losses_total, losses_structure, top1_forget, top1_remain,
                    beta: float, alpha: float, BND: float, batch: int, testloader_forget, testloader_remain,
                    forget_acc_before: float, highest_H_mean: float, cfg: dict, dataloader_open=None,
                    prototype_weight_forget: float = 0.0, prototype_weight_remain: float = 0.0, use_prototype: bool = False,
                    prototype_dict: dict = None, losses_prototype_forget=None, losses_prototype_remain=None):
    model.train()
    criterion.train()
    losses_prototype_forget = losses_prototype_forget or util.AverageMeter()
    losses_prototype_remain = losses_prototype_remain or util.AverageMeter()
    meters = dict(losses_forget=losses_forget, losses_remain=losses_remain, losses_total=losses_total,
                  losses_structure=losses_structure, top1_forget=top1_forget, top1_remain=top1_remain,
                  losses_prototype_forget=losses_prototype_forget, losses_prototype_remain=losses_prototype_remain)
    queue = MeterQueue()
    proto_table = _losses.prototype_table(prototype_dict, device) if use_prototype else None
    use_structure = not (epoch < cfg.get("ALPHA_EPOCH", 0))
    group_type = cfg.get("GROUP_TYPE", "block")
    _check_group_pos(model, cfg.get("GROUP_POS", "FFN"))
    # few-shot inversion: iterate the LONGER forget loader, cycle the remain loader (engine.py:53-236)
    swap = bool(cfg.get("few_shot")) and len(dataloader_forget) > len(dataloader_remain)
    outer, inner = (dataloader_forget, dataloader_remain) if swap else (dataloader_remain, dataloader_forget)
    inner_iter = data_prefetcher(inner, device, prefetch=True)
    xi, yi = inner_iter.next()
    for xo, yo in iter(outer):
        xo, yo = xo.to(device), yo.to(device)
        (x_f, y_f, x_r, y_r) = (xo, yo, xi, yi) if swap else (xi, yi, xo, yo)
        stepper = pick_stepper(model, optimizer, criterion, cfg, x_r.size(0) + x_f.size(0))    # HIP graph for launch-bound batches
        pack = stepper(x_r, y_r, x_f, y_f, beta=beta, alpha=alpha, BND=BND, use_structure=use_structure, group_type=group_type,
                       use_prototype=use_prototype, proto_table=proto_table, w_f=prototype_weight_forget,
                       w_r=prototype_weight_remain, BND_pro=PROTO_BND)
        queue.push(pack, x_r.size(0), x_f.size(0))
        if ((batch + 1) % DISP_FREQ == 0) and batch != 0:
            queue.flush(meters)
            m = meters
            _log({"epoch_loss_forget": m["losses_forget"].avg, "epoch_loss_remain": m["losses_remain"].avg,
                  "epoch_acc_forget": m["top1_forget"].avg, "epoch_acc_remain": m["top1_remain"].avg,
                  "epoch_loss_total": m["losses_total"].avg, "epoch_loss_structure": m["losses_structure"].avg})
            print("Epoch {} Batch {}\tforget {:.4f} ({:.4f})\tremain {:.4f} ({:.4f})\tstructure {:.4f} ({:.4f})\t"
                  "total {:.4f} ({:.4f})\tP@1 forget {:.3f} ({:.3f})\tP@1 remain {:.3f} ({:.3f})".format(
                      epoch + 1, batch + 1, m["losses_forget"].val, m["losses_forget"].avg, m["losses_remain"].val,
                      m["losses_remain"].avg, m["losses_structure"].val, m["losses_structure"].avg, m["losses_total"].val,
                      m["losses_total"].avg, m["top1_forget"].val, m["top1_forget"].avg, m["top1_remain"].val,
                      m["top1_remain"].avg))
            for k in meters:
                meters[k] = util.AverageMeter()
        if ((batch + 1) % VER_FREQ == 0) and batch != 0:
            with torch.no_grad():
                highest_H_mean = evaluate(model, testloader_forget=testloader_forget, testloader_remain=testloader_remain,
                                          device=device, batch=batch, epoch=epoch, forget_acc_before=forget_acc_before,
                                          highest_H_mean=highest_H_mean, cfg=cfg, optimizer=optimizer,
                                          testloader_open=dataloader_open)
            model.train()
        batch += 1
        xi, yi = inner_iter.next()
        if xi is None:
            inner_iter = data_prefetcher(inner, device, prefetch=True)
            xi, yi = inner_iter.next()
    queue.flush(meters)
    return (batch, highest_H_mean, meters["losses_forget"], meters["losses_remain"], meters["top1_forget"],
            meters["top1_remain"], meters["losses_total"], meters["losses_structure"], meters["losses_prototype_forget"],
            meters["losses_prototype_remain"])


@contextlib.contextmanager
def _evaluation_copy(model):
    """What the reference gets from `copy.deepcopy(model).eval()` (engine.py:449, :514), without copying 77 MB of frozen weights per
    call: the un-merged weights of the adapter layers are stashed, the model is evaluated in eval() (merged) mode, and on exit the stashed
    tensors are copied back and every module's `training` flag restored — the training weights come back BIT FOR BIT (an arithmetic
    un-merge would leave one f32 rounding per evaluation, which the reference's training weights never see)."""
    layers = [m for m in model.modules() if isinstance(m, (_lora.Linear, _lora.MergedLinear)) and m.r > 0 and m.merge_weights]
    modes = [(m, m.training) for m in model.modules()]
    stash = [(l, l.weight.detach().clone()) for l in layers if not l.merged]
    try:
        model.eval()
        yield model
    finally:
        with torch.no_grad():
            for l, w in stash:
                l.weight.copy_(w)      # in place: bumps _version, the operand caches of the HIP path refresh
                l.merged = False
        for m, t in modes:
            m.training = t


def evaluate(model, testloader_forget, testloader_remain, device, batch, epoch, forget_acc_before, highest_H_mean, cfg,
             optimizer, testloader_open=None):
    """Reference engine.py:436-498: accuracies of a COPY of the model in eval mode, H-mean (no epsilon in this engine), save of the best
    checkpoint (merged weights, as the copy is in eval mode) and pruning to two checkpoints once the work directory holds >= 3 entries."""
    lr = optimizer.param_groups[0]["lr"]
    print("current learning rate:{:.7f}".format(lr))
    print("Perfom evaluation on test set and save checkpoints...")
    with _evaluation_copy(model) as m:
        forget_acc = _eval_data_cl(m, testloader_forget, device, "forget", batch)
        remain_acc = _eval_data_cl(m, testloader_remain, device, "remain", batch)
        if testloader_open is not None:
            _eval_data_cl(m, testloader_open, device, "open", batch)
        forget_drop = forget_acc_before - forget_acc
        Hmean = 2 * forget_drop * remain_acc / (forget_drop + remain_acc)
        from engine_cl import collective_hmean, save_barrier, save_rank
        Hmean, highest_H_mean = collective_hmean(Hmean, highest_H_mean, device)      # rank 0's values decide for every rank (the branch holds a barrier)
        if Hmean > highest_H_mean:
            highest_H_mean = Hmean
            net = m.module if cfg["MULTI_GPU"] else m
            if save_rank():      # one process per GPU: one writer / pruner of the shared work directory (see engine_cl.evaluate)
                torch.save(net.state_dict(), os.path.join(cfg["WORK_PATH"], "Backbone_{}_Epoch_{}_Batch_{}_Time_{}_checkpoint.pth".format(
                    cfg["BACKBONE_NAME"], epoch + 1, batch + 1, get_time())))
                if len(os.listdir(cfg["WORK_PATH"])) >= 3:
                    ckpts = sorted((f for f in os.listdir(cfg["WORK_PATH"]) if f.endswith(".pth")),
                                   key=lambda f: os.path.getmtime(os.path.join(cfg["WORK_PATH"], f)))
                    os.remove(os.path.join(cfg["WORK_PATH"], ckpts[0]))
            save_barrier()
    return highest_H_mean


def eval_data(model, dataloader, device, mode: str, batch: int = 0):
    """Reference engine.py:501-529: accuracy (0-100) of a copy of the model in eval mode; the caller's model keeps its mode and weights."""
    with _evaluation_copy(model) as m:
        return _eval_data_cl(m, dataloader, device, mode, batch)


def _check_group_pos(model, group_pos):
    """group_pos follows the model's lora_pos (reference engine.py:585-658: the group names are taken from one or the other)."""
    site = getattr(_unwrap(model), "lora_pos", "FFN")
    if group_pos not in ("FFN", "Attention"):
        raise ValueError("group_pos must be 'FFN' or 'Attention'")
    if group_pos != site:
        raise ValueError(f"group_pos={group_pos!r} but the model was built with lora_pos={site!r}")


def get_structure_loss(model: torch.nn.Module, num_layers: int = 6, group_type: str = "block", group_pos: str = "FFN"):
    """Group lasso over the FFN adapter groups (block / lora / matrix) or, with group_pos='Attention', over one
    (to_qkv.lora_A, to_qkv.lora_B) group per block (reference :651-656, group_type ignored there)."""
    _check_group_pos(model, group_pos)
    return _losses.structure_loss(_unwrap(model), group_type)
