"""engine_cl — continual-forgetting engine, MI355X-native drop-in for the reference `engine_cl.py`
(train_one_epoch :12-244, evaluate :247-315, eval_data :318-346, get_structure_loss :349-432,
get_prototype_loss :571-603). Same 31-keyword signature and 10-tuple return; the arithmetic runs in
libgslora_hip.so via gslora_hip.step.gs_lora_step. Differences that are deliberate:
  * one deferred host sync per display interval instead of >= 8 `.item()` per step (meter values
    are identical);
  * under torch.distributed (one process per GPU) the step is data-parallel (see gslora_hip/step.py);
  * `train_one_epoch_regularzation` / `get_reg_loss` (EWC/MAS/L2 baselines) keep their names only.
"""
import os

import torch
import torch.nn as nn

import util.utils as util
from gslora_hip import losses as _losses
from gslora_hip.step import MeterQueue, gs_lora_step, pick_stepper  # noqa: F401
from util.data_prefetcher import data_prefetcher
from util.utils import get_time, train_accuracy  # noqa: F401  (re-exported like the reference)

try:  # logging stays a host-side concern; absent on the GPU box
    import wandb
except Exception:  # pragma: no cover
    wandb = None

DISP_FREQ = 5
VER_FREQ = 100


def _log(payload):
    if wandb is not None and getattr(wandb, "run", None) is not None:
        wandb.log(payload)


def _unwrap(model):
    return model.module if isinstance(model, nn.DataParallel) else model


def train_one_epoch(model: torch.nn.Module, dataloader_forget, dataloader_remain, device, criterion, optimizer,
                    epoch: int, losses_forget, losses_remain, losses_total, losses_structure, top1_forget, top1_remain,
                    beta: float, alpha: float, BND: float, batch: int, testloader_forget, testloader_remain,
                    forget_acc_before: float, highest_H_mean: float, cfg: dict, task_i: str, use_prototype: bool,
                    prototype_dict: dict, prototype_weight_forget: float, prototype_weight_remain: float,
                    losses_prototype_forget, losses_prototype_remain, dataloader_open=None):
    """Train for one epoch (one pass over the remain loader, forget loader cycled), evaluate every
    VER_FREQ steps. Returns the reference's 10-tuple."""
    model.train()
    criterion.train()
    meters = dict(losses_forget=losses_forget, losses_remain=losses_remain, losses_total=losses_total,
                  losses_structure=losses_structure, top1_forget=top1_forget, top1_remain=top1_remain,
                  losses_prototype_forget=losses_prototype_forget, losses_prototype_remain=losses_prototype_remain)
    queue = MeterQueue()
    proto_table = _losses.prototype_table(prototype_dict, device) if use_prototype else None
    # cfg["DATA_ROOT"] == "./data/imagenet100/" selects the 12 ViT-B/16 block groups in the reference (:84); here the
    # groups come from the model's own LoRA bucket (6 for ViT_face, 12 for ModifiedViT), so no switch is needed.
    forget_iter = data_prefetcher(dataloader_forget, device, prefetch=True)
    x_f, y_f = forget_iter.next()
    for x_r, y_r in iter(dataloader_remain):
        x_r, y_r = x_r.to(device), y_r.to(device)
        # small batches are launch-bound: the step is captured once as a HIP graph and replayed (cfg["HIP_GRAPH"], default "auto")
        stepper = pick_stepper(model, optimizer, criterion, cfg, x_r.size(0) + x_f.size(0))
        pack = stepper(x_r, y_r, x_f, y_f, beta=beta, alpha=alpha, BND=BND, use_structure=True, group_type="block",
                       use_prototype=use_prototype, proto_table=proto_table, w_f=prototype_weight_forget,
                       w_r=prototype_weight_remain, BND_pro=cfg.get("BND_pro", 0.0))
        queue.push(pack, x_r.size(0), x_f.size(0))

        if ((batch + 1) % DISP_FREQ == 0) and batch != 0:
            queue.flush(meters)
            m = meters
            _log({f"epoch_loss_forget-{task_i}": m["losses_forget"].avg, f"epoch_loss_remain-{task_i}": m["losses_remain"].avg,
                  f"epoch_acc_forget-{task_i}": m["top1_forget"].avg, f"epoch_acc_remain-{task_i}": m["top1_remain"].avg,
                  f"epoch_loss_total-{task_i}": m["losses_total"].avg, f"epoch_loss_structure-{task_i}": m["losses_structure"].avg,
                  f"epoch_loss_prototype_forget-{task_i}": m["losses_prototype_forget"].avg,
                  f"epoch_loss_prototype_remain-{task_i}": m["losses_prototype_remain"].avg})
            print("Task {} Epoch {} Batch {}\t"
                  "forget {:.4f} ({:.4f})\tremain {:.4f} ({:.4f})\tproto_f {:.4f}\tproto_r {:.4f}\t"
                  "structure {:.4f} ({:.4f})\ttotal {:.4f} ({:.4f})\tP@1 forget {:.3f} ({:.3f})\tP@1 remain {:.3f} ({:.3f})".format(
                      task_i, epoch + 1, batch + 1, m["losses_forget"].val, m["losses_forget"].avg, m["losses_remain"].val,
                      m["losses_remain"].avg, m["losses_prototype_forget"].val, m["losses_prototype_remain"].val,
                      m["losses_structure"].val, m["losses_structure"].avg, m["losses_total"].val, m["losses_total"].avg,
                      m["top1_forget"].val, m["top1_forget"].avg, m["top1_remain"].val, m["top1_remain"].avg))
            # the reference re-binds fresh meters after each display (engine_cl.py:179-187)
            for k in meters:
                meters[k] = util.AverageMeter()

        if ((batch + 1) % VER_FREQ == 0) and batch != 0:
            with torch.no_grad():
                kw = dict(testloader_forget=testloader_forget, testloader_remain=testloader_remain, device=device, batch=batch,
                          epoch=epoch, task_i=task_i, forget_acc_before=forget_acc_before, highest_H_mean=highest_H_mean,
                          cfg=cfg, optimizer=optimizer)
                if dataloader_open is not None:
                    kw["testloader_open"] = dataloader_open
                highest_H_mean = evaluate(model, **kw)
            model.train()

        batch += 1
        x_f, y_f = forget_iter.next()
        if x_f is None:
            forget_iter = data_prefetcher(dataloader_forget, device, prefetch=True)
            x_f, y_f = forget_iter.next()

    queue.flush(meters)
    return (batch, highest_H_mean, meters["losses_forget"], meters["losses_remain"], meters["top1_forget"],
            meters["top1_remain"], meters["losses_total"], meters["losses_structure"], meters["losses_prototype_forget"],
            meters["losses_prototype_remain"])


# dtype eval_data() evaluates in: "fp32" (default: the reference's arithmetic), "bf16", or "model" (the model's own training mode)
EVAL_DTYPE = os.environ.get("GSLORA_EVAL_DTYPE", "fp32").lower()
_EVAL_SAME = ("model", "train", "same", "")
if EVAL_DTYPE not in _EVAL_SAME:      # validated at import (a typo must not surface at the first evaluate(), an eval interval into the run)
    from vit_pytorch_face.vit_face import compute_dtype_of as _cdt
    try:
        _cdt(EVAL_DTYPE)
    except ValueError as e:
        raise ValueError(f"GSLORA_EVAL_DTYPE={EVAL_DTYPE!r}: use one of {_EVAL_SAME[:3]} or a compute dtype name ({e})") from None


def save_rank():
    """True on the rank that owns the shared work directory (rank 0 of an initialised process group, or the only process)."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def collective_hmean(Hmean: float, highest_H_mean: float, device=None):
    """The (H-mean, best H-mean so far) pair of evaluate() as ONE pair for the whole process group: rank 0's values are broadcast, so
    every rank takes the same "new best" branch and returns the same best value. The branch holds a barrier: ranks that disagreed — test
    loaders sharded by a DistributedSampler, an eval dtype set on some ranks only, a per-rank `highest_H_mean` after a resume — would
    otherwise hang there or mismatch the next training collective. A single process keeps its own values."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return Hmean, highest_H_mean
    dev = (device if device is not None else "cuda") if dist.get_backend() == "nccl" else "cpu"
    pair = torch.tensor([Hmean, highest_H_mean], dtype=torch.float64, device=dev)
    dist.broadcast(pair, src=0)
    h, best = pair.tolist()
    return h, best


def evaluate(model, testloader_forget, testloader_remain, device, batch: int, epoch: int, forget_acc_before: float,
             highest_H_mean: float, cfg: dict, optimizer, task_i: str, testloader_open=None):
    """Eval-mode accuracies, H-mean, best-checkpoint save + prune to two (reference :247-315)."""
    model.eval()
    lr = optimizer.param_groups[0]["lr"]
    print("current learning rate:{:.7f}".format(lr))
    print("Perfom evaluation on test set and save checkpoints...")
    forget_acc = eval_data(model, testloader_forget, device, "forget-{}".format(task_i), batch)
    remain_acc = eval_data(model, testloader_remain, device, "remain-{}".format(task_i), batch)
    if testloader_open is not None:
        eval_data(model, testloader_open, device, "open-{}".format(task_i), batch)
    forget_drop = forget_acc_before - forget_acc
    Hmean = 2 * forget_drop * remain_acc / (forget_drop + remain_acc + 1e-8)
    Hmean, highest_H_mean = collective_hmean(Hmean, highest_H_mean, device)      # rank 0's values decide for every rank
    if Hmean > highest_H_mean:
        highest_H_mean = Hmean
        net = model.module if cfg["MULTI_GPU"] else model
        # one process per GPU: every rank evaluates the (replicated) test loaders; ONE rank writes and prunes the shared work
        # directory, the others wait (concurrent writers corrupt the file, the second pruner finds it gone)
        if save_rank():
            path = os.path.join(cfg["WORK_PATH"], "Backbone_{}_Epoch_{}_Batch_{}_Time_{}_checkpoint.pth".format(
                cfg["BACKBONE_NAME"], epoch + 1, batch + 1, get_time()))
            torch.save(net.state_dict(), path)
            if len(os.listdir(cfg["WORK_PATH"])) >= 4:   # keep the two newest checkpoints (+ config.txt)
                ckpts = sorted((f for f in os.listdir(cfg["WORK_PATH"]) if f.endswith(".pth")),
                               key=lambda f: os.path.getmtime(os.path.join(cfg["WORK_PATH"], f)))
                os.remove(os.path.join(cfg["WORK_PATH"], ckpts[0]))
        save_barrier()
    return highest_H_mean


def eval_data(model, dataloader, device, mode: str, batch: int = 0):
    """Accuracy (0-100) in eval mode; the CosFace margin is applied because labels are passed, exactly
    as the reference does (:336). Hits are counted on the device; one host read at the end."""
    from gslora_hip import ops
    model.eval()
    hits, total = None, 0
    # The accuracies — the numbers a forgetting run reports, and what north_star's "< 0.1 pp vs the reference" is about — are taken in the
    # reference's own arithmetic (f32; vit_face.py has no AMP anywhere): the exact-f32 parity kernels, WHATEVER mode the model trains in.
    # bf16 operands flip near-tie predictions (0 of 4 000 predictions differ in f32, 78 in bf16). Evaluation is
    # forward-only on the test set; its cost is in bench.py's `--eval` leg. GSLORA_EVAL_DTYPE=model evaluates in the model's training
    # mode instead (bf16 speed), =bf16 / =fp32 force a mode.
    net = _unwrap(model)
    eval_dt, train_dt = EVAL_DTYPE, getattr(net, "compute_dtype", None)
    if eval_dt in _EVAL_SAME or not hasattr(net, "set_compute_dtype"):
        eval_dt = None
    if eval_dt:
        net.set_compute_dtype(eval_dt)
    try:
        with torch.no_grad():
            for images, labels in dataloader:
                images, labels = images.to(device), labels.to(device).long()
                outputs, _ = model(images, labels)
                h = ops.ce_fwd(outputs.float().contiguous(), labels.contiguous())[1]
                hits = h if hits is None else hits + h
                total += labels.size(0)
    finally:
        if eval_dt:
            net.set_compute_dtype(train_dt)
    accuracy = 100 * (hits.item() if hits is not None else 0.0) / max(total, 1)
    print("Test {} Accuracy:{:2f}%".format(mode, accuracy))
    _log({"Test {} Accuracy".format(mode): accuracy})
    return accuracy


def get_structure_loss(model: torch.nn.Module, imagenet=False):
    """sum over the per-block LoRA groups (6 for ViT_face; 12 `encoder.layers.encoder_layer_{i}.mlp.{0,3}` groups with
    imagenet=True) of sqrt(sum of squares) — differentiable, one HIP launch (reference :349-432 walks named_parameters()
    and launches ~60 micro-kernels)."""
    net = _unwrap(model)
    is_tv = hasattr(net, "encoder") and hasattr(net, "conv_proj")
    if bool(imagenet) != is_tv:
        raise ValueError("get_structure_loss: imagenet=True goes with ModifiedViT (ViT-B/16), imagenet=False with ViT_face")
    return _losses.structure_loss(net, "block")


def get_prototype_loss(output, labels, prototype_dict, distance="kl"):
    """KL(softmax(prototype[label]) || softmax(feature)), batchmean (reference :571-603)."""
    if distance != "kl":
        raise NotImplementedError("gs-lora_amd implements the 'kl' prototype distance the engines use")
    table = _losses.prototype_table(prototype_dict, output.device)
    return _losses.proto_kl_sum(output, labels, table) / output.shape[0]


def get_reg_loss(*args, **kwargs):
    raise NotImplementedError("EWC/MAS/L2 regularisation baselines are outside the GS-LoRA hot path")


def train_one_epoch_regularzation(*args, **kwargs):
    raise NotImplementedError("EWC/MAS/L2/retrain baselines (reference engine_cl.py:463-568) are outside the GS-LoRA hot path")
