"""Step helpers of the reference's util/utils.py that sit on the GS-LoRA path
(AverageMeter :316-332, train_accuracy :354-368, count_trainable_parameters :423-425,
reinitialize_lora_parameters :428-441, calculate_prototypes :502-549), backed by the HIP model.
Data plumbing, verification and the ImageNet head surgery of that file are out of scope."""
import datetime
import math

import torch
import torch.nn as nn


class AverageMeter(object):
    """val / avg / sum / count running mean — same update arithmetic as the reference."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def train_accuracy(output, target, topk=(1,)):
    """top-1 precision in percent (the engines only ever ask for topk=(1,)); one fused HIP launch."""
    if tuple(topk) != (1,):
        raise NotImplementedError("gs-lora_amd train_accuracy implements topk=(1,) (all the engines use)")
    from gslora_hip import ops
    out = ops.ce_fwd(output.detach().float().contiguous(), target.to(output.device, torch.int64).contiguous())
    return out[1] * (100.0 / target.size(0))


def count_trainable_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def reinitialize_lora_parameters(model):
    """Fresh adapters for the next task: A ~ kaiming_uniform(a=sqrt(50)), B = 0 (in place, so the
    parameters stay views of the flat LoRA bucket)."""
    with torch.no_grad():
        for name, param in model.named_parameters():
            if "lora" in name:
                if not isinstance(param, nn.Parameter):
                    raise ValueError(f"Parameter {name} is not an instance of nn.Parameter.")
                if "lora_A" in name:
                    nn.init.kaiming_uniform_(param, a=math.sqrt(50))
                elif "lora_B" in name:
                    nn.init.zeros_(param)


def calculate_prototypes(backbone, dataset, batch_size=32, device="cuda", aug_num=0):
    """Per-class mean embedding in eval (merged-LoRA) mode; leaves the model in eval() like the
    reference does. Class sums are accumulated on the device (one index_add per batch) instead of a
    per-sample Python loop; the result dict holds CPU tensors as before."""
    if aug_num != 0:
        raise NotImplementedError("RandAugment prototype augmentation (aug_num>0) is data plumbing outside the hot path")
    from torch.utils.data import DataLoader
    backbone.eval()
    backbone.to(device)
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False)
    sums = counts = None
    with torch.no_grad():
        for images, labels in loader:
            images, labels = images.to(device), labels.to(device).long()
            _, emb = backbone(images, labels)
            if sums is None:
                ncls = backbone.loss.weight.shape[0]
                sums = torch.zeros(ncls, emb.shape[1], device=emb.device)
                counts = torch.zeros(ncls, device=emb.device)
            sums.index_add_(0, labels, emb)
            counts.index_add_(0, labels, torch.ones_like(labels, dtype=torch.float32))
    sums, counts = sums.cpu(), counts.cpu()
    return {int(c): (sums[c] / counts[c]) for c in torch.nonzero(counts).flatten().tolist()}


def get_time():
    return (str(datetime.datetime.now())[:-10]).replace(" ", "-").replace(":", "-")
