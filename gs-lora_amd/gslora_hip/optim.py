"""Optimizer / LR schedule of the GS-LoRA drivers, restated for the HIP path.

The reference builds them with timm 0.9.2 (train/train_own_forget_cl.py:811-820):
  create_optimizer(args, model) -> torch.optim.AdamW over requires_grad params, weight decay on
      params with ndim > 1 that are not biases (all LoRA matrices), lr=args.lr, eps=args.opt_eps
  create_scheduler(args, opt)   -> CosineLRScheduler(t_initial=epochs, lr_min=min_lr, warmup 0),
      stepped once per epoch with the epoch index.
FusedAdamW keeps torch.optim.Optimizer's interface (param_groups, state_dict, zero_grad) but runs
one gsl_adamw_flat launch over the model's flat LoRA bucket instead of ~10 foreach kernels.
"""
import math

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}
        self.graph_mode = False     # set by gslora_hip.step.GraphedStep: step count / lr are read from device memory
        # fp16 operands: a device float (ViTRunner.overflow_guard()) — step() leaves p / m / v untouched when it holds >= 65504 or a non-finite
        # value (a gradient store of this step's loss-scaled backward saturated); armed per step by gslora_hip.step.gs_lora_step
        self.overflow_guard = None

    # ---- HIP-graph support: a captured step() must not bake the step count (bias corrections) or the lr into the graph
    def graph_sync(self):
        """Bring the device-resident (step, lr) of every flat group in line with the host values (cheap fills, outside the graph)."""
        for gi, group in enumerate(self.param_groups):
            ent = self._flat.get(gi)
            if ent is None or not ent.get("ok"):
                continue
            if "step_dev" not in ent:
                dev = ent["p"].device
                ent["step_dev"] = torch.zeros(1, device=dev, dtype=torch.int64)
                ent["lr_dev"] = torch.zeros(1, device=dev, dtype=torch.float32)
                ent["step_dev_val"], ent["lr_dev_val"] = None, None
            if ent["step_dev_val"] != ent["step"]:
                ent["step_dev"].fill_(ent["step"])
                ent["step_dev_val"] = ent["step"]
            if ent["lr_dev_val"] != group["lr"]:
                ent["lr_dev"].fill_(group["lr"])
                ent["lr_dev_val"] = group["lr"]

    def graph_replayed(self):
        """Host-side bookkeeping after one replay of a captured step(): the device counters advanced by one."""
        for ent in self._flat.values():
            if ent.get("ok") and "step_dev" in ent:
                ent["step"] += 1
                ent["step_dev_val"] += 1
                for p in ent["order"]:
                    torch.autograd.graph.increment_version(p)

    def graph_capturable(self):
        return bool(self._flat) and all(e.get("ok") for e in self._flat.values())

    # ---- checkpointing: the moments live in flat buffers; state_dict() / load_state_dict() speak torch.optim.AdamW's layout
    # (per parameter: step, exp_avg, exp_avg_sq), so an optimizer checkpoint moves between this class and torch.optim.AdamW
    def _param_index(self):
        return {id(p): i for i, p in enumerate(p for g in self.param_groups for p in g["params"])}

    def state_dict(self):
        sd = super().state_dict()
        idx = self._param_index()
        state = {}
        for ent in self._flat.values():
            if ent.get("ok"):
                off = 0
                for p in ent["order"]:
                    n = p.numel()
                    state[idx[id(p)]] = {"step": torch.tensor(float(ent["step"])), "exp_avg": ent["m"][off:off + n].view(p.shape).clone(),
                                         "exp_avg_sq": ent["v"][off:off + n].view(p.shape).clone()}
                    off += n
        for p, st in self.state.items():      # scattered parameters (one launch per tensor)
            if st:
                state[idx[id(p)]] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["m"].clone(), "exp_avg_sq": st["v"].clone()}
        for pid, st in (getattr(self, "_loaded", None) or {}).items():      # loaded, not yet consumed by a step(): still this optimizer's state
            if pid in idx and idx[pid] not in state:
                state[idx[pid]] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
        sd["state"] = state
        return sd

    def load_state_dict(self, state_dict):
        state = state_dict.get("state", {})
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        params = [p for g in self.param_groups for p in g["params"]]
        self._loaded = {id(params[int(i)]): st for i, st in state.items()}
        self.state.clear()
        # Flat groups that exist keep their buffers — captured HIP graphs hold the addresses of m / v / step_dev / lr_dev — and take the
        # loaded moments and step count IN PLACE; graph_sync() then brings the device counters in line before the next replay.
        # A parameter without an entry in the loaded state (an empty or partial checkpoint, e.g. one saved from a fresh optimizer) restarts
        # from zero moments and step 0, as torch.optim does: reset first, then apply what was loaded.
        for ent in self._flat.values():
            if ent.get("ok"):
                ent["m"].zero_()
                ent["v"].zero_()
                ent["step"] = 0
                self._apply_loaded(ent)
        for gi in [gi for gi, ent in self._flat.items() if not ent.get("ok")]:
            del self._flat[gi]

    def _apply_loaded(self, ent):
        """Fill a freshly built flat group from a loaded checkpoint (called once, when the group is first seen after load_state_dict)."""
        loaded = getattr(self, "_loaded", None)
        if not loaded:
            return
        off = 0
        for p in ent["order"]:
            st = loaded.pop(id(p), None)
            n = p.numel()
            if st is not None:
                ent["m"][off:off + n].copy_(st["exp_avg"].reshape(-1).to(ent["m"].device))
                ent["v"][off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(ent["v"].device))
                ent["step"] = int(float(st["step"]))
            off += n

    def _flat_state(self, gi, group):
        """Detect that a group's grad-bearing params tile one contiguous f32 range (the LoRA bucket)."""
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in ps)
        ent = self._flat.get(gi)
        if ent is not None and ent["sig"] == sig:
            return ent
        order = sorted(ps, key=lambda p: p.data_ptr())
        ok = all(p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() for p in order)
        for a, b in zip(order[:-1], order[1:]):
            ok = ok and b.data_ptr() == a.data_ptr() + 4 * a.numel() and b.grad.data_ptr() == a.grad.data_ptr() + 4 * a.numel()
        n = sum(p.numel() for p in order)
        ent = dict(sig=sig, ok=ok, order=order, n=n)
        if ok:
            first = order[0]
            # views over the whole range, built from the underlying storages
            ent["p"] = torch.as_strided(first.data, (n,), (1,))
            ent["g"] = torch.as_strided(first.grad, (n,), (1,))
            old = self._flat.get(gi)
            if old is not None and old.get("ok") and old["n"] == n:
                ent["m"], ent["v"], ent["step"] = old["m"], old["v"], old["step"]
            else:
                ent["m"] = torch.zeros(n, device=first.device, dtype=torch.float32)
                ent["v"] = torch.zeros(n, device=first.device, dtype=torch.float32)
                ent["step"] = 0
                self._apply_loaded(ent)
        self._flat[gi] = ent
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            ent = self._flat_state(gi, group)
            if ent is None:
                continue
            if ent["ok"] and self.graph_mode:      # being captured: counters on the device, host bookkeeping in graph_replayed()
                ent["step_dev"].add_(1)
                ops.adamw_flat_dev(ent["p"], ent["g"], ent["m"], ent["v"], ent["lr_dev"], b1, b2, group["eps"],
                                   group["weight_decay"], ent["step_dev"], guard=self.overflow_guard)
                continue
            if ent["ok"]:
                ent["step"] += 1
                ops.adamw_flat(ent["p"], ent["g"], ent["m"], ent["v"], group["lr"], b1, b2, group["eps"],
                               group["weight_decay"], ent["step"], guard=self.overflow_guard)
                for p in ent["order"]:      # the kernel wrote through raw pointers: tell torch (and the
                    torch.autograd.graph.increment_version(p)   # runner's operand caches) the data changed
                continue
            for p in ent["order"]:            # scattered params: same kernel, one launch per tensor
                st = self.state[p]
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                    ld = getattr(self, "_loaded", {}).pop(id(p), None)
                    if ld is not None:
                        st["step"] = int(float(ld["step"]))
                        st["m"].copy_(ld["exp_avg"].to(p.device)); st["v"].copy_(ld["exp_avg_sq"].to(p.device))
                st["step"] += 1
                ops.adamw_flat(p.data.view(-1), p.grad.contiguous().view(-1), st["m"].view(-1), st["v"].view(-1),
                               group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"], guard=self.overflow_guard)
                torch.autograd.graph.increment_version(p)
        return loss


class CosineLRScheduler:
    """timm CosineLRScheduler subset the drivers use: t_in_epochs, cycle_limit=1, optional linear warm-up."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0):
        self.optimizer, self.t_initial, self.lr_min = optimizer, t_initial, lr_min
        self.warmup_t, self.warmup_lr_init = warmup_t, warmup_lr_init
        self.base = [g["lr"] for g in optimizer.param_groups]
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        if warmup_t:
            for g in optimizer.param_groups:
                g["lr"] = warmup_lr_init

    def lr_at(self, t, base):
        if t < self.warmup_t:
            return self.warmup_lr_init + t * (base - self.warmup_lr_init) / self.warmup_t
        if t >= self.t_initial:
            return self.lr_min
        return self.lr_min + 0.5 * (base - self.lr_min) * (1.0 + math.cos(math.pi * t / self.t_initial))

    def step(self, epoch, metric=None):
        for g, base in zip(self.optimizer.param_groups, self.base):
            g["lr"] = self.lr_at(epoch, base)


def create_optimizer(args, model):
    """timm.optim.create_optimizer for opt='adamw' (util/args.py:38-62 defaults)."""
    opt = getattr(args, "opt", "adamw").lower()
    if opt != "adamw":
        raise NotImplementedError(f"gs-lora_amd implements the AdamW path of timm.create_optimizer, not '{opt}'")
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith(".bias")) else decay).append(p)
    groups = [g for g in (dict(params=no_decay, weight_decay=0.0), dict(params=decay, weight_decay=args.weight_decay))
              if g["params"]]
    kw = dict(lr=args.lr, eps=getattr(args, "opt_eps", None) or 1e-8)
    betas = getattr(args, "opt_betas", None)
    if betas:
        kw["betas"] = tuple(betas)
    return FusedAdamW(groups, **kw)


def create_scheduler(args, optimizer):
    """timm.scheduler.create_scheduler for sched='cosine' -> (scheduler, num_epochs)."""
    if getattr(args, "sched", "cosine") != "cosine":
        raise NotImplementedError("gs-lora_amd implements the cosine schedule the GS-LoRA scripts use")
    sch = CosineLRScheduler(optimizer, t_initial=args.epochs, lr_min=getattr(args, "min_lr", 1e-5),
                            warmup_t=getattr(args, "warmup_epochs", 0), warmup_lr_init=getattr(args, "warmup_lr", 1e-6))
    return sch, args.epochs + getattr(args, "cooldown_epochs", 0)
