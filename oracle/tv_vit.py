"""CPU restatement of the ViT-B/16 side of the path — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Two layers:
  * `VisionTransformer` and its blocks restate the architecture of the third-party dependency the reference builds on:
    torchvision==0.15.1 (reference requirements.txt:16), `torchvision/models/vision_transformer.py` — conv16 patch
    embedding, class token, learned position embedding, pre-norm encoder blocks `x + drop(MHA(ln_1 x))`,
    `x + mlp(ln_2 x)` with mlp = Linear, GELU, Dropout, Linear, Dropout, LayerNorm eps 1e-6, `heads.head` Linear.
    torchvision is NOT installed in this image and its source is not under /root/reference, so this layer is a
    restatement of the published architecture: **parity unpinned** for the torchvision composition itself. What does
    execute for real is torch's own `nn.MultiheadAttention`, `nn.LayerNorm`, `nn.GELU`, `nn.Conv2d` — the same operators
    torchvision calls — and the parameter names / shapes are pinned by the reference's hard-coded group names
    (engine_cl.py:395-403, util/cal_norm.py:91-107) and by the published parameter count 86 567 656 of vit_b_16.
  * `ModifiedViT`, `replace_ffn_with_lora`, `modify_head` below restate the reference's adapter
    (vit_pytorch_face/modified_VIT.py:5-45, util/utils.py:552-621). These ARE pinned: oracle/make_golden_vitb.py runs the
    real reference adapter, engine and helpers on top of this VisionTransformer and tests/test_oracle_golden.py checks
    this file against those vectors (tests/golden/vitb_small*.npz).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from oracle.shims import loralib as lora


class MLPBlock(nn.Sequential):
    def __init__(self, dim, mlp_dim, dropout):
        super().__init__(nn.Linear(dim, mlp_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(mlp_dim, dim), nn.Dropout(dropout))


class EncoderBlock(nn.Module):
    def __init__(self, heads, dim, mlp_dim, dropout, attention_dropout):
        super().__init__()
        self.num_heads = heads
        self.ln_1 = nn.LayerNorm(dim, eps=1e-6)
        self.self_attention = nn.MultiheadAttention(dim, heads, dropout=attention_dropout, batch_first=True)
        self.dropout = nn.Dropout(dropout)
        self.ln_2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = MLPBlock(dim, mlp_dim, dropout)

    def forward(self, inp):
        x = self.ln_1(inp)
        x, _ = self.self_attention(x, x, x, need_weights=False)
        x = self.dropout(x) + inp
        return x + self.mlp(self.ln_2(x))


class Encoder(nn.Module):
    def __init__(self, seq, layers, heads, dim, mlp_dim, dropout, attention_dropout):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.zeros(1, seq, dim))
        self.dropout = nn.Dropout(dropout)
        self.layers = nn.Sequential(OrderedDict(
            (f"encoder_layer_{i}", EncoderBlock(heads, dim, mlp_dim, dropout, attention_dropout)) for i in range(layers)))
        self.ln = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, x):
        return self.ln(self.layers(self.dropout(x + self.pos_embedding)))


class VisionTransformer(nn.Module):
    def __init__(self, cfg, dropout=0.0, attention_dropout=0.0):
        super().__init__()
        p, d = cfg["patch_size"], cfg["dim"]
        self.patch_size, self.hidden_dim = p, d
        self.conv_proj = nn.Conv2d(cfg["channels"], d, kernel_size=p, stride=p)
        self.class_token = nn.Parameter(torch.zeros(1, 1, d))
        seq = (cfg["image_size"] // p) ** 2 + 1
        self.encoder = Encoder(seq, cfg["depth"], cfg["heads"], d, cfg["mlp_dim"], dropout, attention_dropout)
        self.heads = nn.Sequential(OrderedDict(head=nn.Linear(d, cfg["num_class"])))

    def _process_input(self, x):
        n = x.shape[0]
        x = self.conv_proj(x)                       # [n, d, h/p, w/p]
        return x.reshape(n, self.hidden_dim, -1).permute(0, 2, 1)   # tokens row-major over (h, w)

    def forward(self, x):
        x = self._process_input(x)
        x = torch.cat([self.class_token.expand(x.shape[0], -1, -1), x], dim=1)
        return self.heads(self.encoder(x)[:, 0])


# ---- the reference's adapter, restated ------------------------------------------------------------------------------
class ModifiedViT(nn.Module):
    """modified_VIT.py:5-45 — returns (logits, cls embedding); label unused."""

    def __init__(self, vit):
        super().__init__()
        self.conv_proj, self._process_input = vit.conv_proj, vit._process_input
        self.class_token, self.encoder, self.heads = vit.class_token, vit.encoder, vit.heads

    def forward(self, x, label=None):
        x = self._process_input(x)
        x = torch.cat([self.class_token.expand(x.shape[0], -1, -1), x], dim=1)
        emb = self.encoder(x)[:, 0]
        return self.heads(emb), emb


def replace_ffn_with_lora(model, rank):
    """util/utils.py:552-577 — fresh lora.Linear(in, out, r=rank) in place of every nn.Linear of a `.mlp`."""
    for _, mod in list(model.named_modules()):
        if hasattr(mod, "mlp"):
            for name, lyr in list(mod.mlp.named_children()):
                if isinstance(lyr, nn.Linear):
                    setattr(mod.mlp, name, lora.Linear(lyr.in_features, lyr.out_features, r=rank))
    return model


def head_rows(weight, bias, current_id_to_original_id):
    """util/utils.py:580-621 — rows of the old classifier in dict-value order."""
    ids = list(current_id_to_original_id.values())
    return torch.stack([weight[i] for i in ids]), torch.stack([bias[i] for i in ids])


def build(cfg, state_np, dropout=0.0, double=False):
    """ModifiedViT + LoRA with the recipe weights, LoRA-only trainable."""
    m = replace_ffn_with_lora(ModifiedViT(VisionTransformer(cfg, dropout=dropout)), cfg["lora_rank"])
    m.load_state_dict({k: torch.tensor(v) for k, v in state_np.items()}, strict=True)
    lora.mark_only_lora_as_trainable(m)
    return m.double() if double else m


def group_names(depth):
    """engine_cl.py:395-403 / util/cal_norm.py:91-107."""
    return [[f"encoder.layers.encoder_layer_{i}.mlp.{j}.lora_{ab}" for j in (0, 3) for ab in "AB"] for i in range(depth)]


def structure_loss(model):
    """engine_cl.py:349-432 with imagenet=True: sum over blocks of sqrt(sum of squares of the 4 LoRA tensors)."""
    total = 0.0
    for grp in group_names(len(model.encoder.layers)):
        total = total + torch.sqrt(sum((model.get_parameter(n) ** 2).sum() for n in grp))
    return total


def cal_norm(model):
    """util/cal_norm.py:4-146 with imagenet=True: per block, the SUM of the member tensors' Frobenius norms."""
    with torch.no_grad():
        return [sum(torch.norm(model.get_parameter(n), p=2) for n in grp) for grp in group_names(len(model.encoder.layers))]


# ---- the forgetting step on this family (engine_cl.py:59-125 with cfg DATA_ROOT == "./data/imagenet100/") -------------
def step_losses(model, x_r, y_r, x_f, y_f, hyper, proto=None):
    import torch.nn.functional as F
    from oracle.gslora_oracle import prototype_kl, top1_percent
    lo_r, em_r = model(x_r, y_r)
    lo_f, em_f = model(x_f, y_f)
    ce_r, ce_f = F.cross_entropy(lo_r, y_r), F.cross_entropy(lo_f, y_f)
    loss_forget = F.relu(hyper["BND"] - ce_f)
    sl = structure_loss(model)
    out = dict(ce_r=ce_r, ce_f=ce_f, loss_forget=loss_forget, structure=sl, top1_r=top1_percent(lo_r, y_r),
               top1_f=top1_percent(lo_f, y_f), logits_r=lo_r, logits_f=lo_f, emb_r=em_r, emb_f=em_f)
    pro = torch.zeros((), dtype=ce_r.dtype)
    if proto is not None:
        kl_f, kl_r = prototype_kl(em_f, y_f, proto), prototype_kl(em_r, y_r, proto)
        pro = hyper["pro_f_weight"] * F.relu(hyper["BND_pro"] - kl_f) + hyper["pro_r_weight"] * kl_r
        out.update(kl_f=kl_f, kl_r=kl_r)
    out["prototype"] = pro
    out["total"] = loss_forget * hyper["beta"] + ce_r + sl * hyper["alpha"] + pro
    return out


def train_step(model, x_r, y_r, x_f, y_f, hyper, opt_state=None, step=1, lr=1e-2, proto=None):
    """One step in place on `model` (autograd + the oracle's AdamW). Returns (losses, grads {name: tensor}, opt_state)."""
    from oracle.gslora_oracle import adamw_update
    losses = step_losses(model, x_r, y_r, x_f, y_f, hyper, proto)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    gs = torch.autograd.grad(losses["total"], [p for _, p in named], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(p)) for (n, p), g in zip(named, gs)}
    opt_state = opt_state or {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in named}
    new_opt = {}
    with torch.no_grad():
        for n, p in named:
            m, v = opt_state[n]
            q, m, v = adamw_update(p.detach(), grads[n], m, v, step, lr, hyper.get("wd", 0.05))
            p.copy_(q)
            new_opt[n] = (m, v)
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}, grads, new_opt
