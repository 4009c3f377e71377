"""CPU oracle: a restatement of the GS-LoRA forgetting train step of bjzhb666/GS-LoRA.

TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product package (gs-lora_amd/) never
does and fails loudly when its HIP library is missing.

Parity status: **pinned** — tests/golden/*.npz hold outputs of the real reference
(imported from /root/reference in the build container by oracle/make_golden.py,
third-party loralib/timm restated in oracle/shims/) and tests/test_oracle_golden.py
checks every function here against them. The reference's own test/ directory holds
no vectors for this path.

Everything is plain fp32 PyTorch on CPU (floating-point path; torch autograd supplies
the backward so the oracle's gradients are independent of the hand-derived HIP
backward). Each function cites the reference file:line it follows (paths relative
to /root/reference).
"""
import math
import random

import torch
import torch.nn.functional as F

COS_S = 64.0   # vit_face.py:158 CosFace s
COS_M = 0.35   # vit_face.py:158 CosFace m
LN_EPS = 1e-5  # nn.LayerNorm default (vit_face.py:319, :499)


def to_torch(state_np, requires_grad_lora=False):
    st = {k: torch.tensor(v, dtype=torch.float32) for k, v in state_np.items()}
    if requires_grad_lora:
        for k in st:
            if "lora_" in k:
                st[k].requires_grad_(True)
    return st


# ---------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------
def patchify(img, p):
    """einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)'  (vit_face.py:530):
    token t = h*W + w ; feature f = (p1*p + p2)*C + c."""
    b, c, hh, ww = img.shape
    h, w = hh // p, ww // p
    x = img.reshape(b, c, h, p, w, p).permute(0, 2, 4, 3, 5, 1)
    return x.reshape(b, h * w, p * p * c)


def lora_linear(x, W, bias, A, B, r, merged):
    """loralib 0.1.2 Linear.forward (call sites vit_face.py:330,333): lora_alpha=1 so
    scaling = 1/r; when merged (eval) W already holds W + (B@A)/r."""
    y = F.linear(x, W, bias)
    if r > 0 and not merged:
        y = y + (x @ A.t() @ B.t()) * (1.0 / r)
    return y


def merged_delta(A, B, r):
    """loralib 0.1.2 MergedLinear.merge_AB with enable_lora=[True]*3: rows of output group g = B_g @ A_g
    (A [3r, in] rows g*r.., B [out, r] rows of group g) — the grouped conv1d written as a batched matmul."""
    ng = A.shape[0] // r
    return torch.bmm(B.view(ng, -1, r), A.view(ng, r, -1)).reshape(B.shape[0], A.shape[1])


def vit_forward(st, img, label, cfg, merged=False, dropout_masks=None, dropout_p=0.0):
    """ViT_face.forward (vit_face.py:523-548) with Transformer (:442-446), Attention
    (:358-379, scale = dim**-0.5 :346), FeedForward (:329-335, exact-erf GELU) and
    CosFace (:171-208). Dropout is the identity for parity runs (p = 0 / eval mode); dropout_p > 0 applies torch's own Bernoulli
    dropout at the reference's 19 sites (emb_dropout :537, to_out :353-356, FeedForward :329-335) — used only by bench.py's CPU
    baseline, which times the step "as the reference trains" (dropout 0.1).
    Returns (logits|None, emb)."""
    dp = (lambda t: F.dropout(t, dropout_p, True)) if dropout_p > 0 else (lambda t: t)
    p, d, hds, r = cfg["patch_size"], cfg["dim"], cfg["heads"], cfg["lora_rank"]
    attn_lora = cfg.get("lora_pos", "FFN") == "Attention"
    x = patchify(img.float(), p)
    x = F.linear(x, st["patch_to_embedding.weight"], st["patch_to_embedding.bias"])
    b, n, _ = x.shape
    x = torch.cat((st["cls_token"].expand(b, -1, -1), x), dim=1)
    x = dp(x + st["pos_embedding"][:, : n + 1])
    scale = d ** -0.5
    for i in range(cfg["depth"]):
        a = f"transformer.layers.{i}.0.fn"
        f = f"transformer.layers.{i}.1.fn"
        xn = F.layer_norm(x, (d,), st[f"{a}.norm.weight"], st[f"{a}.norm.bias"], LN_EPS)
        qkv = F.linear(xn, st[f"{a}.fn.to_qkv.weight"])
        if attn_lora and not merged:      # loralib.MergedLinear: one rank-r adapter per q / k / v group, scaling 1/r
            qkv = qkv + (xn @ merged_delta(st[f"{a}.fn.to_qkv.lora_A"], st[f"{a}.fn.to_qkv.lora_B"], r).t()) * (1.0 / r)
        q, k, v = qkv.chunk(3, dim=-1)
        sp = lambda t: t.reshape(b, n + 1, hds, -1).permute(0, 2, 1, 3)
        q, k, v = sp(q), sp(k), sp(v)
        dots = torch.einsum("bhid,bhjd->bhij", q, k) * scale
        attn = dots.softmax(dim=-1)
        o = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(b, n + 1, -1)
        x = dp(F.linear(o, st[f"{a}.fn.to_out.0.weight"], st[f"{a}.fn.to_out.0.bias"])) + x
        xn = F.layer_norm(x, (d,), st[f"{f}.norm.weight"], st[f"{f}.norm.bias"], LN_EPS)
        rf = 0 if attn_lora else r
        h = lora_linear(xn, st[f"{f}.fn.net.0.weight"], st[f"{f}.fn.net.0.bias"],
                        st.get(f"{f}.fn.net.0.lora_A"), st.get(f"{f}.fn.net.0.lora_B"), rf, merged)
        h = dp(F.gelu(h))
        y = lora_linear(h, st[f"{f}.fn.net.3.weight"], st[f"{f}.fn.net.3.bias"],
                        st.get(f"{f}.fn.net.3.lora_A"), st.get(f"{f}.fn.net.3.lora_B"), rf, merged)
        x = dp(y) + x
    x = x.mean(dim=1) if cfg.get("pool", "cls") == "mean" else x[:, 0]      # vit_face.py:540
    emb = F.layer_norm(x, (d,), st["mlp_head.0.weight"], st["mlp_head.0.bias"], LN_EPS)
    if label is None:
        return None, emb
    return cosface(emb, st["loss.weight"], label), emb


def cosface(emb, W, label):
    """CosFace.forward (vit_face.py:171-208): F.normalize eps 1e-12, margin applied at
    the label column whenever a label is given (train AND eval)."""
    cosine = F.linear(F.normalize(emb), F.normalize(W))
    one_hot = torch.zeros_like(cosine)
    one_hot.scatter_(1, label.view(-1, 1).long(), 1)
    return (one_hot * (cosine - COS_M) + (1.0 - one_hot) * cosine) * COS_S


def merge_lora(st, cfg, sign=+1.0):
    """loralib Linear.train(False)/train(True): W += / -= (B@A)/r in place."""
    r = cfg["lora_rank"]
    out = dict(st)
    if cfg.get("lora_pos", "FFN") == "Attention":
        for i in range(cfg["depth"]):
            pre = f"transformer.layers.{i}.0.fn.fn.to_qkv"
            out[f"{pre}.weight"] = st[f"{pre}.weight"] + sign * merged_delta(st[f"{pre}.lora_A"], st[f"{pre}.lora_B"], r) / r
        return out
    for i in range(cfg["depth"]):
        for j in (0, 3):
            pre = f"transformer.layers.{i}.1.fn.fn.net.{j}"
            out[f"{pre}.weight"] = st[f"{pre}.weight"] + sign * (st[f"{pre}.lora_B"] @ st[f"{pre}.lora_A"]) / r
    return out


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------
def lora_groups(cfg, group_type="block"):
    """Group naming of engine_cl.get_structure_loss (engine_cl.py:388-394) and
    engine.get_structure_loss block/lora/matrix (engine.py:585-650)."""
    L = cfg["depth"]
    if cfg.get("lora_pos", "FFN") == "Attention":      # engine.py:651-656, util/cal_norm.py:108-120: one (A, B) group per block
        return [[f"transformer.layers.{i}.0.fn.fn.to_qkv.lora_A", f"transformer.layers.{i}.0.fn.fn.to_qkv.lora_B"] for i in range(L)]
    n = lambda i, j, ab: f"transformer.layers.{i}.1.fn.fn.net.{j}.lora_{ab}"
    if group_type == "block":
        return [[n(i, 0, "A"), n(i, 0, "B"), n(i, 3, "A"), n(i, 3, "B")] for i in range(L)]
    if group_type == "lora":
        return [[n(i, 0, "A"), n(i, 0, "B")] for i in range(L)] + [[n(i, 3, "A"), n(i, 3, "B")] for i in range(L)]
    if group_type == "matrix":
        return ([[n(i, 0, "A")] for i in range(L)] + [[n(i, 0, "B")] for i in range(L)]
                + [[n(i, 3, "A")] for i in range(L)] + [[n(i, 3, "B")] for i in range(L)])
    raise ValueError(group_type)


def group_lasso_norms(st, cfg, group_type="block"):
    """Per-group sqrt(sum over the group's tensors of sum(t**2)) (engine_cl.py:416-425)."""
    return torch.stack([torch.sqrt(sum(torch.sum(st[k] ** 2) for k in g)) for g in lora_groups(cfg, group_type)])


def structure_loss(st, cfg, group_type="block"):
    """get_structure_loss (engine_cl.py:349-432): sum of the group-lasso norms, no eps."""
    return group_lasso_norms(st, cfg, group_type).sum()


def cal_norm_of_lora(st, cfg, group_type="block"):
    """util/cal_norm.py:130-137 type 'L2': per group the SUM of the tensors' Frobenius norms."""
    return torch.stack([sum(torch.norm(st[k]) for k in g) for g in lora_groups(cfg, group_type)])


def group_mask(norms, tau=0.0):
    """Group-selection mask (not present in the reference):
    selected[g] = norm_g > tau on the fp32 group-lasso norms."""
    return norms > tau


def prototype_kl(emb, labels, proto_table):
    """get_prototype_loss 'kl' (engine_cl.py:571-603):
    kl_div(log_softmax(emb), log_softmax(proto[label]), batchmean, log_target=True)."""
    pt = proto_table[labels.long()]
    return F.kl_div(F.log_softmax(emb, dim=1), F.log_softmax(pt, dim=1), reduction="batchmean", log_target=True)


def top1_percent(logits, labels):
    """train_accuracy (util/utils.py:354-368), topk=(1,)."""
    return (logits.argmax(dim=1) == labels).float().sum() * (100.0 / labels.shape[0])


def step_losses(st, cfg, x_r, y_r, x_f, y_f, hyper, proto=None):
    """Loss side of the engine_cl.train_one_epoch loop body (engine_cl.py:59-120)."""
    lr_logits, lr_emb = vit_forward(st, x_r, y_r, cfg, dropout_p=hyper.get("dropout", 0.0))
    lf_logits, lf_emb = vit_forward(st, x_f, y_f, cfg, dropout_p=hyper.get("dropout", 0.0))
    ce_r = F.cross_entropy(lr_logits, y_r)
    ce_f = F.cross_entropy(lf_logits, y_f)
    loss_forget = F.relu(hyper["BND"] - ce_f)
    s_loss = structure_loss(st, cfg, hyper.get("group_type", "block"))
    out = dict(ce_r=ce_r, ce_f=ce_f, loss_forget=loss_forget, structure=s_loss,
               top1_r=top1_percent(lr_logits, y_r), top1_f=top1_percent(lf_logits, y_f),
               logits_r=lr_logits, logits_f=lf_logits, emb_r=lr_emb, emb_f=lf_emb)
    if proto is not None:
        kl_f = prototype_kl(lf_emb, y_f, proto)
        kl_r = prototype_kl(lr_emb, y_r, proto)
        pro = hyper["pro_f_weight"] * F.relu(hyper["BND_pro"] - kl_f) + hyper["pro_r_weight"] * kl_r
        out.update(kl_f=kl_f, kl_r=kl_r)
    else:
        pro = torch.tensor(0.0)
    out["prototype"] = pro
    out["total"] = loss_forget * hyper["beta"] + ce_r + s_loss * hyper["alpha"] + pro
    return out


# ---------------------------------------------------------------------------
# optimizer / schedule
# ---------------------------------------------------------------------------
def adamw_update(p, g, m, v, step, lr, wd=0.05, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor math as timm.create_optimizer configures it
    (train_own_forget_cl.py:811-813, util/args.py:38-62). Returns new (p, m, v)."""
    p = p * (1.0 - lr * wd)
    m = m + (g - m) * (1.0 - b1)
    v = v * b2 + g * g * (1.0 - b2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def cosine_lr(epoch, lr0=1e-2, lr_min=1e-5, epochs=100):
    """timm CosineLRScheduler, warmup 0, cycle_limit 1, stepped per epoch
    (train_own_forget_cl.py:818-820, :1013)."""
    if epoch >= epochs:
        return lr_min
    return lr_min + 0.5 * (lr0 - lr_min) * (1.0 + math.cos(math.pi * epoch / epochs))


def train_step(st_np, cfg, x_r, y_r, x_f, y_f, hyper, opt_state=None, step=1, lr=1e-2, proto=None):
    """One full step (engine_cl.py:59-125): losses, backward (autograd), AdamW on the
    LoRA tensors. Returns (losses dict, grads {name: tensor}, new state {name: tensor},
    new opt_state)."""
    st = to_torch(st_np, requires_grad_lora=True)
    losses = step_losses(st, cfg, x_r, y_r, x_f, y_f, hyper, proto)
    names = [k for k in st if "lora_" in k]
    grads = torch.autograd.grad(losses["total"], [st[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(st[k])) for k, g in zip(names, grads)}
    opt_state = opt_state or {k: (torch.zeros_like(st[k]), torch.zeros_like(st[k])) for k in names}
    new_st = {k: v.detach() for k, v in st.items()}
    new_opt = {}
    with torch.no_grad():
        for k in names:
            m, v = opt_state[k]
            p, m, v = adamw_update(new_st[k], grads[k], m, v, step, lr, hyper.get("wd", 0.05))
            new_st[k] = p
            new_opt[k] = (m, v)
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}, grads, new_st, new_opt


# ---------------------------------------------------------------------------
# host-side known answers
# ---------------------------------------------------------------------------
def class_order(num_class=100, seed=1337):
    """train_own_forget_cl.py:198-204."""
    order = list(range(num_class))
    random.seed(seed)
    random.shuffle(order)
    return order


def reinit_bound(fan_in, a=math.sqrt(50)):
    """reinitialize_lora_parameters (util/utils.py:428-441): kaiming_uniform_(a=sqrt(50))
    -> U(+-sqrt(6/((1+a^2) fan_in)))."""
    return math.sqrt(6.0 / ((1.0 + a * a) * fan_in))


def calculate_prototypes(st, cfg, images, labels, batch_size=500):
    """util/utils.py:502-549: eval-mode (merged) forward, per-class mean of emb."""
    stm = merge_lora(st, cfg, +1.0)
    sums, counts = {}, {}
    with torch.no_grad():
        for i in range(0, images.shape[0], batch_size):
            _, emb = vit_forward(stm, images[i:i + batch_size], labels[i:i + batch_size], cfg, merged=True)
            for e, l in zip(emb, labels[i:i + batch_size]):
                l = int(l)
                sums[l] = sums.get(l, 0) + e
                counts[l] = counts.get(l, 0) + 1
    return {l: sums[l] / counts[l] for l in sums}
