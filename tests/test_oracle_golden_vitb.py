"""Pin the ViT-B/16-side oracle (oracle/tv_vit.py: restated adapter, head surgery, grouping, step) against the outputs of
the REAL reference adapter / helpers / engine (tests/golden/vitb_small*.npz from oracle/make_golden_vitb.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tv_vit as T

CASES = {"vitb_small_b4": (recipe.cfg_vitb_small(), 4), "vitb_small2_b3": (recipe.cfg_vitb_small2(), 3)}
HYPER = dict(lr=1e-2, wd=0.05, beta=0.15, alpha=1e-2, BND=8.0, BND_pro=2.0, pro_f_weight=0.05, pro_r_weight=0.1)


def head_map(cfg):
    n = cfg["num_class"]
    return {i: (7 * i + 3) % n for i in range(n // 2)}


def sub_state(cfg):
    """Recipe state with the classifier reduced as modify_head does."""
    st = recipe.make_tv_state(cfg)
    w, b = T.head_rows(torch.tensor(st["heads.head.weight"]), torch.tensor(st["heads.head.bias"]), head_map(cfg))
    st["heads.head.weight"], st["heads.head.bias"] = w.numpy(), b.numpy()
    return st, dict(cfg, num_class=len(head_map(cfg)))


def batches(sub, cfg, batch, s=0):
    n = sub["num_class"]
    nf = max(2, n // 5)
    return (torch.tensor(recipe.make_images(cfg, batch, seed=300 + s, tag="xr")),
            torch.tensor(recipe.make_labels(sub, batch, seed=300 + s, tag="yr", lo=0, hi=n - nf)),
            torch.tensor(recipe.make_images(cfg, batch, seed=400 + s, tag="xf")),
            torch.tensor(recipe.make_labels(sub, batch, seed=400 + s, tag="yf", lo=n - nf, hi=n)))


@pytest.mark.parametrize("tag", list(CASES))
def test_forward_and_head_surgery(tag, golden_dir):
    cfg, b = CASES[tag]
    g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    full = T.build(cfg, recipe.make_tv_state(cfg)).train()
    x0 = torch.tensor(recipe.make_images(cfg, b, seed=300, tag="xr"))
    with torch.no_grad():
        lo, em = full(x0, None)
    assert np.abs(lo.numpy() - g["fwd_logits_full"]).max() < 2e-5
    assert np.abs(em.numpy() - g["fwd_emb"]).max() < 2e-5
    st, sub = sub_state(cfg)
    assert np.array_equal(st["heads.head.weight"], g["head_w"]) and np.array_equal(st["heads.head.bias"], g["head_b"])
    assert np.array_equal(recipe.make_tv_state(cfg)["heads.head.weight"], g["resumed_head_w"])
    m = T.build(sub, st).train()
    xr, yr, _, _ = batches(sub, cfg, b)
    with torch.no_grad():
        assert np.abs(m(xr, yr)[0].numpy() - g["fwd_logits"]).max() < 2e-5
    m.eval()
    assert np.abs(m.state_dict()["encoder.layers.encoder_layer_0.mlp.0.weight"].numpy() - g["merged_w_l0_mlp0"]).max() < 1e-7
    with torch.no_grad():
        assert np.abs(m(xr, yr)[0].numpy() - g["eval_logits"]).max() < 2e-5


def test_groups_and_three_engine_steps(golden_dir):
    cfg, b = CASES["vitb_small_b4"]
    g = np.load(os.path.join(golden_dir, "vitb_small_b4.npz"))
    st, sub = sub_state(cfg)
    m = T.build(sub, st).train()
    assert abs(T.structure_loss(m).item() - float(g["structure_loss"])) < 1e-4
    assert np.abs(np.array([float(v) for v in T.cal_norm(m)]) - g["cal_norm"]).max() < 1e-5
    proto = torch.tensor(recipe.make_prototypes(sub))
    opt = None
    for s in range(3):
        xr, yr, xf, yf = batches(sub, cfg, b, s)
        losses, grads, opt = T.train_step(m, xr, yr, xf, yf, HYPER, opt_state=opt, step=s + 1, lr=HYPER["lr"], proto=proto)
        if s == 0:
            meters = np.array([HYPER["beta"] * float(losses["loss_forget"]), float(losses["ce_r"]), float(losses["total"]),
                               HYPER["alpha"] * float(losses["structure"]), float(losses["top1_f"]), float(losses["top1_r"]),
                               HYPER["pro_f_weight"] * max(0.0, HYPER["BND_pro"] - float(losses["kl_f"])),
                               HYPER["pro_r_weight"] * float(losses["kl_r"])])
            assert np.abs(meters - g["meters1"]).max() < 1e-4
            for k, v in grads.items():
                r = g[f"grad1::{k}"]
                assert np.abs(v.numpy() - r).max() < 2e-5 * max(1.0, np.abs(r).max()), k
        if s in (0, 2):
            for k, p in m.named_parameters():
                if p.requires_grad:
                    assert np.abs(p.detach().numpy() - g[f"param{s + 1}::{k}"]).max() < 5e-5, (s, k)


def test_grads_small2(golden_dir):
    cfg, b = CASES["vitb_small2_b3"]
    g = np.load(os.path.join(golden_dir, "vitb_small2_b3.npz"))
    st, sub = sub_state(cfg)
    m = T.build(sub, st).train()
    xr, yr, xf, yf = batches(sub, cfg, b)
    losses, grads, _ = T.train_step(m, xr, yr, xf, yf, HYPER, proto=torch.tensor(recipe.make_prototypes(sub)))
    got = [losses["ce_f"], losses["ce_r"], losses["total"], losses["structure"], losses["kl_f"], losses["kl_r"]]
    for a, r in zip(got, g["losses1"]):
        assert abs(float(a) - r) < 2e-5 * max(1.0, abs(r))
    for k, v in grads.items():
        r = g[f"grad1::{k}"]
        assert np.abs(v.numpy() - r).max() < 1e-5 * max(1.0, np.abs(r).max()), k


def test_vit_b16_geometry_known_answers():
    """Published numbers of torchvision vit_b_16: 86 567 656 parameters; the reference divides by 85 875 556 = the same
    network with a 100-way head (train_own_forget_cl.py:483-489)."""
    sh = recipe.tv_param_shapes(recipe.cfg_vitb(lora_rank=0, num_class=1000))
    assert sum(int(np.prod(s)) for s in sh.values()) == 86_567_656
    sh100 = recipe.tv_param_shapes(recipe.cfg_vitb(lora_rank=0, num_class=100))
    assert sum(int(np.prod(s)) for s in sh100.values()) == 85_875_556


def test_encoder_restatement_against_an_independent_second_restatement():
    """VERDICT r03 (missing #3): torchvision is installable in neither container, so the torchvision-0.15.1 encoder composition under
    oracle/tv_vit.py stays "parity unpinned" against torchvision's own code. What CAN be cross-checked is done here: an INDEPENDENT second
    restatement of the published architecture — explicit tensor algebra, no nn.MultiheadAttention / nn.LayerNorm / nn.Conv2d modules —
    reading the SAME state dict by torchvision's parameter names (packed `self_attention.in_proj_weight` [3d, d] in q | k | v row order,
    `in_proj_bias`, `out_proj`, `ln_1`/`ln_2` eps 1e-6, conv patch embedding as an unfold + matmul over (c, p1, p2), class token first,
    learned positions added before the blocks, final `encoder.ln`, `heads.head`) must reproduce tv_vit.VisionTransformer's logits to f64
    round-off. torch's nn.MultiheadAttention (which tv_vit uses, and which IS torchvision's operator) thereby pins the packed in_proj
    convention of the restatement the HIP path is tested against (gs-lora_amd/vit_pytorch_face/modified_VIT.py)."""
    import math
    cfg = recipe.cfg_vitb_small2()
    st = {k: torch.tensor(v).double() for k, v in recipe.make_tv_state(cfg).items() if "lora_" not in k}
    vit = T.VisionTransformer(cfg).double()
    missing = vit.load_state_dict(st, strict=True)
    x = torch.tensor(recipe.make_images(cfg, 3, seed=11, tag="x2")).double()
    with torch.no_grad():
        want = vit(x)

    def ln(t, w, b):
        mu = t.mean(-1, keepdim=True)
        var = ((t - mu) ** 2).mean(-1, keepdim=True)
        return (t - mu) / torch.sqrt(var + 1e-6) * w + b

    p, d, H = cfg["patch_size"], cfg["dim"], cfg["heads"]
    n, c, hh, ww = x.shape
    patches = x.reshape(n, c, hh // p, p, ww // p, p).permute(0, 2, 4, 1, 3, 5).reshape(n, (hh // p) * (ww // p), c * p * p)
    tok = patches @ st["conv_proj.weight"].reshape(d, -1).t() + st["conv_proj.bias"]
    seq = torch.cat([st["class_token"].expand(n, -1, -1), tok], dim=1) + st["encoder.pos_embedding"]
    for i in range(cfg["depth"]):
        pre = f"encoder.layers.encoder_layer_{i}."
        y = ln(seq, st[pre + "ln_1.weight"], st[pre + "ln_1.bias"])
        qkv = y @ st[pre + "self_attention.in_proj_weight"].t() + st[pre + "self_attention.in_proj_bias"]
        q, k, v = (t.reshape(n, -1, H, d // H).transpose(1, 2) for t in qkv.split(d, dim=-1))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d // H), dim=-1) @ v
        att = att.transpose(1, 2).reshape(n, -1, d) @ st[pre + "self_attention.out_proj.weight"].t() + st[pre + "self_attention.out_proj.bias"]
        seq = seq + att
        y = ln(seq, st[pre + "ln_2.weight"], st[pre + "ln_2.bias"])
        h1 = y @ st[pre + "mlp.0.weight"].t() + st[pre + "mlp.0.bias"]
        h1 = 0.5 * h1 * (1.0 + torch.erf(h1 / math.sqrt(2.0)))
        seq = seq + h1 @ st[pre + "mlp.3.weight"].t() + st[pre + "mlp.3.bias"]
    got = ln(seq, st["encoder.ln.weight"], st["encoder.ln.bias"])[:, 0] @ st["heads.head.weight"].t() + st["heads.head.bias"]
    assert (got - want).abs().max().item() < 1e-10 * max(1.0, want.abs().max().item())
