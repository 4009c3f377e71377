"""Pin the CPU oracle (oracle/gslora_oracle.py) against outputs of the real reference
(tests/golden/*.npz, produced by oracle/make_golden.py from /root/reference)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import gslora_oracle as O
from oracle import recipe

CASES = {"small_b5": (recipe.cfg_small(), 5), "small2_b3": (recipe.cfg_small2(), 3), "full_b2": (recipe.cfg_full(), 2),
         "attn_small_b3": (recipe.cfg_small_attn(), 3)}      # --lora_pos Attention
HYPER = dict(lr=1e-2, wd=0.05, beta=0.15, alpha=1e-2, BND=105.0, BND_pro=2.0, pro_f_weight=0.05, pro_r_weight=0.1)


def batches(cfg, batch, s=0):
    nf = max(2, cfg["num_class"] // 5)
    xr = torch.tensor(recipe.make_images(cfg, batch, seed=100 + s, tag="xr"))
    xf = torch.tensor(recipe.make_images(cfg, batch, seed=200 + s, tag="xf"))
    yr = torch.tensor(recipe.make_labels(cfg, batch, seed=100 + s, tag="yr", lo=0, hi=cfg["num_class"] - nf))
    yf = torch.tensor(recipe.make_labels(cfg, batch, seed=200 + s, tag="yf", lo=cfg["num_class"] - nf, hi=cfg["num_class"]))
    return xr, yr, xf, yf


@pytest.mark.parametrize("tag", list(CASES))
def test_forward_matches_reference(tag, golden_dir):
    cfg, b = CASES[tag]
    g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    st = O.to_torch(recipe.make_state(cfg))
    xr, yr, _, _ = batches(cfg, b)
    logits, emb = O.vit_forward(st, xr, yr, cfg)
    assert np.abs(logits.numpy() - g["fwd_logits"]).max() < 2e-5
    assert np.abs(emb.numpy() - g["fwd_emb"]).max() < 1e-5
    _, emb2 = O.vit_forward(st, xr, None, cfg)
    assert np.abs(emb2.numpy() - g["fwd_emb_nolabel"]).max() < 1e-5
    # eval mode == merged weights
    stm = O.merge_lora(st, cfg)
    wkey = "transformer.layers.0.0.fn.fn.to_qkv.weight" if cfg.get("lora_pos") == "Attention" else "transformer.layers.0.1.fn.fn.net.0.weight"
    assert np.abs(stm[wkey].numpy() - g["merged_w_l0_net0"]).max() < 1e-7
    le, ee = O.vit_forward(stm, xr, yr, cfg, merged=True)
    assert np.abs(le.numpy() - g["eval_logits"]).max() < 2e-5
    assert np.abs(ee.numpy() - g["eval_emb"]).max() < 1e-5


@pytest.mark.parametrize("tag", list(CASES))
def test_loss_pieces_match_reference(tag, golden_dir):
    cfg, b = CASES[tag]
    g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    st = O.to_torch(recipe.make_state(cfg))
    for gt in (("block",) if cfg.get("lora_pos") == "Attention" else ("block", "lora", "matrix")):
        assert abs(O.structure_loss(st, cfg, gt).item() - float(g[f"structure_loss_engine_{gt}"])) < 1e-4
        assert np.abs(O.cal_norm_of_lora(st, cfg, gt).numpy() - g[f"cal_norm_{gt}"]).max() < 1e-5
    if cfg["depth"] == 6:
        assert abs(O.structure_loss(st, cfg).item() - float(g["structure_loss"])) < 1e-4
    xr, yr, xf, yf = batches(cfg, b)
    proto = torch.tensor(recipe.make_prototypes(cfg))
    _, er = O.vit_forward(st, xr, yr, cfg)
    _, ef = O.vit_forward(st, xf, yf, cfg)
    assert abs(O.prototype_kl(ef, yf, proto).item() - float(g["proto_kl_f"])) < 1e-5
    assert abs(O.prototype_kl(er, yr, proto).item() - float(g["proto_kl_r"])) < 1e-5


@pytest.mark.parametrize("tag", ["small_b5", "small2_b3", "attn_small_b3"])
def test_grads_small_match_reference(tag, golden_dir):
    cfg, b = CASES[tag]
    g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    st_np = recipe.make_state(cfg)
    xr, yr, xf, yf = batches(cfg, b)
    proto = torch.tensor(recipe.make_prototypes(cfg))
    losses, grads, _, _ = O.train_step(st_np, cfg, xr, yr, xf, yf, HYPER, proto=proto)
    ref = g["losses1"]
    got = [losses["ce_f"], losses["ce_r"], losses["total"], losses["structure"], losses["kl_f"], losses["kl_r"]]
    for a, r in zip(got, ref):
        assert abs(float(a) - r) < 2e-5 * max(1.0, abs(r))
    for k, v in grads.items():
        r = g[f"grad1::{k}"]
        assert np.abs(v.numpy() - r).max() < 1e-5 * max(1.0, np.abs(r).max()), k
    hy2 = dict(HYPER, BND=5.0, BND_pro=0.1)
    losses, grads, _, _ = O.train_step(st_np, cfg, xr, yr, xf, yf, hy2, proto=proto)
    assert abs(float(losses["total"]) - float(g["total_inactive"])) < 1e-4
    for k, v in grads.items():
        r = g[f"grad_inactive::{k}"]
        assert np.abs(v.numpy() - r).max() < 1e-5 * max(1.0, np.abs(r).max()), k


def test_full_three_steps_match_reference_engine(golden_dir):
    """Oracle step == engine_cl.train_one_epoch + torch.optim.AdamW of the reference, 3 steps."""
    cfg, b = CASES["full_b2"]
    g = np.load(os.path.join(golden_dir, "full_b2.npz"))
    st_np = recipe.make_state(cfg)
    proto = torch.tensor(recipe.make_prototypes(cfg))
    opt = None
    sums = np.zeros(8)
    for s in range(3):
        xr, yr, xf, yf = batches(cfg, b, s)
        losses, grads, new_st, opt = O.train_step(st_np, cfg, xr, yr, xf, yf, HYPER, opt_state=opt, step=s + 1,
                                                  lr=HYPER["lr"], proto=proto)
        meters = np.array([HYPER["beta"] * float(losses["loss_forget"]), float(losses["ce_r"]), float(losses["total"]),
                           HYPER["alpha"] * float(losses["structure"]), float(losses["top1_f"]), float(losses["top1_r"]),
                           HYPER["pro_f_weight"] * max(0.0, HYPER["BND_pro"] - float(losses["kl_f"])),
                           HYPER["pro_r_weight"] * float(losses["kl_r"])])
        sums += meters
        if s == 0:
            assert np.abs(meters - g["meters1"]).max() < 1e-4
            for k, v in grads.items():
                r = g[f"grad1::{k}"]
                assert np.abs(v.numpy() - r).max() < 2e-5 * max(1.0, np.abs(r).max()), k
        if s in (0, 2):
            for k in grads:
                r = g[f"param{s + 1}::{k}"]
                assert np.abs(new_st[k].numpy() - r).max() < 5e-5, (s, k)   # Adam's m/sqrt(v) amplifies 1e-7 grad noise
        st_np = {k: v.numpy() for k, v in new_st.items()}
    assert np.abs(sums / 3 - g["meters3_avg"]).max() < 1e-3
    assert int(g["batch_ctr"]) == 3


def test_prototypes_match_reference(golden_dir):
    for tag, (cfg, b) in CASES.items():
        g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
        st = O.to_torch(recipe.make_state(cfg))
        xr, yr, xf, yf = batches(cfg, b)
        protos = O.calculate_prototypes(st, cfg, torch.cat([xr, xf]), torch.cat([yr, yf]), batch_size=3)
        assert sorted(protos) == list(g["proto_keys"])
        got = np.stack([protos[k].numpy() for k in sorted(protos)])
        assert np.abs(got - g["proto_vals"]).max() < 1e-5


def test_host_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "host_kats.npz"))
    assert O.class_order() == list(g["class_order"])
    assert O.class_order()[:5] == [83, 17, 10, 9, 89]          # SURVEY.md §8c KAT
    # cosine lr KATs
    for e, lr in [(0, 1.0e-2), (1, 9.99754e-3), (25, 8.53700e-3), (50, 5.005e-3), (75, 1.47300e-3), (99, 1.24647e-5)]:
        assert abs(O.cosine_lr(e) - lr) < 2e-8 + 1e-5 * lr
    # AdamW against torch.optim.AdamW
    p = torch.tensor(recipe.uniform("p", (8, 16), 1)); gr = torch.tensor(recipe.uniform("g", (8, 16), 2))
    pt = torch.nn.Parameter(p.clone()); opt = torch.optim.AdamW([pt], lr=1e-2, weight_decay=0.05, eps=1e-8)
    m = torch.zeros_like(p); v = torch.zeros_like(p); po = p.clone()
    for step in (1, 2, 3):
        pt.grad = gr.clone() * step
        opt.step()
        po, m, v = O.adamw_update(po, gr * step, m, v, step, 1e-2)
        assert (po - pt.detach()).abs().max() < 1e-6
    assert abs(O.reinit_bound(512) - math.sqrt(6.0 / (51 * 512))) < 1e-12
    # fresh-model structure-loss KAT: 6*sqrt(16/3) within sampling noise
    cfg = recipe.cfg_full()
    st = O.to_torch(recipe.make_state(cfg, lora_b_std=0.0))
    assert abs(O.structure_loss(st, cfg).item() - 6 * math.sqrt(16 / 3)) < 0.2
    assert O.group_mask(O.group_lasso_norms(st, cfg)).all()
