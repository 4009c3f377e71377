"""Generate tests/golden/vitb_small*.npz by running the REAL reference adapter / helpers / engine of the ViT-B/16
ImageNet100 path (imported unmodified from /root/reference):
    vit_pytorch_face.ModifiedViT            (modified_VIT.py:5-45)
    util.utils.replace_ffn_with_lora / modify_head / resume_head   (utils.py:552-636)
    engine_cl.get_structure_loss(imagenet=True), engine_cl.train_one_epoch with cfg DATA_ROOT="./data/imagenet100/"
    util.cal_norm.get_norm_of_lora(imagenet=True)
on top of oracle/tv_vit.VisionTransformer (the restated torchvision==0.15.1 backbone — torchvision itself is not installed,
see oracle/tv_vit.py header). Runs only in the build container. Usage:  python oracle/make_golden_vitb.py
Fixtures hold OUTPUTS only; inputs / weights are rebuilt from oracle/recipe.py by every consumer.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import recipe  # noqa: E402
from oracle.make_golden import HYPER, ListLoader, install_shims  # noqa: E402

HYPER_TV = dict(HYPER, BND=8.0)      # plain-Linear logits: CE_f ~ 3..6, so BND 8 keeps the forget hinge active


def head_map(cfg):
    """current id -> original id, deliberately unordered (modify_head keeps dict order)."""
    n = cfg["num_class"]
    picks = [(7 * i + 3) % n for i in range(n // 2)]
    return {i: o for i, o in enumerate(picks)}


def run_case(tag, cfg, batch, out, n_steps=3):
    import engine_cl
    from util import utils as rutil
    from util.cal_norm import get_norm_of_lora
    from vit_pytorch_face import ModifiedViT
    import loralib as lora
    from oracle import tv_vit

    state = recipe.make_tv_state(cfg)
    model = rutil.replace_ffn_with_lora(ModifiedViT(tv_vit.VisionTransformer(cfg)), rank=cfg["lora_rank"])
    assert [n for n, _ in model.named_parameters()] == list(recipe.tv_param_shapes(cfg).keys()), "name order drift"
    model.load_state_dict({k: torch.tensor(v) for k, v in state.items()}, strict=True)
    res = {}
    x0 = torch.tensor(recipe.make_images(cfg, batch, seed=300, tag="xr"))
    model.train()
    with torch.no_grad():
        lo, em = model(x0, None)
        res["fwd_logits_full"], res["fwd_emb"] = lo.numpy(), em.numpy()

    # ---- head surgery (chdir: the reference writes results/original_VIT_head/classifier.pth relative to cwd)
    cmap = head_map(cfg)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            model = rutil.modify_head(model, current_id_to_original_id=cmap, device="cpu")
            resumed = rutil.resume_head(model, device="cpu")
        finally:
            os.chdir(cwd)
    res["head_w"], res["head_b"] = model.heads.head.weight.detach().numpy().copy(), model.heads.head.bias.detach().numpy().copy()
    res["resumed_head_w"] = resumed.heads.head.weight.detach().numpy().copy()
    lora.mark_only_lora_as_trainable(model)
    ncls = len(cmap)
    nf = max(2, ncls // 5)
    sub = dict(cfg, num_class=ncls)
    xs_r = [torch.tensor(recipe.make_images(cfg, batch, seed=300 + s, tag="xr")) for s in range(n_steps)]
    xs_f = [torch.tensor(recipe.make_images(cfg, batch, seed=400 + s, tag="xf")) for s in range(n_steps)]
    ys_r = [torch.tensor(recipe.make_labels(sub, batch, seed=300 + s, tag="yr", lo=0, hi=ncls - nf)) for s in range(n_steps)]
    ys_f = [torch.tensor(recipe.make_labels(sub, batch, seed=400 + s, tag="yf", lo=ncls - nf, hi=ncls)) for s in range(n_steps)]
    proto_np = recipe.make_prototypes(sub)
    proto_dict = {c: torch.tensor(proto_np[c]) for c in range(ncls)}

    model.train()
    with torch.no_grad():
        lo, em = model(xs_r[0], ys_r[0])
        res["fwd_logits"] = lo.numpy()
    model.eval()
    with torch.no_grad():
        le, _ = model(xs_r[0], ys_r[0])
        res["eval_logits"] = le.numpy()
        res["merged_w_l0_mlp0"] = model.state_dict()["encoder.layers.encoder_layer_0.mlp.0.weight"].numpy().copy()
    model.train()
    base = {k: v.clone() for k, v in model.state_dict().items()}

    if cfg["depth"] == 12:      # the reference hard-codes 12 groups for imagenet
        res["structure_loss"] = np.float32(engine_cl.get_structure_loss(model, imagenet=True).item())
        res["cal_norm"] = np.array([float(v) for v in get_norm_of_lora(model, type="L2", imagenet=True)], dtype=np.float32)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=HYPER_TV["lr"], weight_decay=HYPER_TV["wd"], eps=1e-8, betas=(0.9, 0.999))
        crit = torch.nn.CrossEntropyLoss()
        mk = lambda: rutil.AverageMeter()
        meters = dict(losses_forget=mk(), losses_remain=mk(), losses_total=mk(), losses_structure=mk(), top1_forget=mk(),
                      top1_remain=mk(), losses_prototype_forget=mk(), losses_prototype_remain=mk())
        cfgd = {"DATA_ROOT": "./data/imagenet100/", "BND_pro": HYPER_TV["BND_pro"], "MULTI_GPU": False, "WORK_PATH": "/tmp",
                "BACKBONE_NAME": "VIT_B16"}
        ctr = 0
        for s in range(n_steps):
            ret = engine_cl.train_one_epoch(
                model=model, dataloader_forget=ListLoader([(xs_f[s], ys_f[s])]), dataloader_remain=ListLoader([(xs_r[s], ys_r[s])]),
                device=torch.device("cpu"), criterion=crit, optimizer=opt, epoch=0, beta=HYPER_TV["beta"], alpha=HYPER_TV["alpha"],
                BND=HYPER_TV["BND"], batch=ctr, testloader_forget=None, testloader_remain=None, forget_acc_before=0.0,
                highest_H_mean=0.0, cfg=cfgd, task_i="0", use_prototype=True, prototype_dict=proto_dict,
                prototype_weight_forget=HYPER_TV["pro_f_weight"], prototype_weight_remain=HYPER_TV["pro_r_weight"], **meters)
            ctr = ret[0]
            if s == 0:
                for n, p in model.named_parameters():
                    if p.requires_grad:
                        res[f"grad1::{n}"] = p.grad.numpy().copy()
                m = meters
                res["meters1"] = np.array([m["losses_forget"].val, m["losses_remain"].val, m["losses_total"].val,
                                           m["losses_structure"].val, m["top1_forget"].val, m["top1_remain"].val,
                                           m["losses_prototype_forget"].val, m["losses_prototype_remain"].val], dtype=np.float64)
            if s in (0, n_steps - 1):
                for n, p in model.named_parameters():
                    if p.requires_grad:
                        res[f"param{s + 1}::{n}"] = p.detach().numpy().copy()
    else:                       # other depths: gradients of the same loss taken by hand through the real adapter
        crit = torch.nn.CrossEntropyLoss()
        lo_r, em_r = model(xs_r[0], ys_r[0])
        lo_f, em_f = model(xs_f[0], ys_f[0])
        ce_r, ce_f = crit(lo_r, ys_r[0]), crit(lo_f, ys_f[0])
        from oracle import tv_vit as T
        sl = T.structure_loss(model)
        kl_f = engine_cl.get_prototype_loss(em_f, ys_f[0], proto_dict)
        kl_r = engine_cl.get_prototype_loss(em_r, ys_r[0], proto_dict)
        pro = HYPER_TV["pro_f_weight"] * torch.relu(HYPER_TV["BND_pro"] - kl_f) + HYPER_TV["pro_r_weight"] * kl_r
        total = HYPER_TV["beta"] * torch.relu(HYPER_TV["BND"] - ce_f) + ce_r + HYPER_TV["alpha"] * sl + pro
        model.zero_grad()
        total.backward()
        for n, p in model.named_parameters():
            if p.requires_grad:
                res[f"grad1::{n}"] = p.grad.numpy().copy()
        res["losses1"] = np.array([ce_f.item(), ce_r.item(), total.item(), sl.item(), kl_f.item(), kl_r.item()], dtype=np.float64)
    del base
    np.savez_compressed(os.path.join(out, f"{tag}.npz"), **res)
    print(f"[golden] {tag}: {len(res)} arrays, {sum(v.nbytes for v in res.values()) / 1e6:.2f} MB raw")


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = os.path.join(ROOT, "tests", "golden")
    run_case("vitb_small_b4", recipe.cfg_vitb_small(), 4, out)
    run_case("vitb_small2_b3", recipe.cfg_vitb_small2(), 3, out)


if __name__ == "__main__":
    main()
