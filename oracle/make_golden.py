"""Generate tests/golden/*.npz by running the REAL reference (imported, unmodified,
from /root/reference) on the deterministic recipe of oracle/recipe.py.

Runs only in the build container (the reference does not travel to the GPU box).
Usage:  python oracle/make_golden.py

What is stubbed so that the reference imports here:
  * wandb, swanlab, mxnet, cv2, torchvision, timm, IPython  -> MagicMock modules
  * loralib -> oracle/shims/loralib.py (restated third-party semantics)
  * torch.Tensor.cuda -> identity, torch.cuda.Stream/stream/current_stream inert,
    Tensor.record_stream no-op; GPU_ID=[0] (CosFace dereferences device_id[0]).
Fixtures hold OUTPUTS only; inputs/weights are rebuilt from the recipe by every consumer.
"""
import contextlib
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import recipe  # noqa: E402


def install_shims():
    for name in ["wandb", "swanlab", "mxnet", "mxnet.ndarray", "mxnet.io", "mxnet.recordio", "cv2",
                 "torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models",
                 "timm", "timm.scheduler", "timm.optim", "timm.models", "sklearn", "sklearn.model_selection",
                 "sklearn.decomposition", "matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image", "bcolz", "scipy",
                 "scipy.interpolate", "scipy.spatial", "scipy.spatial.distance"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = mock.MagicMock(name=name)
    if "IPython" not in sys.modules:
        ip = types.ModuleType("IPython")
        ip.version_info = (8, 30, 0)
        ip.get_ipython = lambda: None
        ip.embed = lambda *a, **k: None
        sys.modules["IPython"] = ip
    from oracle.shims import loralib as shim
    sys.modules["loralib"] = shim
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.record_stream = lambda self, *a, **k: None

    class _Stream:
        def wait_stream(self, *_):
            pass

    torch.cuda.Stream = _Stream
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    if REF not in sys.path:
        sys.path.insert(0, REF)


def build_reference_model(cfg, state_np, dropout=0.0):
    from vit_pytorch_face import ViT_face
    model = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=cfg["num_class"], image_size=cfg["image_size"],
                     patch_size=cfg["patch_size"], dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"],
                     mlp_dim=cfg["mlp_dim"], dropout=dropout, emb_dropout=dropout, lora_rank=cfg["lora_rank"],
                     lora_pos=cfg.get("lora_pos", "FFN"))
    sd = {k: torch.tensor(v) for k, v in state_np.items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    import loralib as lora
    lora.mark_only_lora_as_trainable(model)
    return model


class ListLoader:
    """Minimal stand-in for a DataLoader (the engine only iterates and calls len())."""

    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def run_case(tag, cfg, batch, out, hyper, n_steps=3):
    import engine_cl
    import engine as engine_single
    from util import utils as rutil
    from util.cal_norm import get_norm_of_lora

    state = recipe.make_state(cfg)
    gpos = cfg.get("lora_pos", "FFN")
    model = build_reference_model(cfg, state)
    assert [n for n, _ in model.named_parameters()] == list(recipe.param_shapes(cfg).keys()), "name order drift"

    order = list(range(cfg["num_class"]))
    n_forget = max(2, cfg["num_class"] // 5)
    # batches (remain labels in the first 80 %, forget labels in the last 20 % of classes)
    xs_r = [torch.tensor(recipe.make_images(cfg, batch, seed=100 + s, tag="xr")) for s in range(n_steps)]
    xs_f = [torch.tensor(recipe.make_images(cfg, batch, seed=200 + s, tag="xf")) for s in range(n_steps)]
    ys_r = [torch.tensor(recipe.make_labels(cfg, batch, seed=100 + s, tag="yr", lo=0, hi=cfg["num_class"] - n_forget))
            for s in range(n_steps)]
    ys_f = [torch.tensor(recipe.make_labels(cfg, batch, seed=200 + s, tag="yf", lo=cfg["num_class"] - n_forget,
                                            hi=cfg["num_class"])) for s in range(n_steps)]
    proto_np = recipe.make_prototypes(cfg)
    proto_dict = {c: torch.tensor(proto_np[c]) for c in range(cfg["num_class"])}

    res = {}
    # ---- forward (train mode, un-merged LoRA) ------------------------------------------
    model.train()
    with torch.no_grad():
        logits, emb = model(xs_r[0], ys_r[0])
        res["fwd_logits"], res["fwd_emb"] = logits.numpy(), emb.numpy()
        emb_only = model(xs_r[0])
        res["fwd_emb_nolabel"] = emb_only.numpy()
    # ---- eval mode (merged) --------------------------------------------------------------
    model.eval()
    with torch.no_grad():
        logits_e, emb_e = model(xs_r[0], ys_r[0])
        res["eval_logits"], res["eval_emb"] = logits_e.numpy(), emb_e.numpy()
        wkey = "transformer.layers.0.0.fn.fn.to_qkv.weight" if gpos == "Attention" else "transformer.layers.0.1.fn.fn.net.0.weight"
        res["merged_w_l0_net0"] = model.state_dict()[wkey].numpy().copy()
    model.train()
    with torch.no_grad():
        logits_rt, _ = model(xs_r[0], ys_r[0])
        res["roundtrip_logits"] = logits_rt.numpy()
    # restore exact weights (merge/un-merge drifts by ~1e-8)
    model.load_state_dict({k: torch.tensor(v) for k, v in state.items()})

    # ---- loss pieces -------------------------------------------------------------------------
    res["structure_loss"] = np.float32(engine_cl.get_structure_loss(model).item()) if cfg["depth"] == 6 else np.float32(-1)
    for gt in (("block",) if gpos == "Attention" else ("block", "lora", "matrix")):
        res[f"structure_loss_engine_{gt}"] = np.float32(
            engine_single.get_structure_loss(model, num_layers=cfg["depth"], group_type=gt, group_pos=gpos).item())
        res[f"cal_norm_{gt}"] = np.array(
            [float(v) for v in get_norm_of_lora(model, type="L2", group_num=cfg["depth"], group_type=gt, group_pos=gpos)],
            dtype=np.float32)
    with torch.no_grad():
        _, emb_f = model(xs_f[0], ys_f[0])
        res["proto_kl_f"] = np.float32(engine_cl.get_prototype_loss(emb_f, ys_f[0], proto_dict).item())
        res["proto_kl_r"] = np.float32(engine_cl.get_prototype_loss(emb, ys_r[0], proto_dict).item())

    # ---- full steps through the reference engine ------------------------------------------------
    if cfg["depth"] == 6:
        params = [p for p in model.parameters() if p.requires_grad]
        # timm.create_optimizer(args, model) -> torch.optim.AdamW(lr, weight_decay on 2-D params, eps 1e-8)
        opt = torch.optim.AdamW(params, lr=hyper["lr"], weight_decay=hyper["wd"], eps=1e-8, betas=(0.9, 0.999))
        crit = torch.nn.CrossEntropyLoss()
        mk = lambda: rutil.AverageMeter()
        meters = dict(losses_forget=mk(), losses_remain=mk(), losses_total=mk(), losses_structure=mk(),
                      top1_forget=mk(), top1_remain=mk(), losses_prototype_forget=mk(), losses_prototype_remain=mk())
        cfgd = {"DATA_ROOT": "./data/casia100/", "BND_pro": hyper["BND_pro"], "MULTI_GPU": False, "WORK_PATH": "/tmp",
                "BACKBONE_NAME": "VIT"}
        batch_ctr = 0
        for s in range(n_steps):
            ret = engine_cl.train_one_epoch(
                model=model, dataloader_forget=ListLoader([(xs_f[s], ys_f[s])]),
                dataloader_remain=ListLoader([(xs_r[s], ys_r[s])]), device=torch.device("cpu"), criterion=crit,
                optimizer=opt, epoch=0, beta=hyper["beta"], alpha=hyper["alpha"], BND=hyper["BND"], batch=batch_ctr,
                testloader_forget=None, testloader_remain=None, forget_acc_before=0.0, highest_H_mean=0.0, cfg=cfgd,
                task_i="0", use_prototype=True, prototype_dict=proto_dict,
                prototype_weight_forget=hyper["pro_f_weight"], prototype_weight_remain=hyper["pro_r_weight"], **meters)
            batch_ctr = ret[0]
            if s == 0:
                for n, p in model.named_parameters():
                    if p.requires_grad:
                        res[f"grad1::{n}"] = p.grad.numpy().copy()
                m = meters
                res["meters1"] = np.array([m["losses_forget"].val, m["losses_remain"].val, m["losses_total"].val,
                                           m["losses_structure"].val, m["top1_forget"].val, m["top1_remain"].val,
                                           m["losses_prototype_forget"].val, m["losses_prototype_remain"].val],
                                          dtype=np.float64)
            if s in (0, n_steps - 1):
                for n, p in model.named_parameters():
                    if p.requires_grad:
                        res[f"param{s + 1}::{n}"] = p.detach().numpy().copy()
        m = meters
        res[f"meters{n_steps}_avg"] = np.array([m["losses_forget"].avg, m["losses_remain"].avg, m["losses_total"].avg,
                                                m["losses_structure"].avg, m["top1_forget"].avg, m["top1_remain"].avg,
                                                m["losses_prototype_forget"].avg, m["losses_prototype_remain"].avg],
                                               dtype=np.float64)
        res["batch_ctr"] = np.int64(batch_ctr)
    else:
        # small models: engine_cl hard-codes 6 groups (engine_cl.py:388) -> take grads of the same loss by hand
        model.load_state_dict({k: torch.tensor(v) for k, v in state.items()})
        crit = torch.nn.CrossEntropyLoss()
        lo_r, em_r = model(xs_r[0], ys_r[0])
        lo_f, em_f = model(xs_f[0], ys_f[0])
        ce_r, ce_f = crit(lo_r, ys_r[0]), crit(lo_f, ys_f[0])
        sl = engine_single.get_structure_loss(model, num_layers=cfg["depth"], group_type="block", group_pos=gpos)
        kl_f = engine_cl.get_prototype_loss(em_f, ys_f[0], proto_dict)
        kl_r = engine_cl.get_prototype_loss(em_r, ys_r[0], proto_dict)
        pro = hyper["pro_f_weight"] * torch.relu(hyper["BND_pro"] - kl_f) + hyper["pro_r_weight"] * kl_r
        total = hyper["beta"] * torch.relu(hyper["BND"] - ce_f) + ce_r + hyper["alpha"] * sl + pro
        model.zero_grad()
        total.backward()
        for n, p in model.named_parameters():
            if p.requires_grad:
                res[f"grad1::{n}"] = p.grad.numpy().copy()
        res["losses1"] = np.array([ce_f.item(), ce_r.item(), total.item(), sl.item(), kl_f.item(), kl_r.item()],
                                  dtype=np.float64)
        # same batch with both hinges INACTIVE (BND below CE_f, BND_pro below KL_f)
        lo_r, em_r = model(xs_r[0], ys_r[0])
        lo_f, em_f = model(xs_f[0], ys_f[0])
        kl_f = engine_cl.get_prototype_loss(em_f, ys_f[0], proto_dict)
        kl_r = engine_cl.get_prototype_loss(em_r, ys_r[0], proto_dict)
        sl = engine_single.get_structure_loss(model, num_layers=cfg["depth"], group_type="block", group_pos=gpos)
        total = (hyper["beta"] * torch.relu(5.0 - crit(lo_f, ys_f[0])) + crit(lo_r, ys_r[0]) + hyper["alpha"] * sl
                 + hyper["pro_f_weight"] * torch.relu(0.1 - kl_f) + hyper["pro_r_weight"] * kl_r)
        model.zero_grad()
        total.backward()
        for n, p in model.named_parameters():
            if p.requires_grad:
                res[f"grad_inactive::{n}"] = p.grad.numpy().copy()
        res["total_inactive"] = np.float64(total.item())

    # ---- prototypes via the reference helper ---------------------------------------------------------------
    model.load_state_dict({k: torch.tensor(v) for k, v in state.items()})
    ds = torch.utils.data.TensorDataset(torch.cat([xs_r[0], xs_f[0]]), torch.cat([ys_r[0], ys_f[0]]))
    with mock.patch.object(rutil, "DataLoader", torch.utils.data.DataLoader):
        protos = rutil.calculate_prototypes(model, ds, batch_size=3, device="cpu")
    keys = sorted(protos.keys())
    res["proto_keys"] = np.array(keys, dtype=np.int64)
    res["proto_vals"] = np.stack([protos[k].numpy() for k in keys]).astype(np.float32)
    model.train()

    np.savez_compressed(os.path.join(out, f"{tag}.npz"), **res)
    print(f"[golden] {tag}: {len(res)} arrays,",
          f"{sum(v.nbytes for v in res.values()) / 1e6:.2f} MB raw")


HYPER = dict(lr=1e-2, wd=0.05, beta=0.15, alpha=1e-2, BND=105.0, BND_pro=2.0, pro_f_weight=0.05, pro_r_weight=0.1)


def host_kats(out):
    """Known answers for host-side logic taken from the reference / CPython."""
    import random
    order = list(range(100))
    random.seed(1337)
    random.shuffle(order)
    from util import utils as rutil
    m = rutil.AverageMeter()
    for v, n in [(1.5, 4), (2.5, 2), (-1.0, 10)]:
        m.update(v, n)
    logits = torch.tensor(recipe.uniform("kat_logits", (7, 10), 3))
    target = torch.tensor(recipe.make_labels({"num_class": 10}, 7, seed=3))
    acc = rutil.train_accuracy(logits, target, topk=(1,))
    # few-shot sampler (util/utils.py:457-499) on a synthetic label list: 12 classes x 9 samples, 4 shots, seed 2024
    class _DS(torch.utils.data.Dataset):
        def __init__(self):
            self.targets = [(7 * i + 3) % 12 for i in range(108)]
            self.classes = [f"c{i}" for i in range(12)]

        def __len__(self):
            return len(self.targets)

        def __getitem__(self, i):
            return torch.tensor([float(i)]), self.targets[i]
    fs = rutil.create_few_shot_dataset(_DS(), 4, seed=2024)
    np.savez(os.path.join(out, "host_kats.npz"), class_order=np.array(order), meter=np.array([m.val, m.avg, m.sum, m.count]),
             train_acc=np.float32(acc.item()), few_shot_indices=np.array(list(fs.indices), dtype=np.int64),
             few_shot_first=np.array([float(fs[0][0]), float(fs[0][1]), float(len(fs))]))
    print("[golden] host_kats")


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    only = sys.argv[1:]       # e.g. `python oracle/make_golden.py attn_small_b3` regenerates one fixture
    cases = [("small_b5", recipe.cfg_small(), 5), ("small2_b3", recipe.cfg_small2(), 3), ("full_b2", recipe.cfg_full(), 2),
             ("attn_small_b3", recipe.cfg_small_attn(), 3)]      # --lora_pos Attention (reference MergedLinear on to_qkv)
    for tag, cfg, b in cases:
        if not only or tag in only:
            run_case(tag, cfg, b, out, HYPER)
    if not only or "host_kats" in only:
        host_kats(out)


if __name__ == "__main__":
    main()
