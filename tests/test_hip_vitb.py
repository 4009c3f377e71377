"""ViT-B/16 (ModifiedViT) path on the HIP kernels against (a) the golden vectors of the real reference adapter/engine and
(b) the CPU oracle (oracle/tv_vit.py) on the same seeded inputs. Tolerances: f32 mode 1e-4 absolute on logits / group
norms; bf16 mode stated per assertion."""
import os

import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tv_vit as T

pytestmark = pytest.mark.gpu

CASES = {"vitb_small_b4": (recipe.cfg_vitb_small(), 4), "vitb_small2_b3": (recipe.cfg_vitb_small2(), 3)}
HYPER = dict(lr=1e-2, wd=0.05, beta=0.15, alpha=1e-2, BND=8.0, BND_pro=2.0, pro_f_weight=0.05, pro_r_weight=0.1)


def head_map(cfg):
    n = cfg["num_class"]
    return {i: (7 * i + 3) % n for i in range(n // 2)}


def build_full(cfg, dtype="fp32"):
    """The driver's construction order (train_own_forget_cl.py:238-262): vit -> ModifiedViT -> LoRA -> load checkpoint."""
    from util.utils import replace_ffn_with_lora
    from vit_pytorch_face import ModifiedViT
    from vit_pytorch_face.modified_VIT import vit_b_16
    vit = vit_b_16(image_size=cfg["image_size"], patch_size=cfg["patch_size"], num_layers=cfg["depth"], num_heads=cfg["heads"],
                   hidden_dim=cfg["dim"], mlp_dim=cfg["mlp_dim"], num_classes=cfg["num_class"])
    m = replace_ffn_with_lora(ModifiedViT(vit), rank=cfg["lora_rank"])
    assert [n for n, _ in m.named_parameters()] == list(recipe.tv_param_shapes(cfg))
    m.load_state_dict({k: torch.tensor(v) for k, v in recipe.make_tv_state(cfg).items()}, strict=True)
    return m.to("cuda").set_compute_dtype(dtype)


def build_sub(cfg, dtype="fp32", tmp=None):
    import loralib as lora
    from util import utils as U
    m = build_full(cfg, dtype)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        m2 = U.modify_head(m, current_id_to_original_id=head_map(cfg), device="cuda")
        resumed = U.resume_head(m2, device="cuda")
    finally:
        os.chdir(cwd)
    lora.mark_only_lora_as_trainable(m2)
    m2.set_compute_dtype(dtype)
    return m2, resumed, dict(cfg, num_class=len(head_map(cfg)))


def batches(sub, cfg, batch, s=0):
    n = sub["num_class"]
    nf = max(2, n // 5)
    mk = lambda a: torch.tensor(a).cuda()
    return (mk(recipe.make_images(cfg, batch, seed=300 + s, tag="xr")), mk(recipe.make_labels(sub, batch, seed=300 + s, tag="yr", lo=0, hi=n - nf)),
            mk(recipe.make_images(cfg, batch, seed=400 + s, tag="xf")), mk(recipe.make_labels(sub, batch, seed=400 + s, tag="yf", lo=n - nf, hi=n)))


@pytest.mark.parametrize("tag", list(CASES))
def test_forward_f32_and_head_surgery_match_reference(tag, golden_dir, tmp_path):
    cfg, b = CASES[tag]
    g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    full = build_full(cfg).train()
    x0 = torch.tensor(recipe.make_images(cfg, b, seed=300, tag="xr")).cuda()
    with torch.no_grad():
        lo, em = full(x0, None)
    assert np.abs(lo.cpu().numpy() - g["fwd_logits_full"]).max() < 1e-4
    assert np.abs(em.cpu().numpy() - g["fwd_emb"]).max() < 1e-4
    m, resumed, sub = build_sub(cfg, tmp=tmp_path)
    assert np.array_equal(m.heads.head.weight.detach().cpu().numpy(), g["head_w"])
    assert np.array_equal(m.heads.head.bias.detach().cpu().numpy(), g["head_b"])
    assert np.array_equal(resumed.heads.head.weight.detach().cpu().numpy(), g["resumed_head_w"])
    xr, yr, _, _ = batches(sub, cfg, b)
    m.train()
    with torch.no_grad():
        assert np.abs(m(xr, yr)[0].cpu().numpy() - g["fwd_logits"]).max() < 1e-4
    m.eval()      # loralib merge semantics
    assert np.abs(m.state_dict()["encoder.layers.encoder_layer_0.mlp.0.weight"].cpu().numpy() - g["merged_w_l0_mlp0"]).max() < 1e-6
    with torch.no_grad():
        assert np.abs(m(xr, yr)[0].cpu().numpy() - g["eval_logits"]).max() < 1e-4
    m.train()
    with torch.no_grad():
        assert np.abs(m(xr, yr)[0].cpu().numpy() - g["fwd_logits"]).max() < 1e-4
        # the restored 1000-way head runs through the same kernels
        assert np.abs(resumed.set_compute_dtype("fp32").train()(x0, None)[0].cpu().numpy() - g["fwd_logits_full"]).max() < 1e-4


@pytest.mark.parametrize("tag", list(CASES))
def test_forward_bf16_close(tag, golden_dir, tmp_path):
    cfg, b = CASES[tag]
    g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    full = build_full(cfg, "bf16").train()
    x0 = torch.tensor(recipe.make_images(cfg, b, seed=300, tag="xr")).cuda()
    with torch.no_grad():
        lo, em = full(x0, None)
    # bf16 operands through 12 (3) blocks; embeddings are LayerNorm outputs of O(1), logits O(3)
    assert np.abs(em.cpu().numpy() - g["fwd_emb"]).max() < 0.06
    assert np.abs(lo.cpu().numpy() - g["fwd_logits_full"]).max() < 0.12


def test_structure_loss_cal_norm_and_engine_steps_f32(golden_dir, tmp_path):
    """3 steps of engine_cl.train_one_epoch with cfg DATA_ROOT == imagenet100 (12 hard-coded groups in the reference)."""
    import engine_cl
    from gslora_hip.optim import FusedAdamW
    from util.cal_norm import get_norm_of_lora
    from util.utils import AverageMeter
    cfg, b = CASES["vitb_small_b4"]
    g = np.load(os.path.join(golden_dir, "vitb_small_b4.npz"))
    m, _, sub = build_sub(cfg, tmp=tmp_path)
    m.train()
    assert abs(engine_cl.get_structure_loss(m, imagenet=True).item() - float(g["structure_loss"])) < 1e-4
    with pytest.raises(ValueError):
        engine_cl.get_structure_loss(m, imagenet=False)
    cn = np.array([float(v) for v in get_norm_of_lora(m, type="L2", imagenet=True)])
    assert cn.shape == (12,) and np.abs(cn - g["cal_norm"]).max() < 1e-4
    proto_np = recipe.make_prototypes(sub)
    proto = {c: torch.tensor(proto_np[c]) for c in range(sub["num_class"])}
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=HYPER["lr"], weight_decay=HYPER["wd"], eps=1e-8)
    crit = torch.nn.CrossEntropyLoss()
    cfgd = {"DATA_ROOT": "./data/imagenet100/", "BND_pro": HYPER["BND_pro"], "MULTI_GPU": False, "WORK_PATH": str(tmp_path),
            "BACKBONE_NAME": "VIT_B16"}
    # oracle replica stepping alongside (reference params only where the gradient is well conditioned)
    om = T.build(sub, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}).train()
    oopt = None
    ctr = 0
    for s in range(3):
        xr, yr, xf, yf = batches(sub, cfg, b, s)
        mk = AverageMeter
        meters = dict(losses_forget=mk(), losses_remain=mk(), losses_total=mk(), losses_structure=mk(), top1_forget=mk(),
                      top1_remain=mk(), losses_prototype_forget=mk(), losses_prototype_remain=mk())
        ret = engine_cl.train_one_epoch(
            model=m, dataloader_forget=[(xf, yf)], dataloader_remain=[(xr, yr)], device=torch.device("cuda"), criterion=crit,
            optimizer=opt, epoch=0, beta=HYPER["beta"], alpha=HYPER["alpha"], BND=HYPER["BND"], batch=ctr, testloader_forget=None,
            testloader_remain=None, forget_acc_before=0.0, highest_H_mean=0.0, cfg=cfgd, task_i="0", use_prototype=True,
            prototype_dict=proto, prototype_weight_forget=HYPER["pro_f_weight"], prototype_weight_remain=HYPER["pro_r_weight"], **meters)
        ctr = ret[0]
        got = np.array([ret[2].val, ret[3].val, ret[6].val, ret[7].val, ret[4].val, ret[5].val, ret[8].val, ret[9].val])
        grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.requires_grad}
        if s == 0:
            assert np.abs(got - g["meters1"]).max() < 1e-4
            for k, v in grads.items():
                r = g[f"grad1::{k}"]
                assert np.abs(v - r).max() < 1e-4 * max(1.0, np.abs(r).max()), k
        # HIP AdamW == oracle AdamW on the HIP gradients
        from oracle.gslora_oracle import adamw_update
        oopt = oopt or {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in om.named_parameters() if p.requires_grad}
        with torch.no_grad():
            for n, p in om.named_parameters():
                if p.requires_grad:
                    q, mm, vv = adamw_update(p.detach(), torch.tensor(grads[n]), *oopt[n], s + 1, HYPER["lr"], HYPER["wd"])
                    p.copy_(q)
                    oopt[n] = (mm, vv)
        for n, p in m.named_parameters():
            if p.requires_grad:
                assert (p.detach().cpu() - om.get_parameter(n).detach()).abs().max() < 2e-6, (s, n)
        if s in (0, 2):
            for n, p in m.named_parameters():
                if p.requires_grad:
                    r, g1 = g[f"param{s + 1}::{n}"], np.abs(g[f"grad1::{n}"])
                    ok = g1 > 1e-4 if s == 0 else np.ones_like(g1, dtype=bool)
                    tol = 5e-5 if s == 0 else 2e-3      # Adam's m/sqrt(v) amplifies 1e-7 gradient noise; 3 steps of lr 1e-2
                    assert np.abs(p.detach().cpu().numpy() - r)[ok].max() < tol, (s, n)
    assert ctr == 3


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_grads_small2_vs_reference(dtype, tol, golden_dir, tmp_path):
    """rank 16, 2 heads, 37 tokens; loss composed with the public API; gradients against the real reference's autograd."""
    import engine_cl
    from gslora_hip import losses
    cfg, b = CASES["vitb_small2_b3"]
    g = np.load(os.path.join(golden_dir, "vitb_small2_b3.npz"))
    m, _, sub = build_sub(cfg, dtype, tmp=tmp_path)
    m.train()
    xr, yr, xf, yf = batches(sub, cfg, b)
    proto_np = recipe.make_prototypes(sub)
    proto = {c: torch.tensor(proto_np[c]) for c in range(sub["num_class"])}
    lo_r, em_r = m(xr, yr)
    lo_f, em_f = m(xf, yf)
    ce_r = losses.ce_sum_top1(lo_r, yr)[0] / b
    ce_f = losses.ce_sum_top1(lo_f, yf)[0] / b
    sl = losses.structure_loss(m, "block")
    kl_f, kl_r = engine_cl.get_prototype_loss(em_f, yf, proto), engine_cl.get_prototype_loss(em_r, yr, proto)
    total = (HYPER["beta"] * torch.relu(HYPER["BND"] - ce_f) + ce_r + HYPER["alpha"] * sl
             + HYPER["pro_f_weight"] * torch.relu(HYPER["BND_pro"] - kl_f) + HYPER["pro_r_weight"] * kl_r)
    total.backward()
    got = [ce_f.item(), ce_r.item(), total.item(), sl.item(), kl_f.item(), kl_r.item()]
    ltol = 1e-4 if dtype == "fp32" else 5e-2
    for a, r in zip(got, g["losses1"]):
        assert abs(a - r) < ltol * max(1.0, abs(r)), (got, g["losses1"])
    for n, p in m.named_parameters():
        if p.requires_grad:
            r = g[f"grad1::{n}"]
            assert np.abs(p.grad.cpu().numpy() - r).max() < tol * max(1e-2 if dtype == "bf16" else 1.0, np.abs(r).max()) + (0 if dtype == "fp32" else 2e-3), n


def test_vit_b16_full_geometry_bf16_step_runs():
    """Real ViT-B/16 shapes (224 px, 197 tokens, dim 768, mlp 3072, r 16), batch 4+4: one fused step, finite, all 12 groups get gradient."""
    import loralib as lora
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import gs_lora_step
    from util.utils import replace_ffn_with_lora
    from vit_pytorch_face import ModifiedViT
    from vit_pytorch_face.modified_VIT import vit_b_16
    torch.manual_seed(0)
    m = replace_ffn_with_lora(ModifiedViT(vit_b_16(num_classes=100)), rank=16)
    assert sum(p.numel() for n, p in m.named_parameters() if "lora_" not in n) == 85_875_556
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("lora_B"):
                p.normal_(0, 0.02)
        m.heads.head.weight.normal_(0, 0.02)
    lora.mark_only_lora_as_trainable(m)
    m = m.to("cuda").set_compute_dtype("bf16").train()
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.05, eps=1e-8)
    x = torch.rand(8, 3, 224, 224, device="cuda")
    y = torch.randint(0, 100, (8,), device="cuda")
    before = m.lora_bucket().flat.clone()
    pack = gs_lora_step(m, opt, torch.nn.CrossEntropyLoss(), x[:4], y[:4], x[4:], y[4:], beta=0.15, alpha=0.01, BND=110.0)
    vals = pack.tolist()
    assert all(np.isfinite(v) for v in vals), vals
    bucket = m.lora_bucket()
    assert bucket.ngroups_block == 12
    gn = torch.stack([torch.cat([gv.reshape(-1) for gv in bucket.grad_views[4 * i:4 * i + 4]]).norm() for i in range(12)])
    assert torch.isfinite(gn).all() and (gn > 0).all()
    assert (bucket.flat - before).abs().max() > 0


# ---- BASELINE config 4 at its REAL geometry: ViT-B/16 224 px, 197 tokens, dim 768, 12 heads, mlp 3072, rank 16 —
# the K = 768 / 3072 GEMMs, the r = 16 K segment, LN eps 1e-6 and the head_dim^-0.5 scale that the shrunken goldens above do not reach.
# Checker: the CPU oracle (oracle/tv_vit.py, the restatement of torchvision 0.15.1's encoder + the reference's adapter) on the box's host
# cores, same seeded weights (recipe.make_tv_state) and images; 2 + 2 images keep it to seconds.
_FULL = {}


def _full_geometry_case():
    if not _FULL:
        import loralib as lora
        cfg = recipe.cfg_vitb(lora_rank=16, num_class=100)
        st = recipe.make_tv_state(cfg)
        om = T.build(cfg, st).train()
        assert sum(p.numel() for n, p in om.named_parameters() if "lora_" not in n) == 85_875_556      # (100-way head)
        b = 2
        mk = lambda a: torch.tensor(a)
        xr, xf = mk(recipe.make_images(cfg, b, seed=300, tag="xr")), mk(recipe.make_images(cfg, b, seed=400, tag="xf"))
        yr, yf = mk(recipe.make_labels(cfg, b, seed=300, tag="yr", lo=0, hi=80)), mk(recipe.make_labels(cfg, b, seed=400, tag="yf", lo=80, hi=100))
        proto_np = recipe.make_prototypes(cfg)
        proto = torch.tensor(np.stack([proto_np[c] for c in range(cfg["num_class"])]))
        hyper = dict(HYPER, BND=8.0)
        out = T.step_losses(om, xr, yr, xf, yf, hyper, proto)
        named = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        gs = torch.autograd.grad(out["total"], [p for _, p in named])
        _FULL.update(cfg=cfg, st=st, b=b, x=(xr, yr, xf, yf), proto_np=proto_np, hyper=hyper, out={k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()},
                     grads={n: g for (n, _), g in zip(named, gs)}, cal_norm=[float(v) for v in T.cal_norm(om)])
    return _FULL


def _full_geometry_hip(dtype):
    import engine_cl
    import loralib as lora
    from gslora_hip import losses
    c = _full_geometry_case()
    cfg, b = c["cfg"], c["b"]
    m = build_full(cfg, dtype)
    lora.mark_only_lora_as_trainable(m)
    m.train()
    xr, yr, xf, yf = (t.cuda() for t in c["x"])
    proto = {k: torch.tensor(c["proto_np"][k]) for k in range(cfg["num_class"])}
    lo_r, em_r = m(xr, yr)
    lo_f, em_f = m(xf, yf)
    ce_r = losses.ce_sum_top1(lo_r, yr)[0] / b
    ce_f = losses.ce_sum_top1(lo_f, yf)[0] / b
    sl = engine_cl.get_structure_loss(m, imagenet=True)
    kl_f, kl_r = engine_cl.get_prototype_loss(em_f, yf, proto), engine_cl.get_prototype_loss(em_r, yr, proto)
    H = c["hyper"]
    total = (H["beta"] * torch.relu(H["BND"] - ce_f) + ce_r + H["alpha"] * sl + H["pro_f_weight"] * torch.relu(H["BND_pro"] - kl_f) + H["pro_r_weight"] * kl_r)
    total.backward()
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.requires_grad}
    return m, dict(lo_r=lo_r.detach().cpu(), lo_f=lo_f.detach().cpu(), em_r=em_r.detach().cpu(), em_f=em_f.detach().cpu(), sl=sl.item(),
                   total=total.item(), ce_f=ce_f.item(), ce_r=ce_r.item(), kl_f=kl_f.item(), kl_r=kl_r.item()), grads


def test_vit_b16_full_geometry_f32_matches_the_oracle():
    """f32 parity mode at the real ViT-B/16 geometry: logits / embeddings <= 1e-4, the 12-group structure loss and get_norm_of_lora <= 1e-4,
    every loss term <= 1e-4 relative, all 48 LoRA gradients <= 1e-4 * max(1, |g|) — with the hinge on the forget CE and on the forget KL
    both active (checked), so every term of the loss feeds the gradients."""
    from util.cal_norm import get_norm_of_lora
    c = _full_geometry_case()
    o = c["out"]
    assert float(o["loss_forget"]) > 0 and float(o["kl_f"]) < c["hyper"]["BND_pro"], "the scenario must keep both hinges active"
    m, got, grads = _full_geometry_hip("fp32")
    assert len(grads) == 48 and m.lora_bucket().ngroups_block == 12
    for k, ref in (("lo_r", o["logits_r"]), ("lo_f", o["logits_f"]), ("em_r", o["emb_r"]), ("em_f", o["emb_f"])):
        assert (got[k] - ref).abs().max() < 1e-4, (k, float((got[k] - ref).abs().max()))
    for k, ref in (("sl", o["structure"]), ("total", o["total"]), ("ce_f", o["ce_f"]), ("ce_r", o["ce_r"]), ("kl_f", o["kl_f"]), ("kl_r", o["kl_r"])):
        assert abs(got[k] - float(ref)) < 1e-4 * max(1.0, abs(float(ref))), (k, got[k], float(ref))
    cn = np.array([float(v) for v in get_norm_of_lora(m, type="L2", imagenet=True)])
    assert np.abs(cn - np.array(c["cal_norm"])).max() < 1e-4
    worst = 0.0
    for n, r in c["grads"].items():
        e = float((grads[n] - r).abs().max()) / max(1.0, float(r.abs().max()))
        worst = max(worst, e)
        assert e < 1e-4, (n, e)
    print(f"[vit-b/16 full geometry f32] worst LoRA-gradient error {worst:.2e} (48 tensors), logits {float((got['lo_r'] - o['logits_r']).abs().max()):.2e}")


# (fp16 logits band 0.03 since round 5: with LayerNorm 1 folded into the QKV projection the max over these 400 logits moved 0.0167 -> 0.0203 while the
#  LoRA-gradient error fell 0.25 % -> 0.19 % and the op itself is 19 % MORE accurate against f64 — test_gemm_store_with_consumer_side_layernorm)
@pytest.mark.parametrize("dtype,lo_tol,em_tol,g_tol", [("fp16", 0.03, 0.02, 0.01), ("bf16", 0.12, 0.08, 0.06)])
def test_vit_b16_full_geometry_16bit_bands(dtype, lo_tol, em_tol, g_tol):
    """The two speed modes at the real geometry against the same oracle: logits / embeddings within the stated absolute band, every LoRA
    gradient tensor within g_tol relative Frobenius error (fp16: 1 %, bf16: 6 % — the declared band of DESIGN.md section 1)."""
    c = _full_geometry_case()
    o = c["out"]
    m, got, grads = _full_geometry_hip(dtype)
    e_lo = max(float((got["lo_r"] - o["logits_r"]).abs().max()), float((got["lo_f"] - o["logits_f"]).abs().max()))
    e_em = max(float((got["em_r"] - o["emb_r"]).abs().max()), float((got["em_f"] - o["emb_f"]).abs().max()))
    worst = max(float((grads[n] - r).norm() / r.norm().clamp_min(1e-30)) for n, r in c["grads"].items())
    a = torch.cat([grads[n].reshape(-1) for n in c["grads"]]); r = torch.cat([c["grads"][n].reshape(-1) for n in c["grads"]])
    print(f"[vit-b/16 full geometry {dtype}] logits {e_lo:.4f}, emb {e_em:.4f}, LoRA gradients: total rel. {float((a - r).norm() / r.norm()):.5f}, worst tensor {worst:.4f}")
    assert e_lo < lo_tol and e_em < em_tol, (e_lo, e_em)
    assert worst < g_tol, worst
    if dtype == "fp16":      # the headroom of the loss-scaled backward at THIS geometry (depth 12), measured by the overflow guard
        rep = m.runner().loss_scale_report()
        print(f"[vit-b/16 fp16 loss scale] exponent {rep['exponent']}, largest scaled gradient {rep['seen_max']:.0f}, headroom {rep['headroom']:.1f}x")
        assert rep["exponent"] == 11 and not rep["saturated"] and rep["headroom"] >= 8.0, rep
