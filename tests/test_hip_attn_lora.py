"""--lora_pos Attention (adapters on the QKV projection, loralib.MergedLinear) on the HIP path, against golden vectors of the REAL
reference (tests/golden/attn_small_b3.npz from oracle/make_golden.py) and the CPU oracle. f32 mode <= 1e-4; bf16 stated per assert."""
import os

import numpy as np
import pytest
import torch

from oracle import gslora_oracle as O
from oracle import recipe

pytestmark = pytest.mark.gpu
HYPER = dict(lr=1e-2, wd=0.05, beta=0.15, alpha=1e-2, BND=105.0, BND_pro=2.0, pro_f_weight=0.05, pro_r_weight=0.1)


def build(cfg, dtype="fp32", dropout=0.0):
    import loralib as lora
    from vit_pytorch_face import ViT_face
    m = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=cfg["num_class"], image_size=cfg["image_size"], patch_size=cfg["patch_size"],
                 dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], mlp_dim=cfg["mlp_dim"], dropout=dropout, emb_dropout=dropout,
                 lora_rank=cfg["lora_rank"], lora_pos=cfg.get("lora_pos", "FFN"))
    assert [n for n, _ in m.named_parameters()] == list(recipe.param_shapes(cfg))
    m.load_state_dict({k: torch.tensor(v) for k, v in recipe.make_state(cfg).items()}, strict=True)
    lora.mark_only_lora_as_trainable(m)
    return m.to("cuda").set_compute_dtype(dtype)


def batches(cfg, batch, s=0):
    nf = max(2, cfg["num_class"] // 5)
    mk = lambda a: torch.tensor(a).cuda()
    return (mk(recipe.make_images(cfg, batch, seed=100 + s, tag="xr")), mk(recipe.make_labels(cfg, batch, seed=100 + s, tag="yr", lo=0, hi=cfg["num_class"] - nf)),
            mk(recipe.make_images(cfg, batch, seed=200 + s, tag="xf")), mk(recipe.make_labels(cfg, batch, seed=200 + s, tag="yf", lo=cfg["num_class"] - nf, hi=cfg["num_class"])))


def test_forward_merge_and_norms_match_reference(golden_dir):
    import engine
    from util.cal_norm import get_norm_of_lora
    cfg, b = recipe.cfg_small_attn(), 3
    g = np.load(os.path.join(golden_dir, "attn_small_b3.npz"))
    m = build(cfg).train()
    xr, yr, _, _ = batches(cfg, b)
    with torch.no_grad():
        lo, em = m(xr, yr)
    assert np.abs(lo.cpu().numpy() - g["fwd_logits"]).max() < 1e-4 and np.abs(em.cpu().numpy() - g["fwd_emb"]).max() < 1e-4
    m.eval()      # loralib MergedLinear merge
    assert np.abs(m.state_dict()["transformer.layers.0.0.fn.fn.to_qkv.weight"].cpu().numpy() - g["merged_w_l0_net0"]).max() < 1e-6
    with torch.no_grad():
        assert np.abs(m(xr, yr)[0].cpu().numpy() - g["eval_logits"]).max() < 1e-4
    m.train()
    with torch.no_grad():
        assert np.abs(m(xr, yr)[0].cpu().numpy() - g["roundtrip_logits"]).max() < 1e-4
    sl = engine.get_structure_loss(m, num_layers=cfg["depth"], group_type="block", group_pos="Attention").item()
    assert abs(sl - float(g["structure_loss_engine_block"])) < 1e-4
    cn = np.array([float(v) for v in get_norm_of_lora(m, type="L2", group_num=cfg["depth"], group_type="block", group_pos="Attention")])
    assert np.abs(cn - g["cal_norm_block"]).max() < 1e-4
    with pytest.raises(ValueError):
        engine.get_structure_loss(m, num_layers=cfg["depth"], group_type="block", group_pos="FFN")


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_grads_match_reference(dtype, tol, golden_dir):
    import engine
    import engine_cl
    from gslora_hip import losses
    cfg, b = recipe.cfg_small_attn(), 3
    g = np.load(os.path.join(golden_dir, "attn_small_b3.npz"))
    m = build(cfg, dtype).train()
    xr, yr, xf, yf = batches(cfg, b)
    proto_np = recipe.make_prototypes(cfg)
    proto = {c: torch.tensor(proto_np[c]) for c in range(cfg["num_class"])}
    for hy, key, lkey in ((HYPER, "grad1", "losses1"), (dict(HYPER, BND=5.0, BND_pro=0.1), "grad_inactive", None)):
        m.zero_grad()
        lo_r, em_r = m(xr, yr)
        lo_f, em_f = m(xf, yf)
        ce_r, ce_f = losses.ce_sum_top1(lo_r, yr)[0] / b, losses.ce_sum_top1(lo_f, yf)[0] / b
        sl = engine.get_structure_loss(m, num_layers=cfg["depth"], group_type="block", group_pos="Attention")
        kl_f, kl_r = engine_cl.get_prototype_loss(em_f, yf, proto), engine_cl.get_prototype_loss(em_r, yr, proto)
        total = (hy["beta"] * torch.relu(hy["BND"] - ce_f) + ce_r + hy["alpha"] * sl
                 + hy["pro_f_weight"] * torch.relu(hy["BND_pro"] - kl_f) + hy["pro_r_weight"] * kl_r)
        total.backward()
        if lkey:
            got = [ce_f.item(), ce_r.item(), total.item(), sl.item(), kl_f.item(), kl_r.item()]
            for a, r in zip(got, g[lkey]):
                assert abs(a - r) < (1e-4 if dtype == "fp32" else 5e-2) * max(1.0, abs(r)), (got, g[lkey])
        ref = torch.cat([torch.tensor(g[f"{key}::{n}"]).reshape(-1) for n, p in m.named_parameters() if p.requires_grad])
        got = torch.cat([p.grad.detach().cpu().reshape(-1) for n, p in m.named_parameters() if p.requires_grad])
        if dtype == "fp32":
            for n, p in m.named_parameters():
                if p.requires_grad:
                    r = g[f"{key}::{n}"]
                    assert np.abs(p.grad.cpu().numpy() - r).max() < tol * max(1.0, np.abs(r).max()), (key, n)
        else:
            assert float((got - ref).norm() / ref.norm()) < tol, key
            assert float(torch.dot(got, ref) / (got.norm() * ref.norm())) > 0.995


def test_engine_steps_with_graph_and_full_size_bf16():
    """Full ViT-P8S8 geometry with the QKV adapters: two eager + captured/replayed engine steps stay finite, every block's adapter gets
    gradient, eager and HIP-graph replay agree bit for bit."""
    import copy
    import loralib as lora
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import GraphedStep, gs_lora_step
    from vit_pytorch_face import ViT_face
    torch.manual_seed(0)
    m1 = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=100, image_size=112, patch_size=8, dim=512, depth=6, heads=8, mlp_dim=2048,
                  dropout=0.1, emb_dropout=0.1, lora_rank=8, lora_pos="Attention")
    with torch.no_grad():
        for n, p in m1.named_parameters():
            if n.endswith("lora_B"):
                p.normal_(0, 0.02)
    lora.mark_only_lora_as_trainable(m1)
    assert sum(p.numel() for p in m1.parameters() if p.requires_grad) == 6 * (24 * 512 + 1536 * 8)
    m1 = m1.cuda().set_compute_dtype("bf16").train()
    m2 = copy.deepcopy(m1)
    mk = lambda m: FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.05, eps=1e-8)
    o1, o2 = mk(m1), mk(m2)
    crit = torch.nn.CrossEntropyLoss()
    g = GraphedStep(m2, o2, crit)
    kw = dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block")
    for s in range(4):
        x = torch.rand(16, 3, 112, 112, device="cuda", generator=None)
        y = torch.randint(0, 100, (16,), device="cuda")
        p1 = gs_lora_step(m1, o1, crit, x[:8], y[:8], x[8:], y[8:], **kw)
        p2 = g(x[:8], y[:8], x[8:], y[8:], **kw)
        assert torch.isfinite(p1).all() and torch.equal(p1, p2), (s, p1.tolist(), p2.tolist())
    bucket = m1.lora_bucket()
    assert bucket.ngroups_block == 6 and bucket.per_layer == 2
    gn = torch.stack([torch.cat([gv.reshape(-1) for gv in bucket.grad_views[2 * i:2 * i + 2]]).norm() for i in range(6)])
    assert torch.isfinite(gn).all() and (gn > 0).all()
    for (n, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()):
        if a.requires_grad:
            assert torch.equal(a, c), n
    assert (g.eager_steps, g.captures, g.replays) == (1, 1, 3)
