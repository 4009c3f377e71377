"""HIP-graph replay of the forgetting step (gslora_hip.step.GraphedStep) against the eager step: same kernels, same seeds —
the meters and every LoRA parameter must be BIT-IDENTICAL, across a learning-rate change (lr is device resident) and an
eval()/train() round trip (frozen-weight versions change -> one eager step + re-capture)."""
import copy

import pytest
import torch

from oracle import recipe

pytestmark = pytest.mark.gpu


def build(cfg, dtype, dropout):
    import loralib as lora
    from vit_pytorch_face import ViT_face
    m = ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=cfg["num_class"], image_size=cfg["image_size"],
                 patch_size=cfg["patch_size"], dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], mlp_dim=cfg["mlp_dim"],
                 dropout=dropout, emb_dropout=dropout, lora_rank=cfg["lora_rank"])
    m.load_state_dict({k: torch.tensor(v) for k, v in recipe.make_state(cfg).items()}, strict=True)
    lora.mark_only_lora_as_trainable(m)
    return m.to("cuda").set_compute_dtype(dtype).train()


def batch(cfg, b, s):
    nf = max(2, cfg["num_class"] // 5)
    mk = lambda a: torch.tensor(a).cuda()
    return (mk(recipe.make_images(cfg, b, seed=100 + s, tag="xr")), mk(recipe.make_labels(cfg, b, seed=100 + s, tag="yr", lo=0, hi=cfg["num_class"] - nf)),
            mk(recipe.make_images(cfg, b, seed=200 + s, tag="xf")), mk(recipe.make_labels(cfg, b, seed=200 + s, tag="yf", lo=cfg["num_class"] - nf, hi=cfg["num_class"])))


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
def test_graph_replay_bit_identical_to_eager(dtype):
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import GraphedStep, gs_lora_step
    cfg, b = recipe.cfg_small2(), 6
    m1 = build(cfg, dtype, 0.1)
    m2 = copy.deepcopy(m1)
    mk_opt = lambda m: FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    o1, o2 = mk_opt(m1), mk_opt(m2)
    crit = torch.nn.CrossEntropyLoss()
    proto = torch.tensor(recipe.make_prototypes(cfg)).cuda()
    kw = dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block", use_prototype=True, proto_table=proto,
              w_f=0.05, w_r=0.1, BND_pro=2.0)
    g = GraphedStep(m2, o2, crit)
    x_eval = batch(cfg, b, 99)
    for s in range(9):
        if s == 4:      # cosine schedule moves the lr between epochs: no re-capture needed
            for o in (o1, o2):
                o.param_groups[0]["lr"] = 5e-3
        if s == 6:      # evaluation between steps: loralib merge / un-merge bumps the frozen weights' versions
            for m in (m1, m2):
                m.eval()
                with torch.no_grad():
                    m(x_eval[0], x_eval[1])
                m.train()
        xr, yr, xf, yf = batch(cfg, b, s)
        p1 = gs_lora_step(m1, o1, crit, xr, yr, xf, yf, **kw)
        p2 = g(xr, yr, xf, yf, **kw)
        assert torch.equal(p1, p2), (s, p1.tolist(), p2.tolist())
        for (n, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()):
            if a.requires_grad:
                assert torch.equal(a, c), (s, n)
    # steps 0 (first sighting) and 6 (after the eval round trip) ran eagerly, 1 and 7 captured + replayed, the rest replayed
    assert (g.eager_steps, g.captures, g.replays) == (2, 2, 7)
    assert o2._flat[0]["step"] == o1._flat[0]["step"] == 9


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_engine_uses_graph_for_small_batches_and_matches_eager(tmp_path, dtype):
    """engine_cl.train_one_epoch with cfg HIP_GRAPH 'auto' (batch 5+5 -> graph) vs HIP_GRAPH False: identical meters / parameters."""
    import engine_cl
    from gslora_hip.optim import FusedAdamW
    from util.utils import AverageMeter
    cfg, b = recipe.cfg_small(), 5
    proto_np = recipe.make_prototypes(cfg)
    proto = {c: torch.tensor(proto_np[c]) for c in range(cfg["num_class"])}
    res = {}
    for mode in (False, "auto"):
        m = build(cfg, dtype, 0.1)
        opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
        crit = torch.nn.CrossEntropyLoss()
        cfgd = {"DATA_ROOT": "./data/casia100/", "BND_pro": 2.0, "MULTI_GPU": False, "WORK_PATH": str(tmp_path), "BACKBONE_NAME": "VIT",
                "HIP_GRAPH": mode}
        loader_r = [batch(cfg, b, s)[:2] for s in range(6)]
        loader_f = [batch(cfg, b, s)[2:] for s in range(6)]
        mk = AverageMeter
        meters = dict(losses_forget=mk(), losses_remain=mk(), losses_total=mk(), losses_structure=mk(), top1_forget=mk(),
                      top1_remain=mk(), losses_prototype_forget=mk(), losses_prototype_remain=mk())
        ret = engine_cl.train_one_epoch(
            model=m, dataloader_forget=loader_f, dataloader_remain=loader_r, device=torch.device("cuda"), criterion=crit, optimizer=opt,
            epoch=0, beta=0.15, alpha=1e-2, BND=105.0, batch=0, testloader_forget=None, testloader_remain=None, forget_acc_before=0.0,
            highest_H_mean=0.0, cfg=cfgd, task_i="0", use_prototype=True, prototype_dict=proto, prototype_weight_forget=0.05,
            prototype_weight_remain=0.1, **meters)
        res[mode] = ([ret[i].avg for i in range(2, 10)], {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad})
        if mode == "auto":
            g = opt._gsl_graphed[(id(m), id(crit))]
            assert (g.eager_steps, g.captures, g.replays) == (1, 1, 5)
    assert res[False][0] == res["auto"][0]
    for n in res[False][1]:
        assert torch.equal(res[False][1][n], res["auto"][1][n]), n


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_graphs_of_several_batch_shapes_are_kept(dtype):
    """A ragged last batch does not throw away the graph of the regular batch: both configurations are captured once and replayed,
    bit-identical to eager steps on a twin model."""
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import GraphedStep, gs_lora_step
    cfg = recipe.cfg_small2()
    m1 = build(cfg, dtype, 0.1)
    m2 = copy.deepcopy(m1)
    mk_opt = lambda m: FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    o1, o2 = mk_opt(m1), mk_opt(m2)
    crit = torch.nn.CrossEntropyLoss()
    kw = dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block")
    g = GraphedStep(m2, o2, crit)
    for s, b in enumerate([6, 6, 6, 4, 6, 4, 4, 6]):
        xr, yr, xf, yf = batch(cfg, b, s)
        p1 = gs_lora_step(m1, o1, crit, xr, yr, xf, yf, **kw)
        p2 = g(xr, yr, xf, yf, **kw)
        assert torch.equal(p1, p2), (s, b)
    for (n, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()):
        if a.requires_grad:
            assert torch.equal(a, c), n
    assert (g.eager_steps, g.captures, g.replays) == (2, 2, 6) and len(g.graphs) == 2


def test_segmented_graph_replay_under_data_parallel_equals_eager(monkeypatch):
    """With torch.distributed active the step is captured as three graph segments around its two eager collectives (config 5: few-shot
    batches on 8 GPUs are launch-bound on every rank). One process here: the world size is faked to 2 and the collectives are replaced by
    stand-ins that scale their argument (so that a skipped or doubled collective would show), which exercises exactly the
    capture / replay machinery of gslora_hip.step._SegmentedCapture; replays must be BIT-IDENTICAL to eager data-parallel steps."""
    import torch.distributed as dist
    from gslora_hip import step as S
    from gslora_hip.optim import FusedAdamW
    calls = {"n": 0}

    class _Work:
        def wait(self):
            return True

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        calls["n"] += 1
        t.mul_(1.25)              # a visible, deterministic "sum over ranks"
        return _Work() if async_op else None
    monkeypatch.setattr(S, "_world", lambda: 2)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    cfg, b = recipe.cfg_small2(), 4
    m1 = build(cfg, "bf16", 0.1)
    m2 = copy.deepcopy(m1)
    mk_opt = lambda m: FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    o1, o2 = mk_opt(m1), mk_opt(m2)
    crit = torch.nn.CrossEntropyLoss()
    proto = torch.tensor(recipe.make_prototypes(cfg)).cuda()
    kw = dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block", use_prototype=True, proto_table=proto,
              w_f=0.05, w_r=0.1, BND_pro=2.0)
    g = S.GraphedStep(m2, o2, crit)
    for s in range(6):
        if s == 4:
            for o in (o1, o2):
                o.param_groups[0]["lr"] = 5e-3
        xr, yr, xf, yf = batch(cfg, b, s)
        n0 = calls["n"]
        p1 = S.gs_lora_step(m1, o1, crit, xr, yr, xf, yf, **kw)
        n_eager = calls["n"] - n0
        p2 = g(xr, yr, xf, yf, **kw)
        n_graph = calls["n"] - n0 - n_eager
        assert n_eager == 3                       # packed scalars + the two messages of the overlapped gradient reduction
        # every mode posts the same three messages per step: first sighting eager; second sighting capture pass (posts
        # NOTHING) + first replay; then replays — packed scalars, blocks 1..L-1, block 0 between the segments
        assert n_graph == 3
        assert torch.equal(p1, p2), (s, p1.tolist(), p2.tolist())
        for (n, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()):
            if a.requires_grad:
                assert torch.equal(a, c), (s, n)
    assert (g.eager_steps, g.captures, g.replays) == (1, 1, 5), (g.eager_steps, g.captures, g.replays)
    seg = g.graphs[next(iter(g.graphs))]["graph"]
    assert len(seg.graphs) == 3 and [len(c) for c in seg.colls] == [1, 2], (len(seg.graphs), [len(c) for c in seg.colls])      # [fwd, sums] pack [tail, backward] grad[split:], grad[:split] [AdamW]


def test_data_parallel_step_over_a_one_rank_rccl_group_equals_the_plain_step():
    """The data-parallel form of the step (packed scalar all-reduce -> device scalar tail, overlapped two-message gradient all-reduce on a
    side stream, three graph segments around the eager collectives) driven through the REAL "nccl" (= RCCL) backend. The builder's
    boxes have one GPU, so the process group has one rank: every collective is an identity and the results must equal the plain
    single-process step — what is exercised is RCCL's stream / event plumbing against the side-stream reducer and the capture segments.
    Runs in a CHILD interpreter (tests/dp_rccl_child.py): a process that has initialised ProcessGroupNCCL keeps its watchdog /
    heartbeat threads, and the rest of this suite should not share a process with them."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, TORCH_NCCL_ENABLE_MONITORING="0")
    r = subprocess.run([sys.executable, os.path.join(here, "dp_rccl_child.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DP-RCCL-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("knob", ["INK_SMALL", "LGRAD_BATCH", "LOSS_TAIL", "LN_LORA"])
def test_alternative_launch_forms_compute_the_same_step(monkeypatch, knob):
    """Round-3 forms of the launch-bound regime against the forms they replace, on the same model and batch: the in-kernel LoRA of the
    64x64 ring kernel vs the skinny-GEMM + K-segment form, the batched LoRA-gradient reductions vs one gsl_lora_grad each, the one-launch
    loss section vs the separate kernels, LayerNorm + LoRA down-projection in one pass vs two launches. Same meters, same LoRA gradients
    (bf16 operands, f32 accumulation in a different order)."""
    from gslora_hip import step as S, vit_runner as R
    from gslora_hip.optim import FusedAdamW
    cfg, b = recipe.cfg_small2(), 6
    proto = torch.tensor(recipe.make_prototypes(cfg)).cuda()
    kw = dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block", use_prototype=True, proto_table=proto,
              w_f=0.05, w_r=0.1, BND_pro=2.0)
    res = []
    for alt in (False, True):
        if alt:
            if knob == "INK_SMALL":
                monkeypatch.setattr(R, "INK_SMALL", False)
            elif knob == "LGRAD_BATCH":
                monkeypatch.setattr(R, "LGRAD_BATCH_MAX_ROWS", 0)
            elif knob == "LOSS_TAIL":
                monkeypatch.setattr(S, "LOSS_TAIL", False)
            else:
                monkeypatch.setattr(R, "LN_LORA", True)
        m = build(cfg, "bf16", 0.1)
        opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
        pack = S.gs_lora_step(m, opt, torch.nn.CrossEntropyLoss(), *batch(cfg, b, 3), **kw)
        grads = torch.cat([p.grad.reshape(-1).float() for p in m.parameters() if p.requires_grad])
        res.append((pack.clone(), grads.clone()))
    (p0, g0), (p1, g1) = res
    tol = 2e-2 if knob == "LN_LORA" else 5e-3      # LayerNorm rows may differ by a bf16 ulp between the two LayerNorm kernels
    assert torch.allclose(p0, p1, rtol=tol, atol=tol), (p0.tolist(), p1.tolist())
    assert (g0 - g1).abs().max() <= tol * g0.abs().max(), ((g0 - g1).abs().max().item(), g0.abs().max().item())
    assert g0.abs().max() > 0


def test_replay_from_caller_filled_static_inputs_equals_replay_with_staging_copies():
    """GraphedStep.static_inputs(): a caller that writes the batch into the graph's static buffers and passes those buffers back skips the
    four staging copies of a replay; the steps are bit-identical to the ones that copy."""
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import GraphedStep
    cfg, b = recipe.cfg_small2(), 6
    m1 = build(cfg, "bf16", 0.1)
    m2 = copy.deepcopy(m1)
    mk_opt = lambda m: FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    o1, o2 = mk_opt(m1), mk_opt(m2)
    crit = torch.nn.CrossEntropyLoss()
    kw = dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block")
    g1, g2 = GraphedStep(m1, o1, crit), GraphedStep(m2, o2, crit)
    bufs = None
    for s in range(6):
        data = batch(cfg, b, s)
        p1 = g1(*data, **kw)
        if bufs is None:
            p2 = g2(*data, **kw)
            bufs = g2.static_inputs(*data, **kw)          # None until the second sighting has captured the graph
        else:
            for dst, src in zip(bufs, data):
                dst.copy_(src)
            p2 = g2(*bufs, **kw)
        assert torch.equal(p1, p2), s
    assert bufs is not None and g2.replays == 5
    for (n, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()):
        if a.requires_grad:
            assert torch.equal(a, c), n
