"""CPU tests of the host side: drop-in module tree, loralib merge state machine, engine signatures,
meters / schedule known answers, the C-ABI library (loads, exports every symbol of include/gslora_hip.h)
and the loud failure when the HIP path is asked to run without a GPU."""
import copy
import inspect
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import gslora_oracle as O
from oracle import recipe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_model(cfg, **kw):
    from vit_pytorch_face import ViT_face
    return ViT_face(loss_type="CosFace", GPU_ID=[0], num_class=cfg["num_class"], image_size=cfg["image_size"],
                    patch_size=cfg["patch_size"], dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], mlp_dim=cfg["mlp_dim"],
                    lora_rank=cfg["lora_rank"], **kw)


def test_parameter_tree_matches_reference_names_and_counts():
    import loralib as lora
    cfg = recipe.cfg_full()
    m = make_model(cfg, dropout=0.1, emb_dropout=0.1)
    shapes = recipe.param_shapes(cfg)      # asserted equal to the reference's named_parameters() by oracle/make_golden.py
    assert [n for n, _ in m.named_parameters()] == list(shapes)
    assert all(tuple(p.shape) == shapes[n] for n, p in m.named_parameters())
    lora.mark_only_lora_as_trainable(m)
    assert sum(p.numel() for p in m.parameters()) == 19_403_264
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 245_760
    assert all(("lora_" in n) == p.requires_grad for n, p in m.named_parameters())
    from util.utils import count_trainable_parameters
    assert count_trainable_parameters(m) == 245_760
    # the reference's substring tests and get_parameter lookups keep working
    assert m.get_parameter("transformer.layers.3.1.fn.fn.net.3.lora_B").shape == (512, 8)
    assert sum("fn.fn.net" in n for n, _ in m.named_parameters()) == 6 * 8
    # state_dict round trip (pre-trained .pth compatibility), strict=False tolerates missing lora keys only
    sd = {k: v for k, v in m.state_dict().items() if "lora" not in k}
    res = make_model(cfg).load_state_dict(sd, strict=False)
    assert res.unexpected_keys == [] and all("lora" in k for k in res.missing_keys)


def test_constructor_contract():
    cfg = recipe.cfg_small()
    with pytest.raises(AssertionError):
        make_model(dict(cfg, image_size=41))
    with pytest.raises(AssertionError):
        make_model(dict(cfg, image_size=32))          # 16 patches: "way too small"
    with pytest.raises(NotImplementedError):
        from vit_pytorch_face import ViT_face_low
        ViT_face_low()
    m = make_model(cfg)
    assert m.attn_scale == cfg["dim"] ** -0.5         # reference quirk: dim, not dim_head
    assert m.num_tokens == 26 and m.compute_dtype in (torch.float16, torch.bfloat16, torch.float32)


def test_lora_merge_state_machine_and_init():
    import loralib as lora
    torch.manual_seed(0)
    l = lora.Linear(64, 128, r=4)
    assert l.scaling == 0.25 and not l.weight.requires_grad and l.lora_A.requires_grad
    assert (l.lora_B == 0).all() and l.lora_A.abs().max() <= math.sqrt(6 / ((1 + 5) * 64)) + 1e-7
    with torch.no_grad():
        l.lora_B.normal_(0, 0.1)
    w0 = l.weight.detach().clone()
    v0 = l.weight._version
    l.eval()
    assert l.merged and l.weight._version > v0       # in-place: the runner's operand caches see the change
    assert torch.allclose(l.weight, w0 + (l.lora_B @ l.lora_A) * 0.25, atol=1e-7)
    l.eval()                                          # idempotent
    assert torch.allclose(l.weight, w0 + (l.lora_B @ l.lora_A) * 0.25, atol=1e-7)
    l2 = copy.deepcopy(l)
    assert l2.merged                                  # the flag survives deepcopy (EMA path of the CL driver)
    l.train()
    assert not l.merged and torch.allclose(l.weight, w0, atol=1e-6)
    ml = lora.MergedLinear(64, 192, r=0, enable_lora=[True, True, True], bias=False)
    assert ml.bias is None and [n for n, _ in ml.named_parameters()] == ["weight"]
    # r > 0 (--lora_pos Attention): loralib's grouped adapters — shapes, merge round trip against the restated loralib of the oracle
    from oracle.shims import loralib as shim
    mq = lora.MergedLinear(64, 192, r=4, enable_lora=[True, True, True], bias=False)
    assert [n for n, _ in mq.named_parameters()] == ["weight", "lora_A", "lora_B"]
    assert mq.lora_A.shape == (12, 64) and mq.lora_B.shape == (192, 4) and mq.scaling == 0.25 and not mq.weight.requires_grad
    with torch.no_grad():
        mq.lora_B.normal_()
    ref = shim.MergedLinear(64, 192, r=4, enable_lora=[True, True, True], bias=False)
    ref.load_state_dict(mq.state_dict())
    wq0 = mq.weight.clone()
    mq.eval(); ref.eval()
    assert mq.merged and torch.allclose(mq.weight, ref.weight, atol=1e-7) and not torch.allclose(mq.weight, wq0)
    mq.train()
    assert not mq.merged and torch.allclose(mq.weight, wq0, atol=1e-6)


def test_model_eval_train_merges_every_ffn_linear():
    cfg = recipe.cfg_small2()
    m = make_model(cfg)
    m.load_state_dict({k: torch.tensor(v) for k, v in recipe.make_state(cfg).items()})
    st = O.to_torch(recipe.make_state(cfg))
    merged = O.merge_lora(st, cfg)
    m.eval()
    for i in range(cfg["depth"]):
        for j in (0, 3):
            k = f"transformer.layers.{i}.1.fn.fn.net.{j}.weight"
            assert torch.allclose(m.state_dict()[k], merged[k], atol=1e-7)
    m.train()
    for k, v in m.state_dict().items():
        assert torch.allclose(v, st[k], atol=1e-6)
    from util.utils import reinitialize_lora_parameters
    reinitialize_lora_parameters(m)
    bound = O.reinit_bound(cfg["dim"])
    for n, p in m.named_parameters():
        if "lora_B" in n:
            assert (p == 0).all()
        if n.endswith("net.0.lora_A"):
            assert 0.7 * bound < p.abs().max() <= bound + 1e-7


def test_hip_path_fails_loudly_without_gpu():
    cfg = recipe.cfg_small()
    m = make_model(cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        with torch.no_grad():
            m(torch.zeros(2, 3, 40, 40), torch.zeros(2, dtype=torch.long))
    from gslora_hip import ops
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        ops.ce_fwd(torch.zeros(2, 10), torch.zeros(2, dtype=torch.long))
    import loralib as lora
    l = lora.Linear(64, 64, r=4)
    with pytest.raises(RuntimeError):
        l(torch.zeros(3, 64))                           # differentiable stand-alone use is refused, not emulated
    import engine_cl
    with pytest.raises(NotImplementedError):
        engine_cl.train_one_epoch_regularzation()


ENGINE_CL_ARGS = ["model", "dataloader_forget", "dataloader_remain", "device", "criterion", "optimizer", "epoch", "losses_forget",
                  "losses_remain", "losses_total", "losses_structure", "top1_forget", "top1_remain", "beta", "alpha", "BND", "batch",
                  "testloader_forget", "testloader_remain", "forget_acc_before", "highest_H_mean", "cfg", "task_i", "use_prototype",
                  "prototype_dict", "prototype_weight_forget", "prototype_weight_remain", "losses_prototype_forget",
                  "losses_prototype_remain", "dataloader_open"]
ENGINE_ARGS = ENGINE_CL_ARGS[:22] + ["dataloader_open", "prototype_weight_forget", "prototype_weight_remain", "use_prototype",
                                     "prototype_dict", "losses_prototype_forget", "losses_prototype_remain"]


def test_engine_signatures_are_the_references():
    import engine
    import engine_cl
    assert list(inspect.signature(engine_cl.train_one_epoch).parameters) == ENGINE_CL_ARGS
    assert list(inspect.signature(engine.train_one_epoch).parameters) == ENGINE_ARGS
    assert list(inspect.signature(engine_cl.eval_data).parameters) == ["model", "dataloader", "device", "mode", "batch"]
    assert list(inspect.signature(engine_cl.evaluate).parameters) == [
        "model", "testloader_forget", "testloader_remain", "device", "batch", "epoch", "forget_acc_before", "highest_H_mean", "cfg",
        "optimizer", "task_i", "testloader_open"]
    assert list(inspect.signature(engine_cl.get_prototype_loss).parameters) == ["output", "labels", "prototype_dict", "distance"]
    assert list(inspect.signature(engine.get_structure_loss).parameters) == ["model", "num_layers", "group_type", "group_pos"]
    ref = "/root/reference/engine_cl.py"          # only present in the build container: cross-check the committed lists
    if os.path.exists(ref):
        import ast
        for path, fn, want in [(ref, "train_one_epoch", ENGINE_CL_ARGS), ("/root/reference/engine.py", "train_one_epoch", ENGINE_ARGS)]:
            tree = ast.parse(open(path).read())
            f = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == fn)
            assert [a.arg for a in f.args.args] == want


def test_meters_accuracy_and_schedule_known_answers(golden_dir):
    from gslora_hip.optim import CosineLRScheduler, FusedAdamW, create_optimizer, create_scheduler
    from util.utils import AverageMeter
    g = np.load(os.path.join(golden_dir, "host_kats.npz"))
    m = AverageMeter()
    for v, n in [(1.5, 4), (2.5, 2), (-1.0, 10)]:
        m.update(v, n)
    assert np.allclose([m.val, m.avg, m.sum, m.count], g["meter"])
    p = torch.nn.Parameter(torch.zeros(4, 4))
    opt = FusedAdamW([p], lr=1e-2)
    sch = CosineLRScheduler(opt, t_initial=100, lr_min=1e-5)
    for e, lr in [(0, 1.0e-2), (1, 9.99754e-3), (25, 8.53700e-3), (50, 5.005e-3), (75, 1.47300e-3), (99, 1.24647e-5), (100, 1e-5)]:
        sch.step(e)
        assert abs(opt.param_groups[0]["lr"] - lr) < 2e-8 + 1e-5 * lr
        assert abs(opt.param_groups[0]["lr"] - O.cosine_lr(e)) < 1e-12

    class A:
        opt, lr, weight_decay, opt_eps, opt_betas, sched, epochs, min_lr, warmup_epochs, warmup_lr, cooldown_epochs = \
            "adamw", 1e-2, 0.05, 1e-8, None, "cosine", 100, 1e-5, 0, 1e-6, 10
    import loralib as lora
    cfg = recipe.cfg_small()
    model = make_model(cfg)
    lora.mark_only_lora_as_trainable(model)
    o = create_optimizer(A, model)
    assert len(o.param_groups) == 1 and o.param_groups[0]["weight_decay"] == 0.05     # all LoRA tensors are 2-D -> decayed
    assert sum(q.numel() for q in o.param_groups[0]["params"]) == sum(q.numel() for q in model.parameters() if q.requires_grad)
    s, n = create_scheduler(A, o)
    assert n == 110 and isinstance(s, CosineLRScheduler)
    assert O.class_order()[:5] == list(g["class_order"][:5])


def test_meter_queue_replays_in_order():
    from gslora_hip.step import MeterQueue
    from util.utils import AverageMeter
    q = MeterQueue()
    meters = {k: AverageMeter() for k in MeterQueue.ORDER}
    q.push(torch.arange(8, dtype=torch.float32), 4, 2)
    q.push(torch.arange(8, dtype=torch.float32) * 2, 3, 5)
    q.flush(meters)
    assert meters["losses_forget"].count == 7 and meters["losses_remain"].count == 7        # forget-weighted vs remain-weighted
    assert meters["losses_remain"].val == 2.0 and abs(meters["losses_remain"].avg - (1 * 4 + 2 * 3) / 7) < 1e-12
    assert meters["top1_forget"].avg == (4 * 2 + 8 * 5) / 7 and not q.pending


def test_library_loads_and_exports_every_header_symbol():
    from gslora_hip import _lib
    lib = _lib.load()
    assert lib.gsl_version() >= 100 and lib.gsl_last_error() is not None
    header = open(os.path.join(ROOT, "include", "gslora_hip.h")).read()
    declared = set(re.findall(r"\b(gsl_[a-z0-9_]+)\s*\(", header)) - {"gsl_dropout_keep"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    # argument validation happens before any launch: a bad call returns GSL_ERR_ARG and sets the message (no GPU needed)
    rc = lib.gsl_adamw_flat(None, None, None, None, 0, 0.0, 0.9, 0.999, 1e-8, 0.0, 0, None, None)
    assert rc == -1 and b"gsl_adamw_flat" in lib.gsl_last_error()
    rc = lib.gsl_gemm_nt(None, 0, None, 0, 63, None, 0, None, 0, 0, 4, 4, 0, 0, 1.0, None, None, None, None, None, 0, None, None, 0,
                         0.0, 0, 0, None)
    assert rc == -1


def test_prototype_table_from_dict():
    from gslora_hip.losses import prototype_table
    d = {3: torch.ones(8), 0: torch.arange(8.0)}
    t = prototype_table(d, "cpu")
    # classes without a prototype hold NaN (the reference raises KeyError for them; the device look-up turns the loss NaN instead)
    assert t.shape == (4, 8) and (t[3] == 1).all() and torch.isnan(t[1]).all() and torch.isnan(t[2]).all() and torch.equal(t[0], torch.arange(8.0))
    assert prototype_table(d, "cpu") is t            # cached per dict object


# ---- ViT-B/16 adapter: module tree, surgery helpers, loud failure on CPU -----------------------------------------------
def test_modified_vit_tree_names_and_surgery(tmp_path, monkeypatch):
    import loralib as lora
    from oracle import recipe
    from util import utils as U
    from vit_pytorch_face import ModifiedViT
    from vit_pytorch_face.modified_VIT import vit_b_16
    cfg = recipe.cfg_vitb_small()
    vit = vit_b_16(image_size=cfg["image_size"], patch_size=cfg["patch_size"], num_layers=cfg["depth"], num_heads=cfg["heads"],
                   hidden_dim=cfg["dim"], mlp_dim=cfg["mlp_dim"], num_classes=cfg["num_class"])
    m = U.replace_ffn_with_lora(ModifiedViT(vit), rank=cfg["lora_rank"])
    assert [n for n, _ in m.named_parameters()] == list(recipe.tv_param_shapes(cfg))
    assert {n: tuple(p.shape) for n, p in m.named_parameters()} == recipe.tv_param_shapes(cfg)
    m.load_state_dict({k: torch.tensor(v) for k, v in recipe.make_tv_state(cfg).items()}, strict=True)
    monkeypatch.chdir(tmp_path)
    cmap = {0: 5, 1: 2, 2: 19}
    m2 = U.modify_head(m, current_id_to_original_id=cmap, device="cpu")
    assert m2 is not m and m2.heads.head.weight.shape == (3, cfg["dim"])
    assert torch.equal(m2.heads.head.weight, m.heads.head.weight[[5, 2, 19]]) and torch.equal(m2.heads.head.bias, m.heads.head.bias[[5, 2, 19]])
    assert (tmp_path / "results/original_VIT_head/classifier.pth").exists()
    m3 = U.resume_head(m2, device="cpu")
    assert torch.equal(m3.heads.head.weight, m.heads.head.weight) and torch.equal(m3.heads.head.bias, m.heads.head.bias)
    lora.mark_only_lora_as_trainable(m2)
    assert U.count_trainable_parameters(m2) == sum(p.numel() for n, p in m2.named_parameters() if "lora_" in n)
    sp = m2.hip_spec()
    assert (sp.num_tokens, sp.dim, sp.heads, sp.lora_rank, len(sp.blocks), sp.head_kind) == (17, 64, 1, 4, 12, "linear")
    assert abs(sp.ln_eps - 1e-6) < 1e-12 and abs(sp.attn_scale - 0.125) < 1e-12
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m2(torch.zeros(1, 3, 64, 64), None)


def test_vit_b16_parameter_count():
    from vit_pytorch_face.modified_VIT import vit_b_16
    assert sum(p.numel() for p in vit_b_16().parameters()) == 86_567_656      # torchvision's published vit_b_16 size


def test_few_shot_sampler_matches_reference(golden_dir):
    """util.utils.create_few_shot_dataset draws the same indices as the reference for a given seed (golden from the real function)."""
    from image_iter import CustomSubset
    from util.utils import create_few_shot_dataset, get_unique_classes

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            self.targets = [(7 * i + 3) % 12 for i in range(108)]
            self.classes = [f"c{i}" for i in range(12)]

        def __len__(self):
            return len(self.targets)

        def __getitem__(self, i):
            return torch.tensor([float(i)]), self.targets[i]
    g = np.load(os.path.join(golden_dir, "host_kats.npz"))
    fs = create_few_shot_dataset(DS(), 4, seed=2024)
    assert isinstance(fs, CustomSubset) and list(fs.indices) == list(g["few_shot_indices"])
    assert [float(fs[0][0]), float(fs[0][1]), float(len(fs))] == list(g["few_shot_first"])
    assert get_unique_classes(fs, None) == ([f"c{i}" for i in range(12)], 12)
    counts = {}
    for i in fs.indices:
        counts[fs.targets[i]] = counts.get(fs.targets[i], 0) + 1
    assert set(counts.values()) == {4}
    with pytest.raises(ValueError):
        create_few_shot_dataset(DS(), 10, seed=1)


def test_driver_per_task_hyper_parameters_follow_the_reference():
    """driver_cl.task_hyper == train_own_forget_cl.py:999-1011: cl_beta_list[task], cl_prof_list[task] overriding pro_f_weight when the
    list is given, and --warmup_alpha as a FLAG that makes alpha 0 before alpha_epoch and big_alpha (not --alpha) from then on; the
    recipe of scripts/run_cl_forget.sh:217-218 parses. Defaults are the reference's (util/args.py:363-376)."""
    import driver_cl
    a = driver_cl.get_args(["--num_tasks", "4", "--alpha", "0.0001", "--cl_beta_list", "0.2", "0.25", "0.25", "0.2", "--pro_f_weight", "0.01",
                            "--average_weight", "--ema_epoch", "30", "--ema_decay", "0.9", "--cl_prof_list", "0.015", "0.06", "0.025", "0.012"])
    assert [driver_cl.task_hyper(a, t, 0) for t in range(4)] == [(0.2, 0.015, 1e-4), (0.25, 0.06, 1e-4), (0.25, 0.025, 1e-4), (0.2, 0.012, 1e-4)]
    with pytest.raises(IndexError):
        driver_cl.task_hyper(a, 4, 0)
    b = driver_cl.get_args(["--warmup_alpha", "--alpha_epoch", "3", "--big_alpha", "0.002", "--alpha", "0.5", "--beta", "0.3"])
    assert [driver_cl.task_hyper(b, 1, e)[2] for e in range(5)] == [0.0, 0.0, 0.0, 0.002, 0.002]
    assert driver_cl.task_hyper(b, 2, 0)[:2] == (0.3, b.pro_f_weight)          # no lists: --beta / --pro_f_weight for every task
    d = driver_cl.get_args([])
    assert (d.ema_decay, d.ema_epoch, d.big_alpha, d.alpha_epoch, d.warmup_alpha, d.cl_beta_list, d.cl_prof_list) == (0.99, 50, 1e-4, 20, False, [], [])


# ---------------------------------------------------------------------------------------------------------------- ADVICE (round 2)
def test_data_parallel_gradient_messages_do_not_depend_on_the_mode(monkeypatch):
    """Eager with the overlapped first message, eager without overlap (two backwards) and the HIP-graph segments must post the SAME
    all-reduces in the SAME order (pack | blocks 1..L-1 | block 0): ranks that momentarily run different modes — one replays a captured
    graph, one runs a first eager step, one fell back after a failed capture — would otherwise mismatch and hang or corrupt gradients.
    The capture pass itself must post nothing."""
    from gslora_hip import step as st
    sent = []
    monkeypatch.setattr(st.dist, "all_reduce", lambda t, op=None, async_op=False: (sent.append(t.numel() if op in (None, st.dist.ReduceOp.SUM) else ("max", t.dtype)),
                                                                                  type("W", (), {"wait": lambda self: None})())[1])
    flat = torch.zeros(100)

    class Runner:
        grad_hook = None
        guard = None

        def overflow_guard(self):
            return self.guard

    class Net:
        r = _runner = Runner()

        def runner(self):
            return self.r

    class Backend:
        @staticmethod
        def early_grad_slice(net):
            return flat, 30

        @staticmethod
        def grad_bucket(net):
            return flat
    net = Net()
    # eager, overlapped: the hook of layer 1 fires during the backward
    red = st._OverlappedBucketReduce(net, Backend)
    assert net.r.grad_hook is not None
    net.r.grad_hook(3); net.r.grad_hook(1); net.r.grad_hook(1); net.r.grad_hook(0)
    red.finish()
    assert sent == [70, 30] and net.r.grad_hook is None
    # eager, two backwards (fuse_batches=False): no hook is installed, same messages after the backward
    sent.clear()
    red = st._OverlappedBucketReduce(net, Backend, overlap=False)
    assert net.r.grad_hook is None
    red.finish()
    assert sent == [70, 30]
    # a backward that raises: the hook comes off the runner, nothing further is posted
    sent.clear()
    red = st._OverlappedBucketReduce(net, Backend)
    red.cancel()
    assert net.r.grad_hook is None and sent == []
    # graph segments: nothing during the capture pass, the eager sequence at every replay
    cap = st._SegmentedCapture.__new__(st._SegmentedCapture)
    cap.graphs, cap.colls, cap._ctx = [], [], None
    monkeypatch.setattr(st._SegmentedCapture, "begin", lambda self: self.graphs.append(type("G", (), {"replay": lambda s: None})()))
    monkeypatch.setattr(st._SegmentedCapture, "end", lambda self, exc=(None, None, None): None)
    cap.begin()
    pack = torch.zeros(8)
    cap.all_reduce_scalars(pack)
    cap.bucket_reducer(net, Backend).finish()
    cap.end()
    assert sent == []
    cap.replay()
    assert sent == [8, 70, 30]
    # single-block models: one message in every mode
    sent.clear()
    one = type("B1", (), {"grad_bucket": staticmethod(lambda n: flat)})
    st._OverlappedBucketReduce(net, one).finish()
    assert sent == [100]
    # fp16 operands: the overflow guard follows the gradient messages as one int32 MAX word, in every mode
    Runner.guard = torch.zeros(2)
    G = ("max", torch.int32)
    for overlap in (True, False):
        sent.clear()
        red = st._OverlappedBucketReduce(net, Backend, overlap=overlap)
        if overlap:
            net.r.grad_hook(1)
        red.finish()
        assert sent == [70, 30, G], sent
    sent.clear()
    cap.graphs, cap.colls = [], []
    cap.begin()
    cap.all_reduce_scalars(pack)
    cap.bucket_reducer(net, Backend).finish()
    assert sent == []
    cap.replay()
    assert sent == [8, 70, 30, G], sent


def test_meter_queue_stops_on_non_finite_meters():
    from gslora_hip.step import MeterQueue
    from util.utils import AverageMeter
    q = MeterQueue()
    q.push(torch.tensor([1.0, 2.0, 3.0, 0.1, 50.0, 60.0, 0.2, 0.3]), 4, 4)
    meters = {k: AverageMeter() for k in MeterQueue.ORDER}
    q.flush(meters)
    assert meters["losses_total"].avg == 3.0
    q.push(torch.tensor([1.0, 2.0, float("nan"), 0.1, 50.0, 60.0, 0.2, 0.3]), 4, 4)
    with pytest.raises(FloatingPointError, match="non-finite"):
        q.flush(meters)
    assert q.pending == []


def test_fused_adamw_loading_an_empty_state_resets_a_stepped_optimizer_like_torch(monkeypatch):
    """a checkpoint with an EMPTY (or partial) `state` — e.g. saved from a fresh optimizer — loaded into an optimizer that
    has already stepped must restart the moments and the step count from zero, as torch.optim.AdamW does; the flat buffers stay where
    they are (captured graphs hold their addresses)."""
    from gslora_hip import ops
    from gslora_hip.optim import FusedAdamW

    def adamw_flat(p, g, m, v, lr, b1, b2, eps, wd, step, guard=None):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.mul_(1 - lr * wd).addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)
    monkeypatch.setattr(ops, "adamw_flat", adamw_flat)
    torch.manual_seed(3)
    flat, gflat = torch.randn(24), torch.randn(24)
    ps = [torch.nn.Parameter(flat[:16].view(4, 4)), torch.nn.Parameter(flat[16:].view(2, 4))]
    ps[0].grad, ps[1].grad = gflat[:16].view(4, 4), gflat[16:].view(2, 4)
    ref_ps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    for rp, p in zip(ref_ps, ps):
        rp.grad = p.grad.clone()
    opt, ref = FusedAdamW(ps, lr=1e-2, weight_decay=0.05), torch.optim.AdamW(ref_ps, lr=1e-2, weight_decay=0.05)
    empty = FusedAdamW(ps, lr=1e-2, weight_decay=0.05).state_dict()      # a fresh optimizer's checkpoint: no per-parameter state
    assert empty["state"] == {}
    for _ in range(3):
        opt.step(); ref.step()
    ent = opt._flat[0]
    ptr_m = ent["m"].data_ptr()
    assert ent["step"] == 3 and float(ent["m"].abs().sum()) > 0
    opt.load_state_dict(empty)
    ref.load_state_dict(torch.optim.AdamW(ref_ps, lr=1e-2, weight_decay=0.05).state_dict())
    assert opt._flat[0] is ent and ent["m"].data_ptr() == ptr_m
    assert ent["step"] == 0 and float(ent["m"].abs().sum()) == 0.0 and float(ent["v"].abs().sum()) == 0.0
    opt.step(); ref.step()
    assert ent["step"] == 1
    for p, rp in zip(ps, ref_ps):
        assert torch.allclose(p, rp, atol=1e-6, rtol=1e-6)
    # partial state: the parameter without an entry restarts from zero moments
    sd = opt.state_dict()
    part = {"state": {0: sd["state"][0]}, "param_groups": sd["param_groups"]}
    opt.step()
    opt.load_state_dict(part)
    assert torch.equal(ent["m"][:16], sd["state"][0]["exp_avg"].reshape(-1)) and float(ent["m"][16:].abs().sum()) == 0.0


def test_fused_adamw_load_state_dict_keeps_the_buffers_captured_graphs_point_at(monkeypatch):
    """load_state_dict() with live flat buffers copies the loaded moments / step IN PLACE (captured HIP graphs hold the addresses of
    m / v / step_dev / lr_dev), and a state_dict() taken before the next step() still carries the loaded state."""
    from gslora_hip import ops
    from gslora_hip.optim import FusedAdamW

    def adamw_flat(p, g, m, v, lr, b1, b2, eps, wd, step, guard=None):      # torch restatement of gsl_adamw_flat for the CPU
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.mul_(1 - lr * wd).addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)
    monkeypatch.setattr(ops, "adamw_flat", adamw_flat)
    flat, gflat = torch.randn(24), torch.randn(24)
    ps = [torch.nn.Parameter(flat[:16].view(4, 4)), torch.nn.Parameter(flat[16:].view(2, 4))]
    ps[0].grad, ps[1].grad = gflat[:16].view(4, 4), gflat[16:].view(2, 4)
    opt = FusedAdamW(ps, lr=1e-2)
    opt.step(); opt.step()
    ent = opt._flat[0]
    assert ent["ok"] and ent["step"] == 2
    sd = opt.state_dict()
    ptr_m, ptr_v = ent["m"].data_ptr(), ent["v"].data_ptr()
    other = {"state": {0: {"step": torch.tensor(7.0), "exp_avg": torch.full((4, 4), 0.5), "exp_avg_sq": torch.full((4, 4), 0.25)},
                       1: {"step": torch.tensor(7.0), "exp_avg": torch.full((2, 4), -0.5), "exp_avg_sq": torch.full((2, 4), 0.125)}},
             "param_groups": sd["param_groups"]}
    opt.load_state_dict(other)
    ent2 = opt._flat[0]
    assert ent2 is ent and ent["m"].data_ptr() == ptr_m and ent["v"].data_ptr() == ptr_v          # same buffers
    assert ent["step"] == 7 and torch.equal(ent["m"][:16], torch.full((16,), 0.5)) and torch.equal(ent["v"][16:], torch.full((8,), 0.125))
    opt.step()
    assert opt._flat[0]["step"] == 8
    # a fresh optimizer: the loaded state is visible to state_dict() before any step()
    opt2 = FusedAdamW(ps, lr=1e-2)
    opt2.load_state_dict(other)
    st = opt2.state_dict()["state"]
    assert float(st[0]["step"]) == 7.0 and torch.equal(st[1]["exp_avg"], torch.full((2, 4), -0.5))


def test_gelu_table_is_the_generated_one_and_as_accurate_as_declared():
    """csrc/gelu_g8_table.inc (the LDS table of the fused FFN1 epilogue, GSL_EPI_BIAS_GELU_G8 on the 8-phase kernel) is exactly what
    tools/gen_gelu_table.py emits, and its entries hold Phi(a_i) / the 8-bit GELU'(a_i) code of the exact-erf GELU (vit_face.py:331) to the
    declared accuracy: Phi to a 16-bit mantissa, the code = rne(GELU' * 200 + 26); a lookup at the NEAREST grid point is within 4.5e-4 of
    Phi(a) and within 9e-4 + half a code step of GELU'(a) for every a."""
    import math
    import struct
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.call([_sys.executable, os.path.join(root, "tools", "gen_gelu_table.py"), "--check"]) == 0, "regenerate the table"
    txt = open(os.path.join(root, "gs-lora_amd", "csrc", "gelu_g8_table.inc")).read()
    words = [int(w.strip().rstrip("u"), 16) for line in txt.splitlines() if not line.startswith("//") for w in line.split(",") if w.strip()]
    N, R = 4096, 4.5
    assert len(words) == N and words[0] == 0x1A and words[-1] == 0x3F8000E2
    d = 2 * R / (N - 1)
    phi = np.array([struct.unpack("<f", struct.pack("<I", w & 0xFFFFFF00))[0] for w in words])
    code = np.array([w & 0xFF for w in words])
    a = -R + d * np.arange(N)
    Phi = np.array([0.5 * math.erfc(-x / math.sqrt(2)) for x in a])
    gp = Phi + a * np.exp(-0.5 * a * a) / math.sqrt(2 * math.pi)
    assert (np.abs(phi - Phi) <= 2.0 ** -16 * Phi * 1.01)[1:-1].all() and abs(phi[0]) == 0 and phi[-1] == 1.0      # 15 explicit mantissa bits
    assert np.abs((code - 26) / 200.0 - gp)[1:-1].max() <= 0.0025 + 1e-12 and code.max() <= 252 and code.min() >= 0
    # the kernel's index: trunc(med3(a, -R, R) * 1820 + 8192) & ~3 -> nearest grid point; scan a fine grid incl. beyond the range
    x = np.linspace(-6.0, 6.0, 200001).astype(np.float32)
    t = np.clip(x, -R, R).astype(np.float32) * np.float32(1820.0) + np.float32(8192.0)
    idx = (t.astype(np.int64) & ~3) // 4
    assert idx.min() == 0 and idx.max() == N - 1
    Phix = np.array([0.5 * math.erfc(-float(v) / math.sqrt(2)) for v in x])
    gpx = Phix + x * np.exp(-0.5 * x.astype(np.float64) ** 2) / math.sqrt(2 * math.pi)
    assert np.abs(phi[idx] - Phix).max() <= 4.5e-4
    assert np.abs(x * (phi[idx] - Phix)).max() <= 2.8e-4          # error of h = a * Phi
    assert np.abs((code[idx] - 26) / 200.0 - gpx).max() <= 0.0025 + 9e-4


def test_no_kernel_of_the_product_library_spills_registers():
    """Round 6: a spilled register is a scratch access on the CU's in-order vector-memory pipe (the attention forward's 4 spilled VGPRs, VERDICT r05
    #5; the de-branched residual epilogue's 22 - 29 spilled dwords cost the FFN2 forward +91 MB of writes per launch until it was re-ordered). The code
    objects embedded in libgslora_hip.so (gfx950) must declare .vgpr_spill_count 0 and no private segment for every kernel."""
    import re
    import struct
    import subprocess
    import tempfile
    from gslora_hip import build as B
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not in this image")
    data = open(B.OUT, "rb").read()
    kernels, bad = 0, []
    with tempfile.TemporaryDirectory() as d:
        for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
            base = m.start()
            off = base + 32
            for _ in range(struct.unpack_from("<Q", data, base + 24)[0]):
                o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
                triple = data[off:off + tl].decode(); off += tl
                if "gfx950" not in triple or not sz:
                    continue
                fn = os.path.join(d, "co.elf")
                open(fn, "wb").write(data[base + o:base + o + sz])
                notes = subprocess.run([readelf, "--notes", fn], capture_output=True, text=True, check=True).stdout
                for blk in notes.split("- .agpr_count")[1:]:
                    kernels += 1
                    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
                    priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
                    if spill or priv:
                        bad.append((name, spill, priv))
    assert kernels > 300, kernels
    assert not bad, bad


def _gfx950_code_objects(path):
    import re
    import struct
    data = open(path, "rb").read()
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        base = m.start()
        off = base + 32
        for _ in range(struct.unpack_from("<Q", data, base + 24)[0]):
            o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
            triple = data[off:off + tl].decode(); off += tl
            if "gfx950" in triple and sz:
                yield data[base + o:base + o + sz]


def test_no_barrier_of_the_libraries_leaves_an_lds_store_unpublished():
    """Round 6: gfx950 has the back-off barrier — the compiler puts NO wait in front of a raw `s_barrier`, and `__builtin_amdgcn_s_barrier()` carries no
    fence. The t hand-over of the in-kernel-LoRA tails stored to LDS and took the raw barrier the K loops use: another wave could read a row whose
    `ds_write` was still queued (one wrong 16-row fragment about once in ten process runs of the tail-split test). The disassembly of every kernel of
    both libraries must show an `s_waitcnt ... lgkmcnt(0)` between any LDS store and the next `s_barrier` (`wg_barrier_lds()` / `__syncthreads()`); the
    K loops' raw barriers publish LDS-DMA data, which a counted `vmcnt` wait retires. (A linear scan: it flagged exactly the 56 LoRA-tail kernels
    of the library before the fix and nothing else in 1 205 + 1 969 barriers.)"""
    import re
    import subprocess
    import tempfile
    from gslora_hip import build as B
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    libs = [B.OUT] + ([B.OUT_DEV] if hasattr(B, "OUT_DEV") and os.path.exists(B.OUT_DEV) else [])
    for lib in libs:
        functions = barriers = 0
        bad = []
        with tempfile.TemporaryDirectory() as d:
            for co in _gfx950_code_objects(lib):
                fn = os.path.join(d, "co.elf")
                open(fn, "wb").write(co)
                txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", fn], capture_output=True, text=True, check=True).stdout
                name = pending = None
                for line in txt.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                    if m:
                        name, pending = m.group(1), None
                        functions += 1
                        continue
                    ins = line.split()
                    if not ins or name is None:
                        continue
                    if ins[0].startswith(("ds_write", "ds_store")):
                        pending = " ".join(ins[:4])
                    elif ins[0] == "s_waitcnt" and "lgkmcnt(0)" in line:
                        pending = None
                    elif ins[0] == "s_barrier":
                        barriers += 1
                        if pending is not None:
                            bad.append((name, pending))
                        pending = None
        assert functions > 300 and barriers > 1000, (lib, functions, barriers)
        assert not bad, (lib, bad[:6], len(bad))
