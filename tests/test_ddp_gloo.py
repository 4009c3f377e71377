"""world_size-2 gloo test of the data-parallel arithmetic of gslora_hip.step.gs_lora_step.

The step function is the product code; only its four device operations are swapped for CPU stand-ins
built on the oracle (test infrastructure), because the HIP kernels cannot run here. What is proven:
two ranks with half the batch each, after the packed scalar all-reduce (hinges on GLOBAL batch means),
the 1/world pre-scaling of the parameter-only structure gradient and the flat gradient all-reduce, hold
exactly the gradient / updated parameters of one process on the whole batch — for active AND inactive
hinges (a per-rank hinge would differ: the two half-batches straddle the bound)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import gslora_oracle as O
from oracle import recipe

CFG = recipe.cfg_small()


class OracleNet(nn.Module):
    """CPU stand-in with ViT_face's call signature: (img, label) -> (logits, emb); LoRA tensors are Parameters."""

    def __init__(self, state):
        super().__init__()
        self.names = list(state)
        self.frozen = {k: torch.tensor(v) for k, v in state.items() if "lora_" not in k}
        self.lora = nn.ParameterDict({k.replace(".", "/"): nn.Parameter(torch.tensor(v)) for k, v in state.items() if "lora_" in k})

    def state(self):
        st = dict(self.frozen)
        st.update({k.replace("/", "."): p for k, p in self.lora.items()})
        return st

    def forward(self, img, label):
        return O.vit_forward(self.state(), img, label, CFG)


class OracleBackend:
    @staticmethod
    def ce_sum_top1(logits, labels):
        return (torch.nn.functional.cross_entropy(logits, labels, reduction="sum"),
                (logits.argmax(1) == labels).float().sum())

    @staticmethod
    def proto_kl_sum(emb, labels, table):
        return O.prototype_kl(emb, labels, table) * emb.shape[0]

    @staticmethod
    def structure_loss(net, group_type, grad_scale=1.0):
        s = O.structure_loss(net.state(), CFG, group_type)
        return s.detach() + grad_scale * (s - s.detach())       # value unscaled, gradient pre-divided by world

    @staticmethod
    def combine(ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r, hit_f, n_r, n_f, beta, BND, alpha, w_f, w_r, BND_pro):
        """torch restatement of gsl_loss_combine (engine_cl.py:65-125)"""
        loss_remain = ce_r_sum / n_r
        loss_forget = torch.relu(BND - ce_f_sum / n_f)
        zero = torch.zeros(())
        st = structure if structure is not None else zero
        pro_f = w_f * torch.relu(BND_pro - kl_f_sum / n_f) if kl_f_sum is not None else zero
        pro_r = w_r * (kl_r_sum / n_r) if kl_r_sum is not None else zero
        total = loss_forget * beta + loss_remain + st * alpha + (pro_f + pro_r)
        meters = torch.stack([(beta * loss_forget).detach(), loss_remain.detach(), total.detach(), (alpha * st).detach(),
                              hit_f * (100.0 / n_f), hit_r * (100.0 / n_r),
                              pro_f.detach() if kl_f_sum is not None else torch.tensor(w_f * max(BND_pro, 0.0)), pro_r.detach()])
        return total, meters

    @staticmethod
    def combine_pack(pack, ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, beta, BND, alpha, w_f, w_r, BND_pro):
        """value from the all-reduced pack, gradient through this rank's local sums (straight-through)"""
        g = lambda local, tot: local + (tot - local.detach())
        has = kl_f_sum is not None
        return OracleBackend.combine(g(ce_r_sum, pack[0]), g(ce_f_sum, pack[1]), g(kl_f_sum, pack[6]) if has else None,
                                     g(kl_r_sum, pack[7]) if has else None, structure, pack[2], pack[3], pack[4], pack[5], beta, BND,
                                     alpha, w_f, w_r, BND_pro)

    @staticmethod
    def grad_bucket(net):
        ps = list(net.lora.values())
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        OracleBackend._pending = (ps, flat)
        return flat


def _scatter_back():
    ps, flat = OracleBackend._pending
    off = 0
    for p in ps:
        p.grad.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()


class SGD1(torch.optim.SGD):
    """plain SGD whose step() first writes the all-reduced flat bucket back into the .grad tensors"""

    def step(self):
        if getattr(OracleBackend, "_pending", None) is not None and dist.is_initialized():
            _scatter_back()
        OracleBackend._pending = None
        return super().step()


def _batches():
    mk = lambda a: torch.tensor(a)
    return (mk(recipe.make_images(CFG, 6, seed=31, tag="xr")), mk(recipe.make_labels(CFG, 6, seed=31, tag="yr", hi=8)),
            mk(recipe.make_images(CFG, 6, seed=32, tag="xf")), mk(recipe.make_labels(CFG, 6, seed=32, tag="yf", lo=8)))


def _run_step(net, xs, hyper):
    from gslora_hip.step import gs_lora_step
    opt = SGD1(net.parameters(), lr=0.1)
    proto = torch.tensor(recipe.make_prototypes(CFG))
    pack = gs_lora_step(net, opt, nn.CrossEntropyLoss(), *xs, beta=0.15, alpha=1e-2, BND=hyper["BND"],
                        use_structure=True, group_type="block", use_prototype=True, proto_table=proto, w_f=0.05, w_r=0.1,
                        BND_pro=hyper["BND_pro"], backend=OracleBackend)
    return pack, {k: p.detach().clone() for k, p in net.lora.items()}


def _free_port():
    """A port the OS says is free right now (a fixed pid-derived port can collide with a socket in TIME_WAIT or another process)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, hyper, ret):
    sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gs-lora_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    net = OracleNet(recipe.make_state(CFG))
    xr, yr, xf, yf = _batches()
    sl = slice(rank * 3, rank * 3 + 3)
    pack, params = _run_step(net, (xr[sl], yr[sl], xf[sl], yf[sl]), hyper)
    ret[rank] = (pack.tolist(), {k: v.numpy() for k, v in params.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("hyper", [dict(BND=105.0, BND_pro=2.0), dict(BND=5.0, BND_pro=0.1), dict(BND=None, BND_pro=None)])
def test_two_rank_step_equals_single_process(hyper):
    # single process, whole batch
    net = OracleNet(recipe.make_state(CFG))
    xs = _batches()
    if hyper["BND"] is None:
        # place both bounds BETWEEN the two ranks' local means so that a per-rank hinge would be wrong
        with torch.no_grad():
            lo, em = net(xs[2], xs[3])
            per = torch.nn.functional.cross_entropy(lo, xs[3], reduction="none")
            a, b = per[:3].mean().item(), per[3:].mean().item()
            kl = [O.prototype_kl(em[s], xs[3][s], torch.tensor(recipe.make_prototypes(CFG))).item() for s in (slice(0, 3), slice(3, 6))]
        hyper = dict(BND=0.5 * (a + b) + 0.25 * abs(a - b), BND_pro=0.5 * (kl[0] + kl[1]) + 0.25 * abs(kl[0] - kl[1]))
        assert min(a, b) < hyper["BND"] < max(a, b)
    pack1, params1 = _run_step(net, xs, hyper)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, hyper, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in range(2):
        pack2, params2 = ret[r]
        assert np.allclose(pack2, pack1.tolist(), rtol=1e-5, atol=1e-5), (r, pack2, pack1.tolist())
        for k, v in params1.items():
            assert np.abs(params2[k] - v.numpy()).max() < 2e-6, (r, k)
    # both ranks end with identical replicas
    for k in params1:
        assert np.array_equal(ret[0][1][k], ret[1][1][k])


def _eval_worker(rank, world, port, work, ret):
    sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gs-lora_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import engine
    import engine_cl
    accs = iter([40.0, 70.0] * 8)
    engine_cl.eval_data = lambda *a, **k: next(accs)          # every rank sees the same (replicated) test loaders
    engine._eval_data_cl = lambda *a, **k: next(accs)
    model = nn.Linear(4, 4)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    cfg = {"MULTI_GPU": False, "WORK_PATH": work, "BACKBONE_NAME": "VIT"}
    h = engine_cl.evaluate(model, None, None, "cpu", batch=99, epoch=0, forget_acc_before=100.0, highest_H_mean=0.0, cfg=cfg,
                           optimizer=opt, task_i="0")
    h2 = engine.evaluate(model, None, None, "cpu", batch=199, epoch=0, forget_acc_before=100.0, highest_H_mean=0.0, cfg=cfg, optimizer=opt)
    ret[rank] = (h, h2)
    dist.destroy_process_group()


def test_two_ranks_evaluate_into_one_work_directory(tmp_path):
    """one process per GPU — every rank evaluates and reaches the same best-H-mean decision, ONE rank saves and prunes the
    shared work directory, the others wait at a barrier (no concurrent writers, no FileNotFoundError in the second pruner)."""
    import time
    work = str(tmp_path)
    open(os.path.join(work, "config.txt"), "w").write("cfg\n")
    for i, name in enumerate(["Backbone_VIT_Epoch_1_Batch_10_Time_old_checkpoint.pth", "Backbone_VIT_Epoch_1_Batch_20_Time_old_checkpoint.pth"]):
        p = os.path.join(work, name)
        torch.save({"dummy": torch.zeros(1)}, p)
        os.utime(p, (time.time() - 1000 + 10 * i, time.time() - 1000 + 10 * i))
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, work, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret[0] == ret[1] and ret[0][0] > 0
    files = sorted(f for f in os.listdir(work) if f.endswith(".pth"))
    new = [f for f in files if "_old_" not in f]
    assert len(new) == 2 and any("_Batch_100_" in f for f in new) and any("_Batch_200_" in f for f in new)      # one file per evaluate(), not per rank
    assert len(files) == 2, files          # both engines pruned down to two checkpoints, once


def _eval_disagree_worker(rank, world, port, work, ret):
    sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gs-lora_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import engine
    import engine_cl
    # rank 1 sees other accuracies (a sharded test loader) AND carries a higher best-so-far (a per-rank value after a resume):
    # on its own it would NOT take the save branch, rank 0 would
    accs = iter(([40.0, 70.0] if rank == 0 else [90.0, 20.0]) * 8)
    engine_cl.eval_data = lambda *a, **k: next(accs)
    engine._eval_data_cl = lambda *a, **k: next(accs)
    model = nn.Linear(4, 4)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    cfg = {"MULTI_GPU": False, "WORK_PATH": work, "BACKBONE_NAME": "VIT"}
    best = 0.0 if rank == 0 else 99.0
    h = engine_cl.evaluate(model, None, None, "cpu", batch=99, epoch=0, forget_acc_before=100.0, highest_H_mean=best, cfg=cfg,
                           optimizer=opt, task_i="0")
    h2 = engine.evaluate(model, None, None, "cpu", batch=199, epoch=0, forget_acc_before=100.0, highest_H_mean=best, cfg=cfg, optimizer=opt)
    t = torch.ones(1)
    dist.all_reduce(t)            # the next collective of the run still lines up
    ret[rank] = (h, h2, float(t))
    dist.destroy_process_group()


def test_two_ranks_that_disagree_on_the_best_hmean_still_take_one_decision(tmp_path):
    """the save branch of evaluate() holds a barrier, so the "new best H-mean" decision must be the GROUP's, not each rank's:
    rank 0's (H-mean, best so far) pair is broadcast. Ranks with different accuracies / different best-so-far values neither hang nor
    desynchronise, and return the same best value."""
    work = str(tmp_path)
    open(os.path.join(work, "config.txt"), "w").write("cfg\n")
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_eval_disagree_worker, args=(r, 2, port, work, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    alive = [p.is_alive() for p in procs]
    [p.terminate() for p in procs if p.is_alive()]
    assert not any(alive), "a rank hung in evaluate()"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret[0] == ret[1] and ret[0][0] > 0 and ret[0][2] == 2.0
    new = [f for f in os.listdir(work) if f.endswith(".pth")]
    # rank 0 saved once per evaluate(); engine.evaluate prunes at three directory entries (config.txt + two checkpoints): the newer one is left
    assert len(new) == 1 and "_Batch_200_" in new[0], new
