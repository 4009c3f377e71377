"""Child process of tests/test_hip_dp_two_ranks.py: ONE RANK of a two-rank data-parallel run of the HIP step. Both ranks live on the
one GPU the box has; the process group is gloo (RCCL refuses two ranks on one device), so what is exercised is everything of the N > 1
path except RCCL's transport: the real HIP kernels on half the batch, the packed scalar all-reduce in front of the hinges, the two-message
gradient all-reduce overlapped on a side stream, and the three-segment graph replay with eager collectives in between.
argv: rank world port dtype out.npz [overflow]
With `overflow` (fp16 only): three eager steps; in the second one rank 1 alone reports a saturated backward — the ranks must agree (skip
together, lower their loss scales together, keep identical replicas)."""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "gs-lora_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import recipe  # noqa: E402
from test_hip_graph import batch, build  # noqa: E402

STEPS, B = 4, 4          # images per rank and stream


def hyper(cfg):
    return dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block", use_prototype=True,
                proto_table=torch.tensor(recipe.make_prototypes(cfg)).cuda(), w_f=0.05, w_r=0.1, BND_pro=2.0)


def whole_batch(cfg, world, s):
    """The global batch of step s: rank r's share is rows [r*B, (r+1)*B) of each stream."""
    return batch(cfg, B * world, s)


def overflow_run(rank, world, out):
    from gslora_hip import step as S
    from gslora_hip.optim import FusedAdamW
    cfg = recipe.cfg_small2()
    m = build(cfg, "fp16", 0.0)
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    crit = torch.nn.CrossEntropyLoss()
    kw = hyper(cfg)
    train = [p for p in m.parameters() if p.requires_grad]
    sl = slice(rank * B, (rank + 1) * B)
    post, hit = S._post_guard, {"on": False}

    def poked(net):          # what a saturated 16-bit gradient store on THIS rank leaves in its guard before the ranks compare notes
        if hit["on"]:
            net._runner.gscale[2] = float("inf")
        post(net)
    S._post_guard = poked
    exps, seen, moved = [], [], []
    for s in range(3):
        hit["on"] = (s == 1 and rank == 1)
        before = [p.detach().clone() for p in train]
        S.gs_lora_step(m, opt, crit, *(t[sl].contiguous() for t in whole_batch(cfg, world, s)), **kw)
        torch.cuda.synchronize()
        rep = m._runner.loss_scale_report()
        exps.append(rep["exponent"]); seen.append(rep["seen_max"])
        moved.append(any(not torch.equal(a, b) for a, b in zip(before, train)))
    res = {"exps": np.array(exps), "seen": np.array(seen), "moved": np.array(moved)}
    for n, a in m.named_parameters():
        if a.requires_grad:
            res[n] = a.detach().float().cpu().numpy()
    # the same under graph replay: step 0 runs eagerly, step 1 captures and replays, steps 2 .. 4 replay the three segments with the collectives in between; rank 1
    # reports a saturated backward in step 3 — poked into the guard word right before ITS all-reduce (the only MAX message of a step)
    S._post_guard = post
    m2 = build(cfg, "fp16", 0.0)
    opt2 = FusedAdamW([p for p in m2.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    train2 = [p for p in m2.parameters() if p.requires_grad]
    g = S.GraphedStep(m2, opt2, crit)
    real_ar = dist.all_reduce

    def poking_ar(t, op=dist.ReduceOp.SUM, **kw):
        if hit["on"] and op == dist.ReduceOp.MAX:
            t.fill_(0x7f800000)            # +inf, as the int32 word the ranks compare
        return real_ar(t, op=op, **kw)
    S.dist.all_reduce = poking_ar
    gexps, gseen, gmoved = [], [], []
    try:
        for s in range(5):
            hit["on"] = (s == 3 and rank == 1)
            before = [p.detach().clone() for p in train2]
            g(*(t[sl].contiguous() for t in whole_batch(cfg, world, s)), **kw)
            torch.cuda.synchronize()
            rep = m2._runner.loss_scale_report()
            gexps.append(rep["exponent"]); gseen.append(rep["seen_max"])
            gmoved.append(any(not torch.equal(a, b) for a, b in zip(before, train2)))
    finally:
        S.dist.all_reduce = real_ar
    assert (g.eager_steps, g.captures, g.replays) == (1, 1, 4), (g.eager_steps, g.captures, g.replays)
    res.update(gexps=np.array(gexps), gseen=np.array(gseen), gmoved=np.array(gmoved))
    for n, a in m2.named_parameters():
        if a.requires_grad:
            res["g." + n] = a.detach().float().cpu().numpy()
    np.savez(out, **res)


def main():
    rank, world, port, dtype, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    from gslora_hip import step as S
    from gslora_hip.optim import FusedAdamW
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        if len(sys.argv) > 6 and sys.argv[6] == "overflow":
            overflow_run(rank, world, out)
            print("DP-GLOO-GPU-OK", flush=True)
            return
        cfg = recipe.cfg_small2()
        m1 = build(cfg, dtype, 0.0)
        m2 = copy.deepcopy(m1)
        mk_opt = lambda m: FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
        o1, o2 = mk_opt(m1), mk_opt(m2)
        crit = torch.nn.CrossEntropyLoss()
        kw = hyper(cfg)
        g = S.GraphedStep(m2, o2, crit)
        packs = []
        sl = slice(rank * B, (rank + 1) * B)
        for s in range(STEPS):
            xr, yr, xf, yf = (t[sl].contiguous() for t in whole_batch(cfg, world, s))
            p1 = S.gs_lora_step(m1, o1, crit, xr, yr, xf, yf, **kw)       # eager data-parallel step
            p2 = g(xr, yr, xf, yf, **kw)                                    # graph segments + eager collectives
            torch.cuda.synchronize()
            assert torch.equal(p1, p2), (s, p1.tolist(), p2.tolist())
            packs.append(p1.cpu().numpy())
        assert (g.eager_steps, g.captures, g.replays) == (1, 1, STEPS - 1), (g.eager_steps, g.captures, g.replays)
        res = {"packs": np.stack(packs)}
        for (n, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()):
            if a.requires_grad:
                assert torch.equal(a, c), n
                res[n] = a.detach().float().cpu().numpy()
        np.savez(out, **res)
    finally:
        dist.destroy_process_group()
    print("DP-GLOO-GPU-OK", flush=True)


if __name__ == "__main__":
    main()
