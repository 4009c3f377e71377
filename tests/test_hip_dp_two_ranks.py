"""Two data-parallel ranks of the HIP step on ONE GPU (gloo process group, tests/dp_gloo_gpu_child.py) against the single-process step on
the concatenated batch: the hinges must act on GLOBAL batch means, the structure gradient must be counted once, and both ranks must
finish every step with identical LoRA replicas (the reference's nn.DataParallel scatters one batch, engine_cl.py:59-125)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import recipe

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks(tmp_path, dtype, *extra):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_gloo_gpu_child.py"), str(r), "2", str(port), dtype, outs[r], *extra],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    logs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 and "DP-GLOO-GPU-OK" in lg[0] for p, lg in zip(procs, logs)), [(p.returncode, lg[0][-500:], lg[1][-3000:]) for p, lg in zip(procs, logs)]
    return [dict(np.load(o)) for o in outs]


def test_two_ranks_agree_on_a_saturated_fp16_step(tmp_path):
    """fp16 operands under data parallelism: the overflow guard is MAX-reduced behind the gradient messages (step._guard_message), so a
    backward that saturated on ONE rank makes BOTH ranks skip that update and lower their loss-scale exponents by two binades for the next
    backward — replicas stay identical, no rank applies a gradient sum that holds another rank's clipped values."""
    r0, r1 = _run_ranks(tmp_path, "fp16", "overflow")
    for k in r0:
        assert np.array_equal(r0[k], r1[k], equal_nan=True), (k, r0[k], r1[k])
    assert r0["moved"].tolist() == [True, False, True], r0["moved"]
    assert np.isinf(r0["seen"][1]) and np.isfinite(r0["seen"][[0, 2]]).all(), r0["seen"]
    e = r0["exps"]
    assert e[1] == e[0] and e[2] == e[0] - 2, e
    # ... and under graph replay (eager, capture + replay, replay x 3; the saturated one is a replay)
    assert r0["gmoved"].tolist() == [True, True, True, False, True], r0["gmoved"]
    assert np.isinf(r0["gseen"][3]) and np.isfinite(r0["gseen"][[0, 1, 2, 4]]).all(), r0["gseen"]
    ge = r0["gexps"]
    assert (ge[:4] == ge[0]).all() and ge[4] == ge[0] - 2, ge


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 2e-2), ("fp16", 4e-3)])
def test_two_ranks_on_one_gpu_equal_the_single_process_step(tmp_path, dtype, tol):
    sys.path.insert(0, HERE)
    import dp_gloo_gpu_child as C
    from test_hip_graph import build
    from gslora_hip.optim import FusedAdamW
    from gslora_hip.step import gs_lora_step
    r0, r1 = _run_ranks(tmp_path, dtype)
    for k in r0:      # identical replicas and identical global meters on both ranks, bit for bit
        assert np.array_equal(r0[k], r1[k]), k
    # single process, whole batch
    cfg = recipe.cfg_small2()
    m = build(cfg, dtype, 0.0)
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
    crit = torch.nn.CrossEntropyLoss()
    kw = C.hyper(cfg)
    packs = []
    for s in range(C.STEPS):
        packs.append(gs_lora_step(m, opt, crit, *C.whole_batch(cfg, 2, s), **kw).cpu().numpy())
    packs = np.stack(packs)
    assert np.allclose(r0["packs"], packs, rtol=tol, atol=tol), (r0["packs"], packs)
    worst = 0.0
    for n, p in m.named_parameters():
        if p.requires_grad:
            a, b = p.detach().float().cpu().numpy(), r0[n]
            worst = max(worst, float(np.abs(a - b).max() / (np.abs(a).max() + 1e-12)))
    assert worst < tol * 5, worst
