"""Child process of tests/test_hip_graph.py::test_data_parallel_step_over_a_one_rank_rccl_group_equals_the_plain_step: the data-parallel
form of the step over a ONE-rank RCCL process group must equal the plain step (see the test's docstring)."""
import copy
import os
import socket
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "gs-lora_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import recipe  # noqa: E402
from test_hip_graph import batch, build  # noqa: E402


def main():
    from gslora_hip import step as S
    from gslora_hip.optim import FusedAdamW
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg, b = recipe.cfg_small2(), 4
        m0 = build(cfg, "bf16", 0.1)
        m1, m2 = copy.deepcopy(m0), copy.deepcopy(m0)
        mk_opt = lambda m: FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-8)
        o0, o1, o2 = mk_opt(m0), mk_opt(m1), mk_opt(m2)
        crit = torch.nn.CrossEntropyLoss()
        proto = torch.tensor(recipe.make_prototypes(cfg)).cuda()
        kw = dict(beta=0.15, alpha=1e-2, BND=105.0, use_structure=True, group_type="block", use_prototype=True, proto_table=proto,
                  w_f=0.05, w_r=0.1, BND_pro=2.0)
        g = S.GraphedStep(m2, o2, crit)
        for s in range(5):
            xr, yr, xf, yf = batch(cfg, b, s)
            S._dp_active = lambda: False
            p0 = S.gs_lora_step(m0, o0, crit, xr, yr, xf, yf, **kw)              # plain single-process step
            S._dp_active = lambda: True
            p1 = S.gs_lora_step(m1, o1, crit, xr, yr, xf, yf, **kw)              # data-parallel form, eager, RCCL collectives
            p2 = g(xr, yr, xf, yf, **kw)                                           # data-parallel form, graph segments
            torch.cuda.synchronize()
            assert torch.equal(p1, p2), (s, p1.tolist(), p2.tolist())
            # the packed scalar tail is a different kernel than the single-process one: same formulas, f32 rounding may differ in the last bit
            assert torch.allclose(p0, p1, rtol=1e-5, atol=1e-6), (s, p0.tolist(), p1.tolist())
        for (n, a), (_, c), (_, d) in zip(m0.named_parameters(), m1.named_parameters(), m2.named_parameters()):
            if a.requires_grad:
                assert torch.equal(c, d), n
                assert torch.allclose(a, c, rtol=1e-4, atol=1e-6), (n, (a - c).abs().max().item())
        assert g.captures == 1 and g.replays >= 3
    finally:
        dist.destroy_process_group()
    print("DP-RCCL-OK", flush=True)


if __name__ == "__main__":
    main()
