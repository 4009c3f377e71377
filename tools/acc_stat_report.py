#!/usr/bin/env python
"""Accuracy deltas of the HIP path against the REAL reference (tests/golden/engine_cl_acc_stat.npz) as a table: scenarios x data seeds x
numeric configurations of the speed mode. The cells are tests/test_hip_engines.py::run_acc_stat (the same code the GPU test asserts on).
Usage (GPU box): python tools/acc_stat_report.py [--scenarios harsh,real] [config ...] > gpurun_out/acc_stat.md        configs: see CONFIGS
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd"), os.path.join(ROOT, "tests")]
import contextlib
import io

import numpy as np  # noqa: E402

CONFIGS = {      # name -> (training dtype, vit_runner attributes, engine_cl.EVAL_DTYPE)
    "fp16 (round 5 default: fp16 operands + streams, loss-scaled backward, 8-bit GELU')": ("fp16", {}, "fp32"),
    "fp16, fp16 GELU' (no 8-bit code)": ("fp16", {"GP8": False}, "fp32"),
    "fp16, evaluation in fp16 too": ("fp16", {}, "model"),
    "fp16, all wide (f32 streams, fp16 GELU')": ("fp16", {"FWD_STREAM": "f32", "GRAD_STREAM_BF16": False, "GP8": False}, "fp32"),
    "bf16 (default: fp16 forward stream)": ("bf16", {}, "fp32"),
    "bf16, evaluation in bf16 too": ("bf16", {}, "model"),
    "bf16, bf16 forward stream (round 3)": ("bf16", {"FWD_STREAM": "bf16"}, "fp32"),
    "bf16, f32 forward stream": ("bf16", {"FWD_STREAM": "f32"}, "fp32"),
    "bf16, f32 gradient stream": ("bf16", {"GRAD_STREAM_BF16": False}, "fp32"),
    "bf16, bf16 GELU' (no 8-bit code)": ("bf16", {"GP8": False}, "fp32"),
    "bf16, all three wide": ("bf16", {"FWD_STREAM": "f32", "GRAD_STREAM_BF16": False, "GP8": False}, "fp32"),
    "fp32 (parity mode)": ("fp32", {}, "fp32"),
}


def main():
    import engine_cl
    import test_hip_engines as T
    from gslora_hip import vit_runner as R
    from oracle import scenarios as S
    golden = os.path.join(ROOT, "tests", "golden")
    args = sys.argv[1:]
    scen = None
    if "--scenarios" in args:
        i = args.index("--scenarios")
        scen, args = args[i + 1].split(","), args[:i] + args[i + 2:]
    names = args or list(CONFIGS)
    print("| configuration | scenario | split | reference accuracy % | delta pp: mean +- std over seeds | worst cell pp | predictions that differ |")
    print("|---|---|---|---|---|---|---|")
    for cname in names:
        dtype, attrs, ev = CONFIGS[cname]
        saved = {k: getattr(R, k) for k in attrs}
        saved_ev = engine_cl.EVAL_DTYPE
        for k, v in attrs.items():
            setattr(R, k, v)
        engine_cl.EVAL_DTYPE = ev
        try:
            for sname in (scen or S.ACC_STAT):
                t0 = time.time()
                seeds = tuple(int(v) for v in os.environ["GSL_ACC_SEEDS"].split(",")) if os.environ.get("GSL_ACC_SEEDS") else S.acc_seeds(sname)
                seeds = seeds if dtype != "fp32" else seeds[:2]
                with contextlib.redirect_stdout(io.StringIO()):
                    stat, cells = T.acc_stat_table(dtype, sname, golden, seeds)
                n = len(cells) * S.ACC_STAT[sname]["n_per_split"]
                for split, r in stat.items():
                    print(f"| {cname} | {sname} | {split} | {r['ref_acc']:.2f} | {r['mean']:+.3f} +- {r['std']:.3f} | {r['worst']:.2f} | {r['flips']} of {n} |", flush=True)
                print(f"[{cname} / {sname}: {time.time() - t0:.0f} s]", file=sys.stderr, flush=True)
        finally:
            for k, v in saved.items():
                setattr(R, k, v)
            engine_cl.EVAL_DTYPE = saved_ev


if __name__ == "__main__":
    main()
