"""Build libgslora_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
SOURCES = ["gemm.hip", "norm.hip", "lora.hip", "head.hip", "attention.hip"]
OUT = os.path.join(HERE, "libgslora_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "gsl_common.h"),
                                                       os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "gslora_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[gslora_hip.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
