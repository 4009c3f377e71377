"""Build libgslora_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Two libraries from the same sources:
  libgslora_hip.so      the PRODUCT: the default kernels only, -fvisibility=hidden (exactly the entry points of include/gslora_hip.h
                        are exported), no getenv() on any launch path.
  libgslora_hip_dev.so  the LAB (-DGSL_DEV): additionally the GEMM variants that lost their A/Bs (csrc/gemm_dev_*.inc), the ablation /
                        variant knobs and the cycle-stamp buffers that tools/bench_gemm*.py and tools/probes/ drive through the
                        environment. Built on demand (`python -m gslora_hip.build --dev`), selected with GSLORA_HIP_LIB; never loaded
                        by default.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
SOURCES = ["gemm.hip", "norm.hip", "lora.hip", "head.hip", "attention.hip"]
# translation units that are compiled a second time with -DGSL_OP_F16: the same kernels with IEEE fp16 MFMA operands (dtype GSL_F16), in
# namespace gsl_h16, behind hidden h16_<entry> symbols that the exported entry points forward to (csrc/gsl_common.h, csrc/gsl_h16.h)
SOURCES_F16 = ["gemm.hip", "attention.hip"]
HEADERS = ["gsl_common.h", "gsl_h16.h", "exports.map", "gelu_g8_table.inc"]
DEV_ONLY = ["gemm_dev_a.inc", "gemm_dev_b.inc", "gemm_dev_c.inc", "gemm_w4.inc", "gemm_o4.inc"]
OUT = os.path.join(HERE, "libgslora_hip.so")
OUT_DEV = os.path.join(HERE, "libgslora_hip_dev.so")


def needs_build(dev=False):
    out = OUT_DEV if dev else OUT
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS + (DEV_ONLY if dev else [])]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "gslora_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, dev=False, defines=(), out=None, tag=None):
    """defines / out / tag: an A/B build of the product sources with extra -D flags into another file (tools/probes/, never loaded by
    default: `GSLORA_HIP_LIB=<out> python bench.py`)."""
    variant = out is not None
    out = out or (OUT_DEV if dev else OUT)
    if not variant and not force and not needs_build(dev):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build", tag or ("dev" if dev else "prod"))
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + (["-DGSL_DEV"] if dev else []) + list(defines)
    procs = []
    objs = []
    units = [(src, src.replace(".hip", ".o"), []) for src in SOURCES] + [(src, src.replace(".hip", "_f16.o"), ["-DGSL_OP_F16"]) for src in SOURCES_F16]
    units.sort(key=lambda u: u[0] != "gemm.hip")      # the two gemm.hip compiles dominate: start them first
    for src, oname, extra in units:      # one hipcc per translation unit, in parallel
        obj = os.path.join(objdir, oname)
        objs.append(obj)
        cmd = [hipcc] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[gslora_hip.build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", out] + objs
    if verbose:
        print("[gslora_hip.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    # python -m gslora_hip.build [--force] [--dev] [--variant NAME -DX=1 ...]  (variant -> <repo>/build_variants/libgslora_hip_NAME.so)
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        vdir = os.path.join(os.path.dirname(os.path.dirname(HERE)), "build_variants")
        os.makedirs(vdir, exist_ok=True)
        print(build(defines=[a for a in sys.argv if a.startswith("-D")], out=os.path.join(vdir, f"libgslora_hip_{name}.so"), tag="variant_" + name,
                    dev="--dev" in sys.argv))
    else:
        build(force="--force" in sys.argv, dev="--dev" in sys.argv)
