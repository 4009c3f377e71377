"""ctypes binding of libgslora_hip.so (C ABI declared in include/gslora_hip.h).

This is the binding a maintainer of the reference (pure PyTorch, no FFI of its own) would add:
device pointers are `tensor.data_ptr()`, the stream is `torch.cuda.current_stream().cuda_stream`.
There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

import contextlib

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSLORA_HIP_LIB: development override (A/B of kernel build variants, tools/probes/); the default is the in-tree PRODUCT build.
# libgslora_hip_dev.so (python -m gslora_hip.build --dev) is the same ABI compiled with -DGSL_DEV: it additionally holds the GEMM
# variants that lost their A/Bs and reads the GSL_* ablation / variant knobs from the environment. The product library reads nothing.
LIB_PATH = os.environ.get("GSLORA_HIP_LIB") or os.path.join(_HERE, "libgslora_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "libgslora_hip_dev.so")

F32, BF16, F16 = 0, 1, 2      # F16: IEEE fp16 MFMA operands (round 5) — and, as an x_dtype, the forward residual stream format of both 16-bit modes
EPI_STORE, EPI_BIAS_RES_F32, EPI_BIAS_GELU, EPI_MUL, EPI_PATCH, EPI_STORE_F32, EPI_STORE_QKV_HM, EPI_BIAS_RES_BF16, EPI_PATCH_BF16 = 0, 1, 2, 3, 4, 5, 6, 7, 8
EPI_MUL_G8, EPI_BIAS_GELU_G8, EPI_BIAS_RES_F16, EPI_PATCH_F16, EPI_STORE_LN, EPI_STORE_QKV_HM_LN = 9, 10, 11, 12, 13, 14
NORM_SPLIT = 8
SEED_ON_DEVICE = 0x80000000   # flag bit of a `site` argument: `seed` is a device pointer to a uint64 (HIP-graph replays)

_vp, _i, _l, _f, _u64, _u32 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_uint32

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "gsl_version": [],
    "gsl_last_error": [],
    "gsl_patchify": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gsl_gemm_nt": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i,
                    _vp, _vp, _i, _f, _u64, _u32, _vp],
    "gsl_gemm_nt_lora": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i,
                         _f, _u64, _u32, _vp],
    "gsl_gemm_mulgrad_ws_elems": [_i, _i, _i],
    "gsl_gemm_nt_lora_mulgrad": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _vp, _vp, _i,
                                 _vp, _i, _vp, _l, _l, _vp, _vp, _l, _l, _i, _i, _vp, _i, _f, _i, _vp, _vp],
    "gsl_layernorm_fwd": [_vp, _l, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "gsl_layernorm_fwd_lora": [_vp, _l, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _vp, _i, _f, _vp, _vp],
    "gsl_layernorm_bwd": [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _i, _f, _u64, _u32, _l, _i, _vp, _vp],
    "gsl_attention_fwd": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _vp],
    "gsl_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _vp],
    "gsl_attention_fwd_cls": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _vp],
    "gsl_attention_bwd_cls": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _i, _vp],
    "gsl_lora_grad_ws_elems": [_i, _i, _i],
    "gsl_lora_grad": [_vp, _l, _vp, _i, _vp, _l, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "gsl_lora_grad_batch_ws_elems": [_vp, _i],
    "gsl_lora_grad_batch": [_vp, _i, _vp, _i, _vp, _vp],
    "gsl_cosface_prep": [_vp, _vp, _i, _i, _vp],
    "gsl_head_fwd": [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _i, _i, _vp],
    "gsl_head_bwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _f, _u64, _u32, _i, _i, _i, _vp, _vp, _i, _vp],
    "gsl_ce_fwd": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "gsl_ce_bwd": [_vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp],
    "gsl_proto_kl_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gsl_loss_combine": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp],
    "gsl_loss_combine_pack": [_vp, _vp, _i, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp],
    "gsl_loss_tail_max_rows": [],
    "gsl_loss_tail": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp],
    "gsl_proto_kl_bwd": [_vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp],
    "gsl_group_norms_fwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gsl_group_norms_bwd": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp],
    "gsl_adamw_flat": [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _vp, _vp],
    "gsl_adamw_flat_dev": [_vp, _vp, _vp, _vp, _l, _vp, _f, _f, _f, _f, _vp, _vp, _vp],
    "gsl_cast": [_vp, _vp, _l, _i, _vp],
    "gsl_transpose_cast": [_vp, _vp, _i, _i, _i, _vp],
    "gsl_pack_pad": [_vp, _l, _l, _i, _i, _f, _vp, _i, _i, _i, _vp],
    "gsl_pack_pad_batch": [_vp, _i, _l, _i, _vp],
    "gsl_dropout_mask": [_vp, _l, _f, _u64, _u32, _vp],
}
_RESTYPES = {"gsl_last_error": C.c_char_p, "gsl_lora_grad_ws_elems": C.c_long, "gsl_gemm_mulgrad_ws_elems": C.c_long,
             "gsl_lora_grad_batch_ws_elems": C.c_long}

_lib = None


def _bind(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{os.path.basename(path)} not found at {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The GS-LoRA step has no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # missing libamdhip64 etc.
        raise RuntimeError(f"cannot load {path}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{path} does not export {name}; rebuild the extension") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    return lib


def load():
    """Load the shared library (once). Raises RuntimeError — never falls back to a CPU path."""
    global _lib
    if _lib is None:
        _lib = _bind(LIB_PATH)
    return _lib


_dev = None


@contextlib.contextmanager
def use_dev():
    """Route the calls made inside the block to the development build (tests that compare the product kernels with the
    variants / ablations only the dev build contains, tools/probes/). Raises RuntimeError if it has not been built."""
    global _lib, _dev
    if _dev is None:
        _dev = _bind(DEV_LIB_PATH)
    prev = _lib
    _lib = _dev
    try:
        yield _dev
    finally:
        _lib = prev


def check(rc, what):
    if rc != 0:
        msg = load().gsl_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
