"""ViT-Face with rank-r LoRA on the two FFN linears — MI355X-native drop-in for the reference
`vit_pytorch_face/vit_face.py` (ViT_face :449-548, Transformer :382-446, Attention :341-379,
FeedForward :326-338, PreNorm :316-323, Residual :307-313, CosFace :146-223).

The module tree, parameter names and shapes are exactly the reference's (state_dict compatible:
`transformer.layers.{i}.1.fn.fn.net.{0,3}.lora_{A,B}` ...), the parameters are real
`nn.Parameter`s, `train()/eval()` keep loralib's merge semantics — but none of the sub-modules
computes anything in PyTorch. `ViT_face.forward` hands the whole network to
`gslora_hip.vit_runner.ViTRunner`, which runs the hand-written gfx950 kernels (forward) and, through
one `torch.autograd.Function`, the hand-derived backward that fills `lora_*.grad`.
There is no CPU fallback: calling the model on CPU tensors raises.
"""
import os

import torch
import torch.nn as nn

import loralib as lora
from gslora_hip.vit_runner import BlockSpec, ModelSpec, ViTRunner

MIN_NUM_PATCHES = 16
_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16, "float16": torch.float16, "half": torch.float16,
           "fp32": torch.float32, "f32": torch.float32, "float32": torch.float32}


def compute_dtype_of(name):
    """'fp16' | 'bf16' | 'fp32' (and their aliases) or a torch dtype -> torch dtype; ValueError lists the allowed names."""
    if not isinstance(name, str):
        if name in (torch.float32, torch.bfloat16, torch.float16):
            return name
        raise ValueError(f"gs-lora_amd: compute dtype must be torch.float32 / bfloat16 / float16, not {name!r}")
    try:
        return _DTYPES[name.lower()]
    except KeyError:
        raise ValueError(f"gs-lora_amd: unknown compute dtype {name!r}; allowed: {sorted(_DTYPES)}") from None


class CosFace(nn.Module):
    """Parameter holder for the CosFace head (s=64, m=0.35; reference :146-223). The margin is applied
    whenever a label is passed, in train and eval alike; the arithmetic lives in gsl_head_fwd/bwd."""

    def __init__(self, in_features, out_features, device_id, s=64.0, m=0.35):
        super().__init__()
        self.in_features, self.out_features, self.device_id, self.s, self.m = in_features, out_features, device_id, s, m
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, emb, label):
        raise RuntimeError("CosFace is evaluated inside ViT_face.forward (fused HIP head kernel)")

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, s={self.s}, m={self.m}"


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter holder; call ViT_face.forward (fused HIP path)")


class Residual(_Holder):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class PreNorm(_Holder):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class FeedForward(_Holder):
    def __init__(self, dim, hidden_dim, dropout=0.0, lora_rank=8):
        super().__init__()
        self.net = nn.Sequential(lora.Linear(dim, hidden_dim, r=lora_rank), nn.GELU(), nn.Dropout(dropout),
                                 lora.Linear(hidden_dim, dim, r=lora_rank), nn.Dropout(dropout))

    def lora_params(self):
        a, b = self.net[0], self.net[3]
        return (a.lora_A, a.lora_B, b.lora_A, b.lora_B)


class Attention(_Holder):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, lora_rank=0):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("gs-lora_amd attention kernels are specialised for dim_head = 64")
        inner = dim_head * heads
        self.heads = heads
        self.scale = dim ** -0.5        # reference quirk (:346): dim, not dim_head
        self.to_qkv = lora.MergedLinear(in_features=dim, out_features=inner * 3, r=lora_rank,
                                        enable_lora=[True, True, True], bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))


class Transformer(_Holder):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout, lora_rank, up=False, lora_pos="FFN"):
        super().__init__()
        if lora_pos not in ("FFN", "Attention"):
            raise ValueError("lora_pos must be 'FFN' (GS-LoRA) or 'Attention' (the reference's ablation)")
        # reference :400-425: the adapters sit either on the two FFN linears or on the QKV projection, never on both
        self.layers = nn.ModuleList([
            nn.ModuleList([Residual(PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout,
                                                           lora_rank=lora_rank if lora_pos == "Attention" else 0))),
                           Residual(PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout,
                                                             lora_rank=lora_rank if lora_pos == "FFN" else 0)))])
            for _ in range(depth)])
        self.up = up
        self.depth = depth


class _ViTFaceFn(torch.autograd.Function):
    """One autograd node for the whole network. LoRA parameters are passed so that autograd sees the
    dependency; their gradients are accumulated straight into the flat bucket whose views are the
    parameters' .grad (same observable result as autograd accumulation, no per-tensor copies)."""

    @staticmethod
    def forward(ctx, runner, img, label, *lora_params):
        logits, emb, saved = runner.forward(img, label, save=True)
        ctx.runner, ctx.saved, ctx.n = runner, saved, len(lora_params)
        if logits is None:
            return emb
        return logits, emb

    @staticmethod
    def backward(ctx, *grads):
        if ctx.saved is None:
            raise RuntimeError("ViT_face backward called twice (activations are released after the first pass)")
        if len(grads) == 2:
            dlogits, demb = grads
        else:
            dlogits, demb = None, grads[0]
        ctx.runner.backward(ctx.saved, dlogits, demb)
        ctx.saved = None
        return (None, None, None) + (None,) * ctx.n


# the speed mode a model starts in (GSLORA_DTYPE overrides): IEEE fp16 operands since round 5 — same kernels, bytes and MFMA rate as bf16,
# 3 more significand bits on every matrix-core operand; DESIGN.md section 7 has what that buys in trajectory fidelity
DEFAULT_DTYPE = "fp16"


class HipModelMixin:
    """Shared by the model families that run on ViTRunner (ViT_face, ModifiedViT): compute-dtype switch, the lazily built
    runner / flat LoRA bucket, and the one-autograd-node call."""
    _runner = None
    accepts_batch_tuple = True      # forward(img) also takes a tuple of image batches, processed as one batch (gslora_hip.step)

    def set_compute_dtype(self, name):
        """'fp16' / 'bf16' (speed: 16-bit MFMA operands of that format, f32 accumulate; fp16 runs its backward on loss-scaled gradients)
        or 'fp32' (parity: exact-f32 kernels)."""
        self.compute_dtype = compute_dtype_of(name)
        return self

    def runner(self):
        if self._runner is None:
            self._runner = ViTRunner(self)
        return self._runner

    def lora_bucket(self):
        """Flat f32 storage behind the LoRA parameters (created on first use on the GPU)."""
        return self.runner().ensure_bucket()

    def invalidate_operand_caches(self):
        """Drop the runner's operand-format copies of the frozen weights (16-bit [N,K] / [K,N] forms, the gamma-folded QKV weight with its c / d
        vectors, the normalised CosFace weight). They are keyed on (data_ptr, _version) of their parameters, which in-place autograd-visible
        updates (loralib merge / un-merge, load_state_dict's copy_) bump — writes through `.data` do not: call this after one."""
        if self._runner is not None:
            self._runner.invalidate_operand_caches()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate_operand_caches()      # (copy_ bumps the versions anyway; explicit so that a custom loader writing through .data is covered too)
        return out

    def _hip_call(self, img, label):
        runner = self.runner()
        spec = self.hip_spec()
        lora_params = [p for blk in spec.blocks for p in blk.lora_params()] if spec.lora_rank > 0 else []
        grad_on = torch.is_grad_enabled()
        if grad_on and any(p.requires_grad for n, p in self.named_parameters() if "lora_" not in n):
            raise RuntimeError(f"gs-lora_amd {type(self).__name__} trains LoRA parameters only: call "
                               "loralib.mark_only_lora_as_trainable(model) first (or run under torch.no_grad())")
        if grad_on and any(p.requires_grad for p in lora_params):
            out = _ViTFaceFn.apply(runner, img, label, *lora_params)
            return out if isinstance(out, tuple) else (None, out)
        logits, emb, _ = runner.forward(img, label, save=False)
        return logits, emb


class ViT_face(HipModelMixin, nn.Module):
    def __init__(self, *, loss_type, GPU_ID, num_class, image_size, patch_size, dim, depth, heads, mlp_dim, pool="cls",
                 channels=3, dim_head=64, dropout=0.0, emb_dropout=0.0, lora_rank=8, lora_pos: str = "FFN"):
        super().__init__()
        assert image_size % patch_size == 0, "Image dimensions must be divisible by the patch size."
        num_patches = (image_size // patch_size) ** 2
        patch_dim = channels * patch_size ** 2
        assert num_patches > MIN_NUM_PATCHES, (
            f"your number of patches ({num_patches}) is way too small for attention to be effective (at least 16). "
            "Try decreasing your patch size")
        assert pool in {"cls", "mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"
        if patch_dim % 64 or dim % 64 or mlp_dim % 64:
            raise NotImplementedError("gs-lora_amd GEMM tiles need patch_dim, dim and mlp_dim to be multiples of 64")
        self.patch_size = patch_size
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.patch_to_embedding = nn.Linear(patch_dim, dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout, lora_rank, lora_pos=lora_pos)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim))
        self.loss_type = loss_type
        self.GPU_ID = GPU_ID
        if loss_type == "None":
            print("no loss for vit_face")
        elif loss_type == "CosFace":
            self.loss = CosFace(in_features=dim, out_features=num_class, device_id=GPU_ID)
        else:
            raise NotImplementedError(f"gs-lora_amd implements the CosFace head (all GS-LoRA scripts use it), not {loss_type}")
        # geometry consumed by the runner
        self.dim, self.depth, self.heads, self.mlp_dim = dim, depth, heads, mlp_dim
        self.num_tokens = num_patches + 1
        self.lora_rank = lora_rank
        self.lora_pos = lora_pos
        self.attn_scale = dim ** -0.5
        self.dropout_p, self.emb_dropout_p = float(dropout), float(emb_dropout)
        self.compute_dtype = compute_dtype_of(os.environ.get("GSLORA_DTYPE", DEFAULT_DTYPE))
        self._runner = None

    # ---- helpers for the runner -------------------------------------------------------------
    def blocks(self):
        for attn, ff in self.transformer.layers:
            yield attn.fn, ff.fn          # the two PreNorm modules of a block

    def ffn_blocks(self):
        for _, ff in self.transformer.layers:
            yield ff.fn.fn

    def hip_spec(self):
        """What the kernels need to know about this family (see gslora_hip.vit_runner.ModelSpec)."""
        blocks = []
        for attn, ff in self.transformer.layers:
            a, f = attn.fn, ff.fn
            blocks.append(BlockSpec(a.norm, a.fn.to_qkv.weight, None, a.fn.to_out[0], f.norm, f.fn.net[0], f.fn.net[3],
                                    qkv_lora=a.fn.to_qkv if self.lora_pos == "Attention" and self.lora_rank > 0 else None))
        has_loss = self.loss_type == "CosFace"
        return ModelSpec(patch_size=self.patch_size, num_tokens=self.num_tokens, dim=self.dim, heads=self.heads,
                         attn_scale=self.attn_scale, ln_eps=1e-5, dropout_p=self.dropout_p, emb_dropout_p=self.emb_dropout_p,
                         lora_rank=self.lora_rank, patch_w=self.patch_to_embedding.weight, patch_is_conv=False,
                         patch_b=self.patch_to_embedding.bias, cls=self.cls_token, pos=self.pos_embedding, blocks=blocks,
                         final_ln=self.mlp_head[0], head_kind="cosface", head_w=self.loss.weight if has_loss else None,
                         head_b=None, cos_s=self.loss.s if has_loss else 64.0, cos_m=self.loss.m if has_loss else 0.35,
                         lora_site="attention" if self.lora_pos == "Attention" else "ffn", pool=self.pool)

    # ---- reference API ---------------------------------------------------------------------------
    def forward(self, img, label=None, mask=None):
        """:return: (logits, emb) if label is given else emb — as the reference (:523-548)."""
        if mask is not None:
            raise NotImplementedError("attention masks are never passed by the GS-LoRA engines")
        logits, emb = self._hip_call(img, label)
        return emb if label is None else (logits, emb)


def _not_in_scope(name, why):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} is outside the GS-LoRA hot path covered by gs-lora_amd ({why})")
    _Stub.__name__ = name
    return _Stub


ViT_face_low = _not_in_scope("ViT_face_low", "LIRF baseline half-network, reference vit_face.py:551-781")
ViT_face_up = _not_in_scope("ViT_face_up", "LIRF baseline half-network, reference vit_face.py:551-781")
ViTs_face = _not_in_scope("ViTs_face", "overlapping-patch variant, not used by any GS-LoRA config")
