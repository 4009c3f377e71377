"""ModifiedViT — the ViT-B/16 ImageNet100 adapter of the reference (`vit_pytorch_face/modified_VIT.py:5-45`),
MI355X-native: same constructor (`ModifiedViT(vit_model)`), same sub-module names (`conv_proj`, `class_token`, `encoder`,
`heads`: state_dict compatible with the reference's checkpoints and with torchvision's `vit_b_16`), same
`forward(x, label) -> (logits, cls_embedding)` with the label ignored — but the whole network runs on the hand-written
gfx950 kernels through `gslora_hip.vit_runner.ViTRunner` (conv16 patch embedding = patchify + GEMM, QKV bias, LayerNorm eps
1e-6, softmax scale head_dim^-0.5, plain `nn.Linear` head) and the hand-derived backward fills the LoRA gradients.

`vit_model` is either a real `torchvision.models.VisionTransformer` (when torchvision is installed) or the
parameter-holder tree built by `vit_b_16()` below, which reproduces torchvision's module / parameter names
(`encoder.layers.encoder_layer_{i}.{ln_1, self_attention.{in_proj_weight,in_proj_bias,out_proj}, ln_2, mlp.{0,3}}`,
`encoder.pos_embedding`, `encoder.ln`, `heads.head`) so that `replace_ffn_with_lora`, `modify_head`, `resume_head`,
`get_structure_loss(imagenet=True)` and `get_norm_of_lora(imagenet=True)` address the same parameters as in the reference.
There is no CPU fallback.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from gslora_hip.vit_runner import BlockSpec, ModelSpec
from .vit_face import HipModelMixin, compute_dtype_of, DEFAULT_DTYPE


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter holder; call ModifiedViT.forward (fused HIP path)")


class MLPBlock(nn.Sequential):
    """Linear, GELU, Dropout, Linear, Dropout — indices 0 and 3 are the linears LoRA is attached to."""

    def __init__(self, in_dim, mlp_dim, dropout):
        super().__init__(nn.Linear(in_dim, mlp_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(mlp_dim, in_dim), nn.Dropout(dropout))
        for m in (self[0], self[3]):
            nn.init.xavier_uniform_(m.weight)
            nn.init.normal_(m.bias, std=1e-6)

    def forward(self, *a, **k):
        raise RuntimeError("MLPBlock is a parameter holder; call ModifiedViT.forward (fused HIP path)")


class EncoderBlock(_Holder):
    def __init__(self, num_heads, hidden_dim, mlp_dim, dropout, attention_dropout):
        super().__init__()
        self.num_heads = num_heads
        self.ln_1 = nn.LayerNorm(hidden_dim, eps=1e-6)
        self.self_attention = nn.MultiheadAttention(hidden_dim, num_heads, dropout=attention_dropout, batch_first=True)
        self.dropout = nn.Dropout(dropout)
        self.ln_2 = nn.LayerNorm(hidden_dim, eps=1e-6)
        self.mlp = MLPBlock(hidden_dim, mlp_dim, dropout)


class Encoder(_Holder):
    def __init__(self, seq_length, num_layers, num_heads, hidden_dim, mlp_dim, dropout, attention_dropout):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.dropout = nn.Dropout(dropout)
        self.layers = nn.Sequential(OrderedDict(
            (f"encoder_layer_{i}", EncoderBlock(num_heads, hidden_dim, mlp_dim, dropout, attention_dropout)) for i in range(num_layers)))
        self.ln = nn.LayerNorm(hidden_dim, eps=1e-6)


class VisionTransformer(_Holder):
    """Parameter tree of torchvision's VisionTransformer (no computation here)."""

    def __init__(self, image_size, patch_size, num_layers, num_heads, hidden_dim, mlp_dim, dropout=0.0, attention_dropout=0.0,
                 num_classes=1000):
        super().__init__()
        assert image_size % patch_size == 0, "Input shape indivisible by patch size!"
        self.image_size, self.patch_size, self.hidden_dim, self.mlp_dim = image_size, patch_size, hidden_dim, mlp_dim
        self.attention_dropout, self.dropout, self.num_classes = attention_dropout, dropout, num_classes
        self.conv_proj = nn.Conv2d(3, hidden_dim, kernel_size=patch_size, stride=patch_size)
        self.seq_length = (image_size // patch_size) ** 2 + 1
        self.class_token = nn.Parameter(torch.zeros(1, 1, hidden_dim))
        self.encoder = Encoder(self.seq_length, num_layers, num_heads, hidden_dim, mlp_dim, dropout, attention_dropout)
        self.heads = nn.Sequential(OrderedDict(head=nn.Linear(hidden_dim, num_classes)))
        fan_in = 3 * patch_size * patch_size
        nn.init.trunc_normal_(self.conv_proj.weight, std=(1.0 / fan_in) ** 0.5)
        nn.init.zeros_(self.conv_proj.bias)
        nn.init.zeros_(self.heads.head.weight)
        nn.init.zeros_(self.heads.head.bias)

    def _process_input(self, x):
        raise RuntimeError("VisionTransformer here is a parameter holder; wrap it in ModifiedViT (fused HIP path)")


def vit_b_16(weights=None, **kwargs):
    """ViT-B/16 geometry (224 px, patch 16, 12 layers, 12 heads, dim 768, mlp 3072, 1000 classes). `weights` must be None
    or a state_dict / path to one: this image has no network, the IMAGENET1K_V1 checkpoint cannot be downloaded here."""
    cfg = dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072)
    cfg.update(kwargs)
    model = VisionTransformer(**cfg)
    if weights is not None:
        sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, os.PathLike)) else weights
        if not isinstance(sd, dict):
            raise NotImplementedError("gs-lora_amd vit_b_16: pass weights=None, a state_dict or a checkpoint path "
                                      "(torchvision weight enums need a download)")
        model.load_state_dict(sd)
    return model


class ModifiedViT(HipModelMixin, nn.Module):
    def __init__(self, vit_model):
        super().__init__()
        self.conv_proj = vit_model.conv_proj
        self.class_token = vit_model.class_token
        self.encoder = vit_model.encoder
        self.heads = vit_model.heads
        self.compute_dtype = compute_dtype_of(os.environ.get("GSLORA_DTYPE", DEFAULT_DTYPE))
        self._runner = None
        self.hip_spec()      # validate the geometry once, loudly

    def hip_spec(self):
        cp = self.conv_proj
        if cp.kernel_size != cp.stride or cp.kernel_size[0] != cp.kernel_size[1] or cp.in_channels != 3:
            raise NotImplementedError("gs-lora_amd ModifiedViT: conv_proj must be a non-overlapping square 3-channel patch conv")
        patch = cp.kernel_size[0]
        D = cp.out_channels
        enc = self.encoder
        layers = list(enc.layers.children())
        att0 = layers[0].self_attention
        heads = att0.num_heads
        if D != heads * 64:
            raise NotImplementedError("gs-lora_amd attention kernels are specialised for head_dim = 64")
        if (3 * patch * patch) % 64 or D % 64:
            raise NotImplementedError("gs-lora_amd GEMM tiles need patch_dim and dim to be multiples of 64")
        if self.training and att0.dropout > 0:
            raise NotImplementedError("attention-probability dropout is 0 in vit_b_16 and is not implemented in the HIP attention kernels")
        blocks, rank = [], None
        for lyr in layers:
            sa, l1, l2 = lyr.self_attention, lyr.mlp[0], lyr.mlp[3]
            r = int(getattr(l1, "r", 0)) if hasattr(l1, "lora_A") else 0
            rank = r if rank is None else rank
            if r != rank or (r > 0) != hasattr(l2, "lora_A"):
                raise NotImplementedError("gs-lora_amd ModifiedViT: every FFN linear must carry LoRA of the same rank")
            if lyr.dropout.p != lyr.mlp[2].p or lyr.dropout.p != lyr.mlp[4].p:
                raise NotImplementedError("gs-lora_amd ModifiedViT: one dropout probability per network")
            blocks.append(BlockSpec(lyr.ln_1, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj, lyr.ln_2, l1, l2))
        if rank is not None and rank > 0 and hasattr(blocks[0].l1, "merged") is False:
            raise NotImplementedError("FFN linears must be loralib.Linear (util.utils.replace_ffn_with_lora)")
        head = self.heads.head
        return ModelSpec(patch_size=patch, num_tokens=enc.pos_embedding.shape[1], dim=D, heads=heads, attn_scale=64 ** -0.5,
                         ln_eps=float(enc.ln.eps), dropout_p=float(layers[0].dropout.p), emb_dropout_p=float(enc.dropout.p),
                         lora_rank=rank or 0, patch_w=cp.weight, patch_is_conv=True, patch_b=cp.bias, cls=self.class_token,
                         pos=enc.pos_embedding, blocks=blocks, final_ln=enc.ln, head_kind="linear", head_w=head.weight,
                         head_b=head.bias, cos_s=1.0, cos_m=0.0)

    def forward(self, x, label=None):
        """:return: (logits [bs, classes], cls embeddings [bs, dim]); `label` is not used (reference :23-24)."""
        return self._hip_call(x, label)
