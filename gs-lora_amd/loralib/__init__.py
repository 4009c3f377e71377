"""`import loralib as lora` — the subset of loralib 0.1.2 that bjzhb666/GS-LoRA uses
(call sites: vit_pytorch_face/vit_face.py:330,333,349-355; train/train_own_forget_cl.py:316;
util/utils.py:573), re-implemented as parameter holders for the HIP path.

State machine preserved from loralib: `Linear.train(False)` merges W += (B@A)*scaling in place,
`train(True)` un-merges; `merged` survives deepcopy; checkpoints saved in eval() hold merged
weights. The arithmetic of a layer inside ViT_face runs in libgslora_hip.so (fused into the FFN
GEMM as an extra K segment); a layer called on its own runs the same GEMM kernel, forward only.
"""
import math

import torch
import torch.nn as nn

__all__ = ["Linear", "MergedLinear", "mark_only_lora_as_trainable", "lora_state_dict"]


class Linear(nn.Linear):
    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0, fan_in_fan_out=False,
                 merge_weights=True, **kwargs):
        if lora_dropout != 0.0 or fan_in_fan_out:
            raise NotImplementedError("gs-lora_amd loralib.Linear: lora_dropout / fan_in_fan_out are not used by GS-LoRA")
        super().__init__(in_features, out_features, **kwargs)
        self.r = r
        self.lora_alpha = lora_alpha
        self.merged = False
        self.merge_weights = merge_weights
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r, in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        self.reset_parameters()   # loralib re-draws W here too: keeps the RNG stream identical

    def reset_parameters(self):
        super().reset_parameters()
        if hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def _delta(self):
        return (self.lora_B.detach() @ self.lora_A.detach()) * self.scaling

    def train(self, mode=True):
        super().train(mode)
        if self.merge_weights and self.r > 0:
            with torch.no_grad():
                if mode and self.merged:
                    self.weight.sub_(self._delta())     # in-place: bumps _version -> operand caches refresh
                    self.merged = False
                elif not mode and not self.merged:
                    self.weight.add_(self._delta())
                    self.merged = True
        return self

    def forward(self, x):
        """Stand-alone use (outside ViT_face): forward only, through the same HIP GEMM (f32)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("loralib.Linear (gs-lora_amd) is differentiable only inside ViT_face; "
                               "wrap stand-alone calls in torch.no_grad()")
        from gslora_hip import _lib as L, ops
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        if x2.shape[1] % 64:
            raise RuntimeError("stand-alone loralib.Linear needs in_features % 64 == 0")
        out = torch.empty(x2.shape[0], self.out_features, device=x.device, dtype=torch.float32)
        A2 = W2 = None
        if self.r > 0 and not self.merged:
            A2 = torch.empty(x2.shape[0], 64, device=x.device, dtype=torch.float32)
            ops.gemm_nt(x2, ops.pack_pad(self.lora_A.detach(), self.in_features, 1, self.r, self.in_features, 64,
                                         self.in_features, torch.float32), A2, alpha=self.scaling)
            W2 = ops.pack_pad(self.lora_B.detach(), self.r, 1, self.out_features, self.r, self.out_features, 64, torch.float32)
        ops.gemm_nt(x2, self.weight.detach(), out, epilogue=L.EPI_STORE_F32, A2=A2, W2=W2,
                    bias=None if self.bias is None else self.bias.detach())
        return out.reshape(*x.shape[:-1], self.out_features)


class MergedLinear(nn.Linear):
    """Parameter holder with loralib 0.1.2's MergedLinear state machine (reference call site vit_face.py:349-355:
    `enable_lora=[True, True, True], bias=False`). r = 0 (lora_pos='FFN') is a plain linear holder; r > 0 (--lora_pos Attention)
    carries one rank-r adapter per output group: lora_A [3r, in] (rows g*r.. of group g), lora_B [out, r] (rows of group g),
    delta_W rows of group g = B_g @ A_g, scaling = lora_alpha / r. eval() merges in place, train() un-merges. The arithmetic runs in
    ViTRunner (QKV GEMM with a block-diagonal LoRA K segment)."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0, enable_lora=(False,),
                 fan_in_fan_out=False, merge_weights=True, **kwargs):
        if lora_dropout != 0.0 or fan_in_fan_out:
            raise NotImplementedError("gs-lora_amd loralib.MergedLinear: lora_dropout / fan_in_fan_out are not used by GS-LoRA")
        super().__init__(in_features, out_features, **kwargs)
        assert out_features % len(enable_lora) == 0, "The length of enable_lora must divide out_features"
        self.enable_lora = list(enable_lora)
        self.r, self.lora_alpha = r, lora_alpha
        self.merged, self.merge_weights = False, merge_weights
        if r > 0 and any(self.enable_lora):
            if not all(self.enable_lora):
                raise NotImplementedError("gs-lora_amd loralib.MergedLinear: every output group carries an adapter in GS-LoRA")
            ng = len(self.enable_lora)
            self.lora_A = nn.Parameter(self.weight.new_zeros((r * ng, in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        else:
            self.r = 0
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        if hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def _delta(self):
        ng, r = len(self.enable_lora), self.r
        A = self.lora_A.detach().view(ng, r, self.in_features)
        B = self.lora_B.detach().view(ng, self.out_features // ng, r)
        return torch.bmm(B, A).reshape(self.out_features, self.in_features) * self.scaling

    def train(self, mode=True):
        super().train(mode)
        if self.merge_weights and self.r > 0:
            with torch.no_grad():
                if mode and self.merged:
                    self.weight.sub_(self._delta())
                    self.merged = False
                elif not mode and not self.merged:
                    self.weight.add_(self._delta())
                    self.merged = True
        return self

    def forward(self, *a, **k):
        raise RuntimeError("loralib.MergedLinear (gs-lora_amd) is a parameter holder; call ViT_face.forward (fused HIP path)")


def mark_only_lora_as_trainable(model, bias="none"):
    if bias != "none":
        raise NotImplementedError("bias modes other than 'none' are not used by GS-LoRA")
    for n, p in model.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False


def lora_state_dict(model, bias="none"):
    return {k: v for k, v in model.state_dict().items() if "lora_" in k}
