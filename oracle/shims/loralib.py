"""CPU restatement of the parts of loralib==0.1.2 (requirements.txt:2 of the reference)
that the reference calls (vit_face.py:330,333,349-355; train_own_forget_cl.py:316).

TEST INFRASTRUCTURE: used only by oracle/make_golden.py so that the *unmodified*
reference modules can be imported in the build container, where loralib is not
installed. loralib is a third-party dependency absent from /root/reference; its
published semantics (microsoft/LoRA, loralib/layers.py @0.1.2) are restated here:

  Linear(in, out, r=0, lora_alpha=1, lora_dropout=0., fan_in_fan_out=False, merge_weights=True)
    lora_A = zeros(r, in); lora_B = zeros(out, r); scaling = lora_alpha / r
    weight.requires_grad = False
    reset_parameters: nn.Linear.reset_parameters; kaiming_uniform_(lora_A, a=sqrt(5)); zeros_(lora_B)
    train(True):  if merged: W -= (B@A)*scaling ; merged=False
    train(False): if not merged: W += (B@A)*scaling ; merged=True
    forward: r>0 and not merged -> F.linear(x,W,b) + (drop(x) @ A.T @ B.T)*scaling else F.linear(x,W,b)
  MergedLinear(..., r=0, enable_lora, bias) with r=0 == nn.Linear
  mark_only_lora_as_trainable(model, bias='none'): requires_grad=False for names without 'lora_'
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class Linear(nn.Linear):
    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0,
                 fan_in_fan_out=False, merge_weights=True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        self.r = r
        self.lora_alpha = lora_alpha
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else (lambda x: x)
        self.merged = False
        self.merge_weights = merge_weights
        self.fan_in_fan_out = fan_in_fan_out
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r, in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        self.reset_parameters()

    def reset_parameters(self):
        nn.Linear.reset_parameters(self)
        if hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def train(self, mode=True):
        nn.Linear.train(self, mode)
        if mode:
            if self.merge_weights and self.merged:
                if self.r > 0:
                    self.weight.data -= (self.lora_B @ self.lora_A) * self.scaling
                self.merged = False
        else:
            if self.merge_weights and not self.merged:
                if self.r > 0:
                    self.weight.data += (self.lora_B @ self.lora_A) * self.scaling
                self.merged = True
        return self

    def forward(self, x):
        if self.r > 0 and not self.merged:
            result = F.linear(x, self.weight, bias=self.bias)
            result += (self.lora_dropout(x) @ self.lora_A.transpose(0, 1) @ self.lora_B.transpose(0, 1)) * self.scaling
            return result
        return F.linear(x, self.weight, bias=self.bias)


class MergedLinear(nn.Linear):
    """Only the r=0 form is reachable from the GS-LoRA FFN configuration (lora_pos='FFN')."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0,
                 enable_lora=(False,), fan_in_fan_out=False, merge_weights=True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        if r != 0:
            raise NotImplementedError("MergedLinear r>0 (lora_pos='Attention') is out of scope of the oracle")
        self.r = 0
        self.merged = False


def mark_only_lora_as_trainable(model, bias="none"):
    for n, p in model.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False
    if bias != "none":
        raise NotImplementedError
