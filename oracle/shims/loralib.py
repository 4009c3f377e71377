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
  MergedLinear(..., r, enable_lora, bias): r=0 == nn.Linear; r>0 (only with --lora_pos Attention) = one adapter per enabled
    output group via grouped conv1d (see the class docstring)
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
    """loralib 0.1.2 MergedLinear (reference call site vit_face.py:349-355: enable_lora=[True,True,True], bias=False;
    r > 0 only with --lora_pos Attention): one rank-r adapter per enabled output group, applied as a grouped conv1d.
      lora_A [r * n_enabled, in], lora_B [out / n_groups * n_enabled, r], scaling = lora_alpha / r
      delta_W = zero_pad(conv1d(lora_A[None], lora_B[..., None], groups=n_enabled)[0])   -> [out, in]; group g rows = B_g @ A_g
      forward (not merged): F.linear(x, W, b) + (x @ delta_W.T) * scaling ; train()/eval() un-merge / merge like Linear."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0,
                 enable_lora=(False,), fan_in_fan_out=False, merge_weights=True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        assert out_features % len(enable_lora) == 0, "The length of enable_lora must divide out_features"
        self.r, self.lora_alpha = r, lora_alpha
        self.enable_lora = list(enable_lora)
        self.merged, self.merge_weights = False, merge_weights
        if r > 0 and any(enable_lora):
            self.lora_A = nn.Parameter(self.weight.new_zeros((r * sum(enable_lora), in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features // len(enable_lora) * sum(enable_lora), r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
            ind = self.weight.new_zeros((out_features,), dtype=torch.bool).view(len(enable_lora), -1)
            ind[torch.tensor(self.enable_lora, dtype=torch.bool), :] = True
            self.lora_ind = ind.view(-1)
        self.reset_parameters()

    def reset_parameters(self):
        nn.Linear.reset_parameters(self)
        if hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def zero_pad(self, x):
        result = x.new_zeros((len(self.lora_ind), *x.shape[1:]))
        result[self.lora_ind] = x
        return result

    def merge_AB(self):
        delta_w = F.conv1d(self.lora_A.unsqueeze(0), self.lora_B.unsqueeze(-1), groups=sum(self.enable_lora)).squeeze(0)
        return self.zero_pad(delta_w)

    def train(self, mode=True):
        nn.Linear.train(self, mode)
        if self.r > 0 and any(self.enable_lora) and self.merge_weights:
            if mode and self.merged:
                self.weight.data -= self.merge_AB() * self.scaling
                self.merged = False
            elif not mode and not self.merged:
                self.weight.data += self.merge_AB() * self.scaling
                self.merged = True
        return self

    def forward(self, x):
        result = F.linear(x, self.weight, bias=self.bias)
        if self.r > 0 and any(self.enable_lora) and not self.merged:
            result = result + (x @ self.merge_AB().T) * self.scaling
        return result


def mark_only_lora_as_trainable(model, bias="none"):
    for n, p in model.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False
    if bias != "none":
        raise NotImplementedError
