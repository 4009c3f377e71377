"""Whole-network forward/backward schedule of the ViT-Face + LoRA-FFN model on the HIP kernels.

This is the host-side "engine" behind vit_pytorch_face.ViT_face.forward: it owns
  * the flat f32 LoRA bucket (parameters become views into it, ordered group-by-group so that a
    group-lasso group is one contiguous slice),
  * the frozen-weight operand caches (bf16 casts / transposes, refreshed when a weight's
    (data_ptr, _version) changes, e.g. after loralib merge/un-merge or load_state_dict),
  * the per-forward activation stash that the hand-written backward consumes.
Reference semantics: vit_pytorch_face/vit_face.py:523-548 (forward), autograd of the same.
"""
import os

import torch

from . import _lib as L
from . import ops

PADK = 64                                  # LoRA K-segment width fed to the GEMM (r zero-padded to 64)
OP16 = (torch.bfloat16, torch.float16)     # the two operand formats of the speed mode ("bf16" / "fp16")
SITE_EMB = 1_000_000
# ---- numeric form of the bf16 speed mode: the three PRODUCT knobs (read once from the environment; module attributes, so a caller /
# tools/precision_ablation.py / the tests can also set them in-process). The f32 parity mode ignores them. README.md documents them.
# Residual-GRADIENT stream (the [M, dim] tensor every LayerNorm backward re-reads and re-writes) in bf16: -25 % of the bytes of each
# LayerNorm backward. GSLORA_GRAD_STREAM=f32 keeps it in f32.
GRAD_STREAM_BF16 = os.environ.get("GSLORA_GRAD_STREAM", "bf16").lower() != "f32"
# FORWARD residual stream x (read by every LayerNorm forward, read + written by the out-proj / FFN2 epilogues, re-read by every LayerNorm
# backward) in 2 bytes per element: f32 accumulate in the producing epilogue, one rounding on store. "f16" (default, round 4): IEEE fp16 —
# the stream is never a matrix-core operand, so it can spend its 16 bits on significand instead of exponent range: 8x finer rounding than
# bf16 at the same bytes (values clamp at +-65504; ViT residual streams are O(1 .. 100)); "bf16": round 3's form; "f32": 4 bytes
# (+9.7 GB per step). GSLORA_FWD_STREAM selects. profiles/r04_acc_stat.md has what each buys in trajectory fidelity.
FWD_STREAM = os.environ.get("GSLORA_FWD_STREAM", "f16").lower()
if FWD_STREAM not in ("f16", "bf16", "f32"):
    raise ValueError(f"GSLORA_FWD_STREAM must be f16, bf16 or f32, not {FWD_STREAM!r}")
# g' = GELU'(.) * dropmask / (1 - p) — written by the fused FFN1 epilogue, read once by the FFN2-dX epilogue — as an 8-bit fixed-point
# code (include/gslora_hip.h, GSL_EPI_BIAS_GELU_G8): half the bytes of one of the two [M, mlp] tensors of the FFN. GSLORA_GP8=0: bf16.
GP8 = os.environ.get("GSLORA_GP8", "1") != "0"

# ---- decided schedule choices (A/B'd in rounds 2 - 3, profiles/r03_notes.md). Plain constants: no environment reads; the tests that pin a
# form against the one it replaced (tests/test_hip_graph.py, tests/test_hip_model.py) patch the module attribute.
# The two [M, mlp] LoRA-gradient reductions of a block ride in the FFN2-dX epilogue (False: separate gsl_lora_grad launches).
FUSE_LORA_GRAD = True
# pool='cls' (vit_face.py:540): the head reads token 0 only and everything after a block's attention is token-wise, so in the LAST block
# only the cls query's attention output, its out-proj / LayerNorm / FFN rows are ever consumed — forward and backward of that block's
# tail run on B rows instead of B*T (exact: the skipped rows influence no output of the model). False keeps the dense forward (the
# backward then still runs on the cls rows).
TAIL_CLS = True
# ... and of that block's QKV projection only K and V are needed for every token: Q is projected for the cls rows alone (kv [M, 2*inner]
# + q_cls [B, inner]; the backward's dX GEMM contracts over 2*inner and the cls rows get their dQ term from a [B, inner] GEMM).
QSPLIT = True
# rows from which the LoRA down-projections are computed inside the 256x256 GEMM kernels (below: a separate N = 64 GEMM + a K segment
# on the small-tile kernels). Measured: profiles/r03_c_small_m.md.
INK_MIN_ROWS = 8192
# LayerNorm 1 folded into the QKV projection (16-bit modes whose forward stream has the operand format, i.e. the fp16 default): the GEMM reads
# the stream x itself with gamma folded into the weight and finishes the normalisation in its epilogue (EPI_STORE_LN); LayerNorm 1 shrinks to
# its row statistics (one read of x, no LN(x) tensor). GSLORA_LN1_FOLD=0: the LayerNorm kernel + plain GEMM of rounds 1 - 4.
LN1_FOLD = os.environ.get("GSLORA_LN1_FOLD", "1") != "0"
# fp16 operands: exponent of the loss-scaled backward — gsl_head_bwd picks the power of two S with S * max|head gradient| in [2^(E-1), 2^E)
# (0 = the library default 11: 32x headroom below 65504 at the head; the overflow guard lowers E on the device when a store saturates)
GRAD_TARGET_EXP = int(os.environ.get("GSLORA_GRAD_TARGET_EXP", "0"))
if GRAD_TARGET_EXP and not 4 <= GRAD_TARGET_EXP <= 15:
    raise ValueError(f"GSLORA_GRAD_TARGET_EXP={GRAD_TARGET_EXP}: 0 (default) or 4 .. 15")
INK_SMALL = True      # the in-kernel form on the small-tile kernel (few rows)
# The LoRA-gradient reductions of a backward pass that do not ride in the FFN2-dX epilogue are collected and issued as ONE batched pair of
# launches (gsl_lora_grad_batch) instead of two to three launches each; their operands stay alive until the end of the backward (or until
# the data-parallel hook needs the slice). Measured: few-shot 4+4 1.355 -> 1.155 ms (24 reductions, 48 launches before), ViT-B/16 48+48
# 11.01 -> 10.64 ms, 512+512 24.83 -> 24.69 ms (+1.2 GB of operands held). Row count above which the reductions run where their operands
# are produced instead (0 = always):
LGRAD_BATCH_MAX_ROWS = 1 << 30
# bf16 stream: the LayerNorm in front of the FFN can also emit the FFN1 adapter's down-projection u1 = s * LN(x) A1^T
# (gsl_layernorm_fwd_lora) instead of a skinny GEMM that re-reads LN(x). Measured time-neutral (profiles/r03_notes.md): off.
LN_LORA = False
# layout of the stashed qkv tensor in bf16 mode: head-major [B][H][3][T][64] (the QKV GEMM's store permutes, the attention kernels
# read contiguous per-head panels); False = token-major [B*T, 3*H*64] as the reference's to_qkv output (always used in f32 mode)
QKV_HEAD_MAJOR = True


class BlockSpec:
    """One pre-norm transformer block as the kernels see it: x1 = x + drop(Wo attn(LN1 x) + bo), x2 = x1 + drop(W2' drop(gelu(W1' LN2 x1)))."""
    __slots__ = ("ln1", "qkv_w", "qkv_b", "out", "ln2", "l1", "l2", "qkv_lora")

    def __init__(self, ln1, qkv_w, qkv_b, out, ln2, l1, l2, qkv_lora=None):
        self.ln1, self.qkv_w, self.qkv_b, self.out, self.ln2, self.l1, self.l2 = ln1, qkv_w, qkv_b, out, ln2, l1, l2
        self.qkv_lora = qkv_lora      # loralib.MergedLinear with r > 0 (--lora_pos Attention): adapters on q / k / v instead of the FFN

    def lora_params(self):
        if self.qkv_lora is not None:
            return (self.qkv_lora.lora_A, self.qkv_lora.lora_B)
        return (self.l1.lora_A, self.l1.lora_B, self.l2.lora_A, self.l2.lora_B)


class ModelSpec:
    """Geometry + parameter handles of one model family. Built by the model's `hip_spec()` on every forward (attribute
    look-ups only), so module surgery between calls — replace_ffn_with_lora, modify_head, load_state_dict — is picked up.
      ViT_face      (vit_pytorch_face/vit_face.py:449-548): Linear patch embedding, bias-free QKV, LN eps 1e-5,
                    scale dim^-0.5, CosFace head (s 64, m 0.35).
      ModifiedViT   (vit_pytorch_face/modified_VIT.py:5-45 over torchvision vit_b_16): conv16 patch embedding, QKV bias,
                    LN eps 1e-6, scale head_dim^-0.5, nn.Linear head with bias, the label argument is ignored."""
    __slots__ = ("patch_size", "num_tokens", "dim", "heads", "attn_scale", "ln_eps", "dropout_p", "emb_dropout_p", "lora_rank",
                 "patch_w", "patch_is_conv", "patch_b", "cls", "pos", "blocks", "final_ln", "head_kind", "head_w", "head_b",
                 "cos_s", "cos_m", "lora_site", "pool")

    def __init__(self, **kw):
        kw.setdefault("lora_site", "ffn")      # "ffn" (GS-LoRA) or "attention" (--lora_pos Attention ablation)
        kw.setdefault("pool", "cls")           # "cls" (token 0) or "mean" (mean over the tokens, vit_face.py:540)
        for k in self.__slots__:
            setattr(self, k, kw[k])


class LoraBucket:
    """Flat storage for the trainable LoRA tensors of one model."""

    def __init__(self, layers):
        # layers: list of (A1, B1, A2, B2) nn.Parameters per transformer block
        self.params = [p for grp in layers for p in grp]
        self.groups = [i for i, grp in enumerate(layers) for _ in grp]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                self.offsets.append(off)
                off += k
        self.grad_views = [self.grad[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.params)]
        self.toff = torch.tensor(self.offsets, device=dev, dtype=torch.int64)
        self.tnumel = torch.tensor([p.numel() for p in self.params], device=dev, dtype=torch.int64)
        self.tgroup_block = torch.tensor(self.groups, device=dev, dtype=torch.int32)
        self.ngroups_block = len(layers)
        self.per_layer = len(layers[0])
        self._gtables = {}

    def valid(self):
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def group_table(self, group_type="block"):
        """tensor->group ids for engine.get_structure_loss groupings (engine.py:585-650)."""
        L_ = self.ngroups_block
        if group_type == "block" or self.per_layer != 4:      # attention adapters: always one (A, B) group per block (engine.py:651-656)
            return self.tgroup_block, L_
        if group_type in self._gtables:
            return self._gtables[group_type]
        ids = []
        for i in range(L_):
            if group_type == "lora":
                ids += [i, i, L_ + i, L_ + i]
            elif group_type == "matrix":
                ids += [i, L_ + i, 2 * L_ + i, 3 * L_ + i]
            else:
                raise ValueError(f"unknown group type {group_type}")
        n = 2 * L_ if group_type == "lora" else 4 * L_
        self._gtables[group_type] = (torch.tensor(ids, device=self.flat.device, dtype=torch.int32), n)
        return self._gtables[group_type]

    def attach_grads(self):
        """Give every LoRA parameter its view of the flat gradient bucket. Returns True when the
        bucket had to be (re)attached, i.e. this is the first backward since zero_grad()."""
        fresh = False
        for p, g in zip(self.params, self.grad_views):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                fresh = True
                break
        if fresh:
            self.grad.zero_()
            for p, g in zip(self.params, self.grad_views):
                p.grad = g
        return fresh


class ViTRunner:
    def __init__(self, model):
        self.model = model
        self.bucket = None
        self._wcache = {}
        self._lcache = {}
        self._packs, self._pack_tables, self._retired = {}, {}, []
        self._rank = 0
        self.seed_dev = None      # int64 [1] device tensor: dropout seed of a step that is being captured / replayed as a HIP graph
        # dropout stream: torch.manual_seed() selects it (like the reference's nn.Dropout), and every data-parallel rank draws its own
        self.drop_seed = self._initial_drop_seed()
        self.drop_calls = 0
        self.grad_hook = None     # callable(layer) invoked when the LoRA gradients of `layer` are complete (data-parallel overlap, step.py)
        # fp16 operands: {S, 1/S, largest scaled gradient the LayerNorm backwards of the last backward saw, exponent in use} — device-resident
        # state of the loss scale and its overflow guard (gsl_head_bwd); persists across steps and HIP-graph replays
        self.gscale = None
        self._guard_on = False

    def __deepcopy__(self, memo):   # copies of the model build their own runner lazily
        return None

    def _loss_scale_state(self, device):
        if self.gscale is None or self.gscale.device != device:
            self.gscale = torch.zeros(4, device=device, dtype=torch.float32)
        return self.gscale

    def overflow_guard(self):
        """The device float FusedAdamW checks before it updates (None unless the last backward ran on loss-scaled fp16 gradients)."""
        return self.gscale[2:] if (self._guard_on and self.gscale is not None) else None

    def loss_scale_report(self):
        """Host read (one sync) of the loss-scale state after a backward: S, the exponent in use, the largest scaled gradient the LayerNorm
        backwards saw and the headroom 65504 / that (< = 1: a 16-bit store saturated, the optimizer skipped the step)."""
        if self.gscale is None:
            return None
        S, _, seen, E = self.gscale.tolist()
        return {"S": S, "exponent": int(E), "seen_max": seen, "headroom": (65504.0 / seen) if seen > 0 else float("inf"), "saturated": not seen < 65504.0}

    @staticmethod
    def _initial_drop_seed():
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        z = (torch.initial_seed() * 0x9E3779B97F4A7C15 + (rank + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        return int(z & 0x7FFFFFFFFFF)      # 43 bits: (seed << 20) + call counter stays below 2^63

    # ------------------------------------------------------------------ caches
    def _cached(self, cache, key, param, fn):
        ent = cache.get(key)
        tag = (param.data_ptr(), param._version, param.device)
        if ent is None or ent[0] != tag:
            with torch.no_grad():
                ent = (tag, fn(param.detach()))
            cache[key] = ent
        return ent[1]

    def invalidate_operand_caches(self):
        """Forget every cached operand-format copy of a frozen weight (they are rebuilt on the next forward). Captured HIP graphs hold the old
        copies' addresses: the graph stepper re-captures when the parameter versions it recorded change; after a `.data` write bump them too
        or drop the graphs (GraphedStep.graphs.clear())."""
        self._wcache = {k: v for k, v in self._wcache.items() if k and k[0] == "zeros"}
        self._lcache.clear()

    def w(self, name, param, dtype):
        """[N,K] operand in compute dtype."""
        if dtype == torch.float32:
            return param.detach()
        return self._cached(self._wcache, (name, "n", dtype), param, lambda p: ops.cast(p.contiguous(), dtype))

    def w_ln(self, name, weight, gamma, beta, bias, dtype):
        """Operands of a GEMM with a consumer-side LayerNorm in front (EPI_STORE_LN): W' = W * gamma along K in the operand format,
        c = rowsum(W') of the ROUNDED W' (so that the mean term cancels against what the matrix cores actually multiply), d = W beta (+ bias).
        Cached on the four parameters' versions (all frozen in GS-LoRA: built once)."""
        key = (name, "ln", dtype)
        ps = [weight, gamma, beta] + ([bias] if bias is not None else [])
        tag = tuple((p.data_ptr(), p._version, p.device) for p in ps)
        ent = self._wcache.get(key)
        if ent is None or ent[0] != tag:
            with torch.no_grad():
                w32 = weight.detach().float()
                wf = ops.cast((w32 * gamma.detach().float()[None, :]).contiguous(), dtype)
                c = wf.float().sum(1).contiguous()
                d = (w32 @ beta.detach().float()) + (bias.detach().float() if bias is not None else 0.0)
                ent = (tag, (wf, c, d.contiguous()))
            self._wcache[key] = ent
        return ent[1]

    def w_conv(self, name, param, dtype):
        """conv_proj weight [D, C, p, p] as the [D, p*p*C] operand matching gsl_patchify's (p1 p2 c) feature order."""
        def build(p):
            w2 = p.permute(0, 2, 3, 1).reshape(p.shape[0], -1).contiguous()
            return w2 if dtype == torch.float32 else ops.cast(w2, dtype)
        return self._cached(self._wcache, (name, "conv", dtype), param, build)

    def wT(self, name, param, dtype):
        """[K,N] transposed operand (dX GEMMs)."""
        return self._cached(self._wcache, (name, "t", dtype), param, lambda p: ops.transpose_cast(p.contiguous(), dtype))

    PACK_GEOM = {   # kind -> (si, sj, rows, cols, rows_out, ld_out) of gsl_pack_pad as functions of the LoRA tensor's (rows, cols, r)
        "A_rows": lambda R, C, r: (C, 1, r, C, PADK, C),        # [64, K]  rows j<r = A[j,:]
        "B_cols": lambda R, C, r: (r, 1, R, r, R, PADK),        # [N, 64]  cols j<r = B[:,j]
        "BT_rows": lambda R, C, r: (1, r, r, R, PADK, R),       # [64, N]  out[j, n] = B[n, j]
        "AT_cols": lambda R, C, r: (1, C, C, r, C, PADK),       # [K, 64]  out[k, j] = A[j, k]
        # operands of the in-kernel LoRA GEMM (gsl_gemm_nt_lora): P [16, K], Q [N, 32]
        "A_rows16": lambda R, C, r: (C, 1, r, C, 16, C),
        "B_cols32": lambda R, C, r: (r, 1, R, r, R, 32),
        "BT_rows16": lambda R, C, r: (1, r, r, R, 16, R),
        "AT_cols32": lambda R, C, r: (1, C, C, r, C, 32),
    }

    def lora_pack(self, name, param, kind, dtype):
        """Padded / transposed compute-dtype copy of one LoRA tensor. The output buffers are persistent and registered in a device
        descriptor table: after the first step, refresh_lora_packs() rebuilds ALL of them with one launch per step."""
        key = (name, kind, dtype)
        ent = self._packs.get(key)
        if ent is None or ent["ptr"] != param.data_ptr() or ent["dev"] != param.device:
            rows, cols = param.shape
            si, sj, pr, pc, ro, ld = self.PACK_GEOM[kind](rows, cols, min(rows, cols))
            out = torch.empty(ro, ld, device=param.device, dtype=dtype)
            ent = dict(ptr=param.data_ptr(), dev=param.device, param=param, out=out, geom=(si, sj, pr, pc), version=None)
            self._packs[key] = ent
            self._pack_tables.pop(dtype, None)
        if ent["version"] != param._version:
            si, sj, pr, pc = ent["geom"]
            with torch.no_grad():
                L.check(L.load().gsl_pack_pad(param.data_ptr(), si, sj, pr, pc, 1.0, ent["out"].data_ptr(), ent["out"].shape[0],
                                              ent["out"].shape[1], ops.code(dtype), ops._stream()), "gsl_pack_pad")
            ent["version"] = param._version
        return ent["out"]

    def qkv_lora_ops(self, i, ml, dtype):
        """Operands of the three q / k / v adapters of one MergedLinear, as ONE LoRA K segment (r3 = 3r <= 64 live columns):
          A_rows [64, dim]      rows g*r+j = A_g[j, :]                       (u = s * xn A_all^T)
          Bblk  [3*inner, 64]   row g*inner+n, column g*r+j = B_g[n, j]      (qkv += u Bblk^T: block diagonal)
          BblkT [64, 3*inner]                                                 (v = s * dqkv Bblk)
          AT    [dim, 64]       column g*r+j = A_g[j, :]                      (dxn1 += v A_all)"""
        r, ng = ml.r, len(ml.enable_lora)
        if ng * r > PADK:
            raise NotImplementedError("gs-lora_amd: 3 * lora_rank must not exceed 64 for --lora_pos Attention")

        def build(_):
            A, B = ml.lora_A.detach(), ml.lora_B.detach()
            inner = B.shape[0] // ng
            a_rows = torch.zeros(PADK, A.shape[1], device=A.device, dtype=torch.float32)
            a_rows[:ng * r] = A
            bblk = torch.zeros(B.shape[0], PADK, device=A.device, dtype=torch.float32)
            for g in range(ng):
                bblk[g * inner:(g + 1) * inner, g * r:(g + 1) * r] = B[g * inner:(g + 1) * inner]
            cast = (lambda t: t.contiguous()) if dtype == torch.float32 else (lambda t: ops.cast(t.contiguous(), dtype))
            return dict(A_rows=cast(a_rows), Bblk=cast(bblk), BblkT=cast(bblk.t()), AT=cast(a_rows.t()))
        key = (f"qkvlora{i}", dtype)
        ent = self._lcache.get(key)
        tag = (ml.lora_A.data_ptr(), ml.lora_A._version, ml.lora_B._version, ml.lora_A.device)
        if ent is None or ent[0] != tag:
            with torch.no_grad():
                ent = (tag, build(None))
            self._lcache[key] = ent
        return ent[1]

    def refresh_lora_packs(self, dtype):
        """One launch for every registered pack whose source changed (the optimizer touches all LoRA tensors each step)."""
        ents = [e for k, e in self._packs.items() if k[2] == dtype]
        if len(ents) < 2 or all(e["version"] == e["param"]._version for e in ents):
            return
        tab = self._pack_tables.get(dtype)
        if tab is None or tab[2] != len(ents):
            if torch.cuda.is_current_stream_capturing():
                return      # no H2D copy inside a capture: lora_pack() refreshes tensor by tensor (build_pack_tables() avoids this)
            tab = self.build_pack_tables(dtype)
        ops.pack_pad_batch(tab[0], tab[2], tab[1], dtype)
        for e in ents:
            e["version"] = e["param"]._version

    def build_pack_tables(self, dtype):
        """Device descriptor table of every registered pack of `dtype` (called at the end of an eager backward, so that a following
        HIP-graph capture finds it ready)."""
        ents = [e for k, e in self._packs.items() if k[2] == dtype]
        tab = self._pack_tables.get(dtype)
        if ents and (tab is None or tab[2] != len(ents)):
            if tab is not None:
                self._retired.append(tab[0])      # a captured HIP graph may still launch with the old table: never free it
            t, mx = ops.pack_desc_table([(e["param"], *e["geom"], 1.0, e["out"]) for e in ents], ents[0]["dev"])
            tab = self._pack_tables[dtype] = (t, mx, len(ents))
        return tab

    def _zeros(self, n, dev):
        z = self._wcache.get(("zeros", n, dev))
        if z is None:
            z = self._wcache[("zeros", n, dev)] = torch.zeros(n, device=dev, dtype=torch.float32)
        return z

    def lora_in_kernel(self, dtype, rows, N=None):
        """The bf16 wide GEMMs compute the LoRA down-projection inside the kernel (no extra pass over the activation): on the 256x256
        8-phase kernel from INK_MIN_ROWS rows on, and on the 64x64 ring kernel wherever gsl_gemm_nt_lora picks it (few rows: the
        launch-bound regime, where the separate skinny GEMM is a 6 - 12 us launch per adapted layer and direction). In between, the
        N = 512 GEMMs would run the 8-phase kernel on a handful of workgroups and the two-launch K-segment form wins."""
        if dtype not in OP16 or self._rank > 16:
            return False
        if rows >= INK_MIN_ROWS:
            return True
        if N is None or not INK_SMALL:
            return False
        t256 = ((rows + 255) // 256) * ((N + 255) // 256)
        t128 = ((rows + 127) // 128) * ((N + 127) // 128)
        return (rows < 1024 or t256 < 128) and t128 <= 256      # the tile rule of gsl_gemm_nt_lora (csrc/gemm.hip)

    def ensure_bucket(self, spec=None):
        spec = spec or self.model.hip_spec()
        self._rank = spec.lora_rank
        if spec.lora_rank <= 0:
            return None
        layers = [blk.lora_params() for blk in spec.blocks]
        if self.bucket is None or not self.bucket.valid() or any(a is not b for a, b in zip(self.bucket.params, (p for g in layers for p in g))):
            self.bucket = LoraBucket(layers)
            self._lcache.clear()
            self._retired.extend(t[0] for t in self._pack_tables.values())
            self._retired.extend(e["out"] for e in self._packs.values())
            self._packs.clear()
            self._pack_tables.clear()
        return self.bucket

    # ------------------------------------------------------------------ forward
    def forward(self, img, label, save):
        """img: [B, C, H, W], or a tuple of such batches that are processed as ONE batch (gs_lora_step hands over the remain and the
        forget batch this way: each is patchified into its row range of the token matrix, no concatenated image copy is made)."""
        m = self.model
        parts = [t.float().contiguous() for t in img] if isinstance(img, (tuple, list)) else [img.float().contiguous()]
        img = parts[0]
        if not all(t.is_cuda for t in parts):
            raise RuntimeError(f"{type(m).__name__} (gs-lora_amd): the model runs only on a ROCm GPU through libgslora_hip.so; "
                               "there is no CPU fallback. Move the model and inputs to 'cuda'.")
        if any(t.shape[1:] != img.shape[1:] for t in parts):
            raise ValueError("the batches of one forward must share the image shape")
        L.load()
        sp = m.hip_spec()
        dt = m.compute_dtype
        linear_head = sp.head_kind == "linear"
        if linear_head:
            label = None                       # modified_VIT.py:23-24: "label is not used in this model"
        elif label is not None:
            label = label.to(device=img.device, dtype=torch.int64).contiguous()
        B = sum(t.shape[0] for t in parts)
        T, D, H = sp.num_tokens, sp.dim, sp.heads
        M = B * T
        training = m.training
        p_drop = sp.dropout_p if training else 0.0
        p_emb = sp.emb_dropout_p if training else 0.0
        self.drop_calls += 1
        if self.seed_dev is not None:      # HIP-graph mode: the kernels read the seed from device memory; one captured increment per forward
            self.seed_dev.add_(1)
            seed, sflag = self.seed_dev.data_ptr(), L.SEED_ON_DEVICE
        else:
            seed, sflag = (self.drop_seed << 20) + self.drop_calls, 0
        self.ensure_bucket(sp)
        r = sp.lora_rank
        attn_site = r > 0 and sp.lora_site == "attention"
        if r > 0 and not attn_site:
            self.refresh_lora_packs(dt)
        s_lora = (1.0 / r) if r > 0 else 0.0
        eps = sp.ln_eps

        patches = ops.patchify(parts, sp.patch_size, dt)
        xbf = dt in OP16 and FWD_STREAM != "f32"        # the residual stream in 2 bytes per element
        xf16 = xbf and (FWD_STREAM == "f16" or dt == torch.float16)      # (fp16 operands: the 16-bit stream is fp16 too)
        xdt = (torch.float16 if xf16 else torch.bfloat16) if xbf else torch.float32            # dtype of the residual stream
        epi_res = (L.EPI_BIAS_RES_F16 if xf16 else L.EPI_BIAS_RES_BF16) if xbf else L.EPI_BIAS_RES_F32
        x = torch.empty(M, D, device=img.device, dtype=xdt)
        pw = self.w_conv("pe", sp.patch_w, dt) if sp.patch_is_conv else self.w("pe", sp.patch_w, dt)
        ops.gemm_nt(patches, pw, x, epilogue=(L.EPI_PATCH_F16 if xf16 else L.EPI_PATCH_BF16) if xbf else L.EPI_PATCH, bias=sp.patch_b.detach(),
                    pos=sp.pos.detach()[0, :T].contiguous(), cls=sp.cls.detach().reshape(-1), T=T,
                    p_drop=p_emb, seed=seed, site=SITE_EMB | sflag)
        del patches
        stash = []
        for i, blk in enumerate(sp.blocks):
            n1, n2 = blk.ln1, blk.ln2
            attn_lora_live = attn_site and not blk.qkv_lora.merged      # the q / k / v adapters read LN1's output: no fold
            fold = LN1_FOLD and dt in OP16 and x.dtype == dt and not attn_lora_live
            if fold:
                mean1, rstd1 = ops.layernorm_stats(x, D, M, D, n1.weight.detach(), n1.bias.detach(), eps, dt)
                xn = None
            else:
                xn, mean1, rstd1 = ops.layernorm_fwd(x, D, M, D, n1.weight.detach(), n1.bias.detach(), eps, dt)
            tail = TAIL_CLS and i == len(sp.blocks) - 1 and sp.pool == "cls"
            qsplit = tail and QSPLIT and not attn_site
            inner = H * 64
            hm = 1 if (QKV_HEAD_MAJOR and dt in OP16) else 0
            epi_qkv = L.EPI_STORE_QKV_HM if hm else L.EPI_STORE
            uq = q_cls = None
            if qsplit and fold:      # the same two GEMMs on the stream itself (W', c, d sliced like the weight; the cls rows' statistics gathered)
                wf, cq, dq_ = self.w_ln(f"qkv{i}", blk.qkv_w, n1.weight, n1.bias, blk.qkv_b, dt)
                qkv = torch.empty(M, 2 * inner, device=img.device, dtype=dt)
                ops.gemm_nt(x, wf[inner:], qkv, epilogue=L.EPI_STORE_LN, pos=mean1, cls=rstd1, aux=cq[inner:], bias=dq_[inner:])
                q_cls = torch.empty(B, inner, device=img.device, dtype=dt)
                ops.gemm_nt(x.view(B, T * D)[:, :D], wf[:inner], q_cls, epilogue=L.EPI_STORE_LN, T=T, pos=mean1, cls=rstd1,      # (T: the cls rows' statistics, T apart)
                            aux=cq[:inner], bias=dq_[:inner])
                hm = 2
            elif qsplit:      # K and V for every token, Q for the cls rows only (rows inner .. 3*inner of the fused weight are K | V)
                wq = self.w(f"qkv{i}", blk.qkv_w, dt)
                qb_ = None if blk.qkv_b is None else blk.qkv_b.detach()
                qkv = torch.empty(M, 2 * inner, device=img.device, dtype=dt)
                ops.gemm_nt(xn, wq[inner:], qkv, bias=None if qb_ is None else qb_[inner:])
                q_cls = torch.empty(B, inner, device=img.device, dtype=dt)
                ops.gemm_nt(xn.view(B, T * D)[:, :D], wq[:inner], q_cls, bias=None if qb_ is None else qb_[:inner].contiguous())      # A = the cls rows, T*D apart
                hm = 2
            elif attn_site and not blk.qkv_lora.merged:      # q / k / v adapters: one block-diagonal LoRA K segment
                qkv = torch.empty(M, 3 * inner, device=img.device, dtype=dt)
                qo = self.qkv_lora_ops(i, blk.qkv_lora, dt)
                uq = torch.empty(M, PADK, device=img.device, dtype=dt)
                ops.gemm_nt(xn, qo["A_rows"], uq, alpha=s_lora)
                ops.gemm_nt(xn, self.w(f"qkv{i}", blk.qkv_w, dt), qkv, A2=uq, W2=qo["Bblk"], epilogue=epi_qkv, T=T,
                            bias=None if blk.qkv_b is None else blk.qkv_b.detach())
            elif fold:
                wf, cq, dq_ = self.w_ln(f"qkv{i}", blk.qkv_w, n1.weight, n1.bias, blk.qkv_b, dt)
                qkv = torch.empty(M, 3 * inner, device=img.device, dtype=dt)
                ops.gemm_nt(x, wf, qkv, epilogue=L.EPI_STORE_QKV_HM_LN if hm else L.EPI_STORE_LN, T=T, pos=mean1, cls=rstd1, aux=cq, bias=dq_)
            else:
                qkv = torch.empty(M, 3 * inner, device=img.device, dtype=dt)
                ops.gemm_nt(xn, self.w(f"qkv{i}", blk.qkv_w, dt), qkv, epilogue=epi_qkv, T=T,
                            bias=None if blk.qkv_b is None else blk.qkv_b.detach())
            xn_keep = xn if (attn_site and save) else None
            del xn
            if tail:      # only the cls query of the last block is ever consumed: B rows from here on
                o, lse = ops.attention_fwd_cls(qkv, B, T, H, sp.attn_scale, layout=hm, q_cls=q_cls)
                xres, Mr = x.view(B, T, D)[:, 0].contiguous(), B
            else:
                o, lse = ops.attention_fwd(qkv, B, T, H, sp.attn_scale, layout=hm)
                xres, Mr = x, M
            x1 = torch.empty(Mr, D, device=img.device, dtype=xdt)
            ops.gemm_nt(o, self.w(f"wo{i}", blk.out.weight, dt), x1, epilogue=epi_res,
                        bias=blk.out.bias.detach(), res=xres, p_drop=p_drop, seed=seed, site=(4 * i) | sflag)
            del xres
            l1, l2 = blk.l1, blk.l2
            mlp = l1.weight.shape[0]
            lora_on = r > 0 and not attn_site and not l1.merged
            # bf16 stream: LayerNorm 2 also emits u1 = s * xn2 A1^T (the LoRA K segment of FFN1) instead of a skinny GEMM that re-reads xn2
            ln_u1 = LN_LORA and lora_on and r <= 16 and D in (512, 768) and x1.dtype == torch.bfloat16 and dt == torch.bfloat16      # (bf16 operands + bf16 stream only)
            if ln_u1:
                xn2, mean2, rstd2, u1_ln = ops.layernorm_fwd_lora(x1, D, Mr, D, n2.weight.detach(), n2.bias.detach(), eps,
                                                                   self.lora_pack(f"A1_{i}", l1.lora_A, "A_rows", dt), s_lora)
            else:
                xn2, mean2, rstd2 = ops.layernorm_fwd(x1, D, Mr, D, n2.weight.detach(), n2.bias.detach(), eps, dt)
            if lora_on and (abs(l1.scaling * r - 1.0) > 1e-9 or abs(l2.scaling * r - 1.0) > 1e-9):
                raise NotImplementedError("gs-lora_amd: the fused LoRA path uses scaling = 1 / r (lora_alpha = 1, the only value GS-LoRA "
                                          f"passes); got scaling {l1.scaling} / {l2.scaling} for r = {r}")
            u1 = u2 = u1c = None
            h = torch.empty(Mr, mlp, device=img.device, dtype=dt)
            gp8 = GP8 and dt in OP16 and save and mlp % 64 == 0
            epi_gelu = L.EPI_BIAS_GELU_G8 if gp8 else L.EPI_BIAS_GELU
            gp = torch.empty(Mr, mlp, device=img.device, dtype=torch.uint8 if gp8 else dt) if save else None
            if lora_on:
                if Mr < INK_MIN_ROWS and not ln_u1 and self.lora_in_kernel(dt, Mr, mlp):
                    # few rows: u1 = s * xn2 A1^T inside the FFN1 GEMM (64x64 ring kernel) instead of a skinny launch in front of it.
                    # (At full size the K-segment form below is as fast: the 8 N-tiles of a row panel would each recompute u1.)
                    u1 = torch.empty(Mr, PADK, device=img.device, dtype=dt)
                    ops.gemm_nt_lora(xn2, self.w(f"w1_{i}", l1.weight, dt), self.lora_pack(f"A1_{i}", l1.lora_A, "A_rows16", dt),
                                     self.lora_pack(f"B1_{i}", l1.lora_B, "B_cols32", dt), s_lora, u1, h, epilogue=epi_gelu,
                                     bias=l1.bias.detach(), out2=gp, p_drop=p_drop, seed=seed, site=(4 * i + 1) | sflag, tag="ffn1")
                else:
                    if ln_u1:
                        u1 = u1_ln
                    else:
                        u1 = torch.empty(Mr, PADK, device=img.device, dtype=dt)
                        # (16-bit modes: the GEMM also writes u1's first 16 columns as a compact [M, 16] tensor — the operand form the
                        #  gradient-fused FFN2-dX epilogue reads 32 rows of with one contiguous 1 KB load; rank <= 16)
                        u1c = torch.empty(Mr, 16, device=img.device, dtype=dt) if (save and dt in OP16 and r <= 16 and Mr >= INK_MIN_ROWS) else None
                        ops.gemm_nt(xn2, self.lora_pack(f"A1_{i}", l1.lora_A, "A_rows", dt), u1, alpha=s_lora, out2=u1c)
                    ops.gemm_nt(xn2, self.w(f"w1_{i}", l1.weight, dt), h, epilogue=epi_gelu, A2=u1,
                                W2=self.lora_pack(f"B1_{i}", l1.lora_B, "B_cols", dt), bias=l1.bias.detach(), out2=gp,
                                p_drop=p_drop, seed=seed, site=(4 * i + 1) | sflag, tag="ffn1")
                u2 = torch.empty(Mr, PADK, device=img.device, dtype=dt)
                if not self.lora_in_kernel(dt, Mr, D):
                    ops.gemm_nt(h, self.lora_pack(f"A2_{i}", l2.lora_A, "A_rows", dt), u2, alpha=s_lora)
            else:
                ops.gemm_nt(xn2, self.w(f"w1_{i}", l1.weight, dt), h, epilogue=epi_gelu, bias=l1.bias.detach(),
                            out2=gp, p_drop=p_drop, seed=seed, site=(4 * i + 1) | sflag)
            x2 = torch.empty(Mr, D, device=img.device, dtype=xdt)
            if lora_on and self.lora_in_kernel(dt, Mr, D):
                ops.gemm_nt_lora(h, self.w(f"w2_{i}", l2.weight, dt), self.lora_pack(f"A2_{i}", l2.lora_A, "A_rows16", dt),
                                 self.lora_pack(f"B2_{i}", l2.lora_B, "B_cols32", dt), s_lora, u2, x2, epilogue=epi_res,
                                 bias=l2.bias.detach(), res=x1, p_drop=p_drop, seed=seed, site=(4 * i + 2) | sflag)
            else:
                ops.gemm_nt(h, self.w(f"w2_{i}", l2.weight, dt), x2, epilogue=epi_res, A2=u2,
                            W2=self.lora_pack(f"B2_{i}", l2.lora_B, "B_cols", dt) if lora_on else None,
                            bias=l2.bias.detach(), res=x1, p_drop=p_drop, seed=seed, site=(4 * i + 2) | sflag)
            if save:
                stash.append(dict(x=x, mean1=mean1, rstd1=rstd1, qkv=qkv, qkv_hm=hm, o=o, lse=lse, x1=x1, mean2=mean2, rstd2=rstd2,
                                  xn2=xn2, u1=u1, u1c=u1c, h=h, gp=gp, u2=u2, lora_on=lora_on, xn=xn_keep, uq=uq, tail=tail, q_cls=q_cls))
            x = x2
        hn = sp.final_ln
        Th = x.shape[0] // B      # rows per image of the stream that reaches the head: T, or 1 after a cls-row-only last block
        if linear_head:      # plain classifier: logits for every call, no normalisation, no margin
            Wn = sp.head_w.detach().contiguous()
            logits, emb, meanh, rstdh = ops.head_fwd(x, B, Th, D, hn.weight.detach(), hn.bias.detach(), eps, Wn, None, 1.0, 0.0,
                                                     head_bias=sp.head_b.detach(), linear=True)
        else:
            if label is None:
                Wn = None
            elif sp.head_w.requires_grad:
                Wn = ops.cosface_prep(sp.head_w.detach().contiguous())
            else:      # frozen head (GS-LoRA trains the adapters only): the row-normalised weight is computed once per weight version
                Wn = self._cached(self._wcache, ("cosface_wn",), sp.head_w, lambda p: ops.cosface_prep(p.contiguous()))
            logits, emb, meanh, rstdh = ops.head_fwd(x, B, Th, D, hn.weight.detach(), hn.bias.detach(), eps, Wn, label,
                                                     sp.cos_s, sp.cos_m, pool_mean=(sp.pool == "mean"))
        saved = None
        if save:
            saved = dict(layers=stash, x_last=x, Th=Th, meanh=meanh, rstdh=rstdh, emb=emb, Wn=Wn, B=B, seed=seed, sflag=sflag, p_drop=p_drop,
                         dt=dt, spec=sp)
        return logits, emb, saved

    # ------------------------------------------------------------------ backward
    def backward(self, saved, dlogits, demb):
        """Accumulates d(loss)/d(LoRA) into the flat gradient bucket (views are the params' .grad)."""
        sp = saved["spec"]
        bucket = self.bucket
        if bucket is None:
            return
        bucket.attach_grads()
        if sp.lora_site == "attention":
            return self._backward_attention_site(saved, dlogits, demb)
        dt = saved["dt"]
        B, seed, p_drop, sflag = saved["B"], saved["seed"], saved["p_drop"], saved["sflag"]
        T, D, H = sp.num_tokens, sp.dim, sp.heads
        r = sp.lora_rank
        s_lora = 1.0 / r
        nl = len(saved["layers"])
        hn = sp.final_ln
        linear_head = sp.head_kind == "linear"
        if dlogits is not None:
            dlogits = dlogits.contiguous().float()
        if demb is not None:
            demb = demb.contiguous().float()
        if dlogits is not None and saved["Wn"] is None:
            raise RuntimeError("backward through logits requires a forward with labels")
        # fp16 operands: the backward runs on loss-scaled gradients — gsl_head_bwd picks the power of two S on the device and writes
        # {S, 1/S} here; every LoRA-gradient reduction below multiplies by 1/S on the way out (bf16 / f32: no scaling)
        gscale = self._loss_scale_state(saved["x_last"].device) if dt == torch.float16 else None
        self._guard_on = gscale is not None
        gmax = gscale[2:] if gscale is not None else None      # the overflow guard: raised by every LayerNorm backward below
        dx, dxb = ops.head_bwd(dlogits, demb, saved["x_last"], B, saved["Th"], D, hn.weight.detach(), saved["meanh"], saved["rstdh"],
                               saved["emb"], saved["Wn"], 1.0 if linear_head else sp.cos_s, dt, p_drop=p_drop, seed=seed,
                               site=(4 * (nl - 1) + 2) | sflag, linear=linear_head, pool_mean=(sp.pool == "mean"),
                               stream_dtype=dt if (dt in OP16 and GRAD_STREAM_BF16) else torch.float32,
                               compact=(sp.pool == "cls"), gscale=gscale, target_exp=GRAD_TARGET_EXP)      # pool='cls': [B, D] cls-row gradients, nothing zero-filled
        blocks = sp.blocks
        gv = {id(p): g for p, g in zip(bucket.params, bucket.grad_views)}
        dev = dx.device
        cls_rows = lambda t, w: t.view(B, T, w)[:, 0].contiguous()     # rows b*T of a [B*T, w] tensor
        # (the 8-bit GELU' code tensor is slab-major [w/64][rows][64]: its cls rows, again slab-major for B rows)
        gp_rows = lambda t, w: (t.view(w // 64, B, T, 64)[:, :, 0].contiguous().view(B, w) if t.dtype == torch.uint8 else cls_rows(t, w))
        pending = []      # deferred LoRA-gradient reductions (launch-bound regime): (Y, U, G, gsn, gsj, r, accumulate), operands kept alive

        def lgrad(Y, U, G, gsn, gsj, rr):
            if Y.shape[0] < LGRAD_BATCH_MAX_ROWS and B * T < LGRAD_BATCH_MAX_ROWS and ops.lora_grad_batchable(Y, U, rr):
                pending.append((Y, U, G, gsn, gsj, rr, True))
            else:
                ops.lora_grad(Y, U, G, gsn, gsj, rr, gscale=gscale)

        def flush():
            ops.lora_grad_batch(pending, gscale=gscale)
            pending.clear()

        for i in reversed(range(nl)):
            st = saved["layers"][i]
            blk = blocks[i]
            l1, l2 = blk.l1, blk.l2
            mlp = l1.weight.shape[0]
            if not st["lora_on"]:
                raise RuntimeError("backward with merged LoRA weights is undefined (model.train() un-merges)")
            # The network pools x[:, 0] (vit_face.py:540): the stream gradient entering the LAST block is exactly zero
            # outside the B cls rows, so its FFN backward, LoRA-gradient reductions, LN2 backward, out-proj dX and the
            # attention backward (a rank-1 cls-query form) run on B rows instead of B*T. Exact, not an approximation.
            sparse = (i == nl - 1) and sp.pool == "cls"      # (with pool='mean' every token carries gradient: dense last block)
            tail = st["tail"]      # the forward of this block already ran on the cls rows: every saved tensor behind the attention is [B, .]
            if sparse and not tail:      # dx / dxb arrive compact ([B, D]) from the head backward
                dyb, xn2, h, gp, u1, u2 = (dxb, cls_rows(st["xn2"], D), cls_rows(st["h"], mlp), gp_rows(st["gp"], mlp),
                                           cls_rows(st["u1"], PADK), cls_rows(st["u2"], PADK))
            else:
                dyb, xn2, h, gp, u1, u2 = dxb, st["xn2"], st["h"], st["gp"], st["u1"], st["u2"]
            u1c = None if (sparse and not tail) else st.get("u1c")      # the compact [M, 16] form of u1 (16-bit modes, full-size blocks)
            Mrows = dyb.shape[0]
            # ---- FFN sub-layer: y = x1 + drop(W2' h + b2), h = drop(gelu(W1' xn2 + b1)) -------------
            ink = self.lora_in_kernel(dt, Mrows, mlp)       # FFN2-dX (N = mlp)
            ink1 = self.lora_in_kernel(dt, Mrows, D)        # FFN1-dX (N = dim)
            epi_mul = L.EPI_MUL_G8 if gp.dtype == torch.uint8 else L.EPI_MUL      # g' as the 8-bit code of the forward (decode scale from p_drop)
            v2 = torch.empty(Mrows, PADK, device=dev, dtype=dt)
            da = torch.empty(Mrows, mlp, device=dev, dtype=dt)
            fused_grads = ink and FUSE_LORA_GRAD and Mrows >= INK_MIN_ROWS      # the gradient-fused epilogue lives on the 8-phase kernel
            if fused_grads:
                # v2 = s*dy*B2 is produced inside the dX GEMM, and the two gradient reductions that contract over the rows of its
                # [M, mlp] tiles (dB1 from the da it produces, dA2 from h and the v2 it holds) ride in its epilogue
                ops.gemm_nt_lora_mulgrad(dyb, self.wT(f"w2_{i}", l2.weight, dt), self.lora_pack(f"B2_{i}", l2.lora_B, "BT_rows16", dt),
                                         self.lora_pack(f"A2_{i}", l2.lora_A, "AT_cols32", dt), s_lora, v2, da, gp,
                                         (u1c if u1c is not None else u1), gv[id(l1.lora_B)], (r, 1), h, gv[id(l2.lora_A)], (1, mlp), r, tag="ffn2dx", p_drop=p_drop,
                                         gscale=gscale)
            elif ink:    # v2 = s*dy*B2 is produced inside the dX GEMM
                ops.gemm_nt_lora(dyb, self.wT(f"w2_{i}", l2.weight, dt), self.lora_pack(f"B2_{i}", l2.lora_B, "BT_rows16", dt),
                                 self.lora_pack(f"A2_{i}", l2.lora_A, "AT_cols32", dt), s_lora, v2, da, epilogue=epi_mul, aux=gp, p_drop=p_drop)
            else:
                ops.gemm_nt(dyb, self.lora_pack(f"B2_{i}", l2.lora_B, "BT_rows", dt), v2, alpha=s_lora)
                ops.gemm_nt(dyb, self.wT(f"w2_{i}", l2.weight, dt), da, epilogue=epi_mul, A2=v2,
                            W2=self.lora_pack(f"A2_{i}", l2.lora_A, "AT_cols", dt), aux=gp, p_drop=p_drop)
            lgrad(dyb, u2, gv[id(l2.lora_B)], r, 1, r)                        # dB2[c, j]
            if not fused_grads:
                lgrad(h, v2, gv[id(l2.lora_A)], 1, mlp, r)                    # dA2[j, hid]
            v1 = torch.empty(Mrows, PADK, device=dev, dtype=dt)
            dxn2 = None
            if ink1 and i > 0:   # v1 = s*da*B1 is produced inside the FFN1-dX GEMM
                dxn2 = torch.empty(Mrows, D, device=dev, dtype=dt)
                ops.gemm_nt_lora(da, self.wT(f"w1_{i}", l1.weight, dt), self.lora_pack(f"B1_{i}", l1.lora_B, "BT_rows16", dt),
                                 self.lora_pack(f"A1_{i}", l1.lora_A, "AT_cols32", dt), s_lora, v1, dxn2)
            else:
                ops.gemm_nt(da, self.lora_pack(f"B1_{i}", l1.lora_B, "BT_rows", dt), v1, alpha=s_lora)
            if not fused_grads:
                lgrad(da, u1, gv[id(l1.lora_B)], r, 1, r)                     # dB1[hid, j]
            lgrad(xn2, v1, gv[id(l1.lora_A)], 1, D, r)                        # dA1[j, c]
            if self.grad_hook is not None:
                flush()      # the hook hands finished gradient slices to the all-reduce
                self.grad_hook(i)
            if i == 0:
                flush()
                break   # nothing below the layer-0 FFN input is trainable
            if dxn2 is None:
                dxn2 = torch.empty(Mrows, D, device=dev, dtype=dt)
                ops.gemm_nt(da, self.wT(f"w1_{i}", l1.weight, dt), dxn2, A2=v1,
                            W2=self.lora_pack(f"A1_{i}", l1.lora_A, "AT_cols", dt))
            del da, v1, v2
            n2 = blk.ln2
            if sparse and not tail:   # compact in, compact out: the cls rows of x1 are T*D apart, the dropout counters are those of the dense tensor
                dx1, dx1b = ops.layernorm_bwd(dxn2, st["x1"], T * D, n2.weight.detach(), cls_rows(st["mean2"].view(-1, 1), 1).view(-1),
                                              cls_rows(st["rstd2"].view(-1, 1), 1).view(-1), dx,
                                              p_drop=p_drop, seed=seed, site=(4 * i) | sflag, drop_row_stride=T * D, gmax=gmax)
            else:
                dx1, dx1b = ops.layernorm_bwd(dxn2, st["x1"], D, n2.weight.detach(), st["mean2"], st["rstd2"], dx,
                                              p_drop=p_drop, seed=seed, site=(4 * i) | sflag, gmax=gmax)
            del dxn2
            # ---- attention sub-layer: x1 = x + drop(Wo o + bo) -------------------------------------
            d_o = torch.empty(Mrows, H * 64, device=dev, dtype=dt)
            ops.gemm_nt(dx1b, self.wT(f"wo{i}", blk.out.weight, dt), d_o)
            dxn1 = torch.empty(B * T, D, device=dev, dtype=dt)
            if sparse and st["q_cls"] is not None:      # Q was projected for the cls rows only: dX contracts over K | V, the cls rows get dQ W_q on top
                inner = H * 64
                dkv, dq_cls = ops.attention_bwd_cls(st["qkv"], st["o"], d_o, st["lse"], B, T, H, sp.attn_scale, layout=2, q_cls=st["q_cls"])
                wt = self.wT(f"qkv{i}", blk.qkv_w, dt)                     # [dim, 3*inner]
                ops.gemm_nt(dkv, wt[:, inner:], dxn1)
                rows = dxn1.view(B, T * D)[:, :D]                          # the cls rows of dxn1, T*D apart: updated in place
                ops.gemm_nt(dq_cls, wt[:, :inner], rows, epilogue={torch.bfloat16: L.EPI_BIAS_RES_BF16, torch.float16: L.EPI_BIAS_RES_F16}.get(dt, L.EPI_BIAS_RES_F32),
                            bias=self._zeros(D, dev), res=rows)
                del dkv, dq_cls
            else:
                if sparse:
                    dqkv = ops.attention_bwd_cls(st["qkv"], st["o"], d_o, st["lse"], B, T, H, sp.attn_scale, layout=st["qkv_hm"])
                else:
                    dqkv = ops.attention_bwd(st["qkv"], st["o"], d_o, st["lse"], B, T, H, sp.attn_scale, layout=st["qkv_hm"])
                ops.gemm_nt(dqkv, self.wT(f"qkv{i}", blk.qkv_w, dt), dxn1)
                del dqkv
            del d_o, dx1b
            n1 = blk.ln1
            dx, dxb = ops.layernorm_bwd(dxn1, st["x"], D, n1.weight.detach(), st["mean1"], st["rstd1"], dx1,
                                        p_drop=p_drop, seed=seed, site=(4 * (i - 1) + 2) | sflag,
                                        dres_cls_T=T if sparse else 0, gmax=gmax)      # after the cls-row-only block dx1 is compact [B, D]
            saved["layers"][i] = None   # free this layer's activations
        if not torch.cuda.is_current_stream_capturing():
            self.build_pack_tables(dt)    # every pack of the step is registered now: the next forward refreshes them in one launch

    def _backward_attention_site(self, saved, dlogits, demb):
        """Backward when the adapters sit on the QKV projection (--lora_pos Attention; reference vit_face.py:349-355 with
        loralib.MergedLinear): the FFN is a plain frozen sub-layer (dX only), every block's attention needs its dqkv, and the chain
        stops after the LoRA gradients of block 0. Same kernels as the FFN-site path."""
        sp, bucket = saved["spec"], self.bucket
        dt = saved["dt"]
        B, seed, p_drop, sflag = saved["B"], saved["seed"], saved["p_drop"], saved["sflag"]
        T, D, H = sp.num_tokens, sp.dim, sp.heads
        r = sp.lora_rank
        s_lora = 1.0 / r
        nl = len(saved["layers"])
        hn = sp.final_ln
        linear_head = sp.head_kind == "linear"
        if dlogits is not None:
            dlogits = dlogits.contiguous().float()
        if demb is not None:
            demb = demb.contiguous().float()
        if dlogits is not None and saved["Wn"] is None:
            raise RuntimeError("backward through logits requires a forward with labels")
        # fp16 operands: the backward runs on loss-scaled gradients — gsl_head_bwd picks the power of two S on the device and writes
        # {S, 1/S} here; every LoRA-gradient reduction below multiplies by 1/S on the way out (bf16 / f32: no scaling)
        gscale = self._loss_scale_state(saved["x_last"].device) if dt == torch.float16 else None
        self._guard_on = gscale is not None
        gmax = gscale[2:] if gscale is not None else None      # the overflow guard: raised by every LayerNorm backward below
        dx, dxb = ops.head_bwd(dlogits, demb, saved["x_last"], B, saved["Th"], D, hn.weight.detach(), saved["meanh"], saved["rstdh"],
                               saved["emb"], saved["Wn"], 1.0 if linear_head else sp.cos_s, dt, p_drop=p_drop, seed=seed,
                               site=(4 * (nl - 1) + 2) | sflag, linear=linear_head, pool_mean=(sp.pool == "mean"),
                               stream_dtype=dt if (dt in OP16 and GRAD_STREAM_BF16) else torch.float32,
                               compact=(sp.pool == "cls"), gscale=gscale, target_exp=GRAD_TARGET_EXP)      # pool='cls': [B, D] cls-row gradients, nothing zero-filled
        gv = {id(p): g for p, g in zip(bucket.params, bucket.grad_views)}
        dev = dx.device
        cls_rows = lambda t, w: t.view(B, T, w)[:, 0].contiguous()
        gp_rows = lambda t, w: (t.view(w // 64, B, T, 64)[:, :, 0].contiguous().view(B, w) if t.dtype == torch.uint8 else cls_rows(t, w))
        for i in reversed(range(nl)):
            st = saved["layers"][i]
            blk = sp.blocks[i]
            l1, l2, ml = blk.l1, blk.l2, blk.qkv_lora
            if st["uq"] is None:
                raise RuntimeError("backward with merged LoRA weights is undefined (model.train() un-merges)")
            mlp = l1.weight.shape[0]
            sparse = (i == nl - 1) and sp.pool == "cls"      # only the cls rows of the last block carry gradient (see the FFN-site path)
            tail = st["tail"]
            if sparse and not tail:
                dyb, gp = dxb, gp_rows(st["gp"], mlp)
            else:
                dyb, gp = dxb, st["gp"]
            Mrows = dyb.shape[0]
            # ---- frozen FFN sub-layer: dX only
            da = torch.empty(Mrows, mlp, device=dev, dtype=dt)
            ops.gemm_nt(dyb, self.wT(f"w2_{i}", l2.weight, dt), da, epilogue=L.EPI_MUL_G8 if gp.dtype == torch.uint8 else L.EPI_MUL, aux=gp,
                        p_drop=p_drop)
            dxn2 = torch.empty(Mrows, D, device=dev, dtype=dt)
            ops.gemm_nt(da, self.wT(f"w1_{i}", l1.weight, dt), dxn2)
            del da
            n2 = blk.ln2
            if sparse and not tail:
                dx1, dx1b = ops.layernorm_bwd(dxn2, st["x1"], T * D, n2.weight.detach(), cls_rows(st["mean2"].view(-1, 1), 1).view(-1),
                                              cls_rows(st["rstd2"].view(-1, 1), 1).view(-1), dx,
                                              p_drop=p_drop, seed=seed, site=(4 * i) | sflag, drop_row_stride=T * D, gmax=gmax)
            else:
                dx1, dx1b = ops.layernorm_bwd(dxn2, st["x1"], D, n2.weight.detach(), st["mean2"], st["rstd2"], dx,
                                              p_drop=p_drop, seed=seed, site=(4 * i) | sflag, gmax=gmax)
            del dxn2
            # ---- attention sub-layer with the q / k / v adapters
            d_o = torch.empty(Mrows, H * 64, device=dev, dtype=dt)
            ops.gemm_nt(dx1b, self.wT(f"wo{i}", blk.out.weight, dt), d_o)
            if sparse:
                dqkv = ops.attention_bwd_cls(st["qkv"], st["o"], d_o, st["lse"], B, T, H, sp.attn_scale, layout=st["qkv_hm"])
            else:
                dqkv = ops.attention_bwd(st["qkv"], st["o"], d_o, st["lse"], B, T, H, sp.attn_scale, layout=st["qkv_hm"])
            del d_o, dx1b
            qo = self.qkv_lora_ops(i, ml, dt)
            v = torch.empty(B * T, PADK, device=dev, dtype=dt)
            ops.gemm_nt(dqkv, qo["BblkT"], v, alpha=s_lora)                      # v[:, g*r+j] = s * dqkv_g . B_g[:, j]
            ng, inner = len(ml.enable_lora), H * 64
            gA, gB = gv[id(ml.lora_A)], gv[id(ml.lora_B)]
            for g in range(ng):
                ops.lora_grad(st["xn"], v[:, g * r:], gA[g * r:(g + 1) * r], 1, D, r, gscale=gscale)                       # dA_g[j, c]
                ops.lora_grad(dqkv[:, g * inner:(g + 1) * inner], st["uq"][:, g * r:], gB[g * inner:(g + 1) * inner], r, 1, r, gscale=gscale)   # dB_g[n, j]
            if self.grad_hook is not None:
                self.grad_hook(i)
            if i == 0:
                break      # nothing below the block-0 QKV projection is trainable
            dxn1 = torch.empty(B * T, D, device=dev, dtype=dt)
            ops.gemm_nt(dqkv, self.wT(f"qkv{i}", blk.qkv_w, dt), dxn1, A2=v, W2=qo["AT"])
            del dqkv, v
            n1 = blk.ln1
            dx, dxb = ops.layernorm_bwd(dxn1, st["x"], D, n1.weight.detach(), st["mean1"], st["rstd1"], dx1,
                                        p_drop=p_drop, seed=seed, site=(4 * (i - 1) + 2) | sflag,
                                        dres_cls_T=T if sparse else 0, gmax=gmax)      # after the cls-row-only block dx1 is compact [B, D]
            saved["layers"][i] = None
