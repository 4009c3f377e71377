#!/usr/bin/env python
"""CPU emulation: which 16-bit format on which matrix-core operand costs how much accuracy.

Pure torch on CPU, no HIP: every matrix product of the ViT-P8S8 forward / backward goes through `qmm`, which rounds its operands to a
chosen 16-bit format and accumulates in f32 (what an MFMA with f32 accumulators does — products of two 16-bit values are exact in f32).
Forward operands, stored activations (= the next product's operand) and the backward's operands have independent formats, so the table
answers "what do fp16 forward operands buy while the backward's gradient operands stay bf16" BEFORE any kernel is touched.
The residual stream is fp16 (the round-4 default), LayerNorm / softmax / GELU arithmetic f32 as in the kernels.

Usage: python tools/emu_operand_precision.py [B]   (B images per half batch, default 8)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", 8)))
FMT = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}


def rnd(t, fmt):
    d = FMT[fmt]
    return t if d is None else t.to(d).to(torch.float32)


class QMM(torch.autograd.Function):
    """y = Qf(a) @ Qf(w)^T ; da = Qb(g) @ Qb(w) ; dw = Qb(g)^T @ Qb(a_saved)   (a_saved = the forward's rounded operand)."""

    @staticmethod
    def forward(ctx, a, w, ff, fb):
        aq, wq = rnd(a, ff), rnd(w, ff)
        ctx.save_for_backward(aq, w)
        ctx.fb = fb
        return aq @ wq.transpose(-1, -2)

    @staticmethod
    def backward(ctx, g):
        aq, w = ctx.saved_tensors
        fb = ctx.fb
        gq = rnd(g, fb)
        da = gq @ rnd(w, fb)
        dw = None
        if ctx.needs_input_grad[1]:
            g2, a2 = gq.reshape(-1, gq.shape[-1]), rnd(aq, fb).reshape(-1, aq.shape[-1])
            dw = g2.t() @ a2
        return da, dw, None, None


class QBMM(torch.autograd.Function):
    """batched a @ b with the same rounding rules (attention's two products)."""

    @staticmethod
    def forward(ctx, a, b, ff, fb):
        aq, bq = rnd(a, ff), rnd(b, ff)
        ctx.save_for_backward(aq, bq)
        ctx.fb = fb
        return aq @ bq

    @staticmethod
    def backward(ctx, g):
        aq, bq = ctx.saved_tensors
        fb = ctx.fb
        gq = rnd(g, fb)
        return gq @ rnd(bq, fb).transpose(-1, -2), rnd(aq, fb).transpose(-1, -2) @ gq, None, None


class RoundST(torch.autograd.Function):
    """a stored tensor: rounded on the way forward, its gradient rounded on the way back (the 16-bit gradient tensors)."""

    @staticmethod
    def forward(ctx, t, ff, fb):
        ctx.fb = fb
        return rnd(t, ff)

    @staticmethod
    def backward(ctx, g):
        return rnd(g, ctx.fb), None, None


def forward(st, img, label, cfg, P):
    """P: dict gemm / attn (forward operand formats), bwd (backward operand format), stream (residual stream format)."""
    from oracle import gslora_oracle as O
    p, d, hds, r = cfg["patch_size"], cfg["dim"], cfg["heads"], cfg["lora_rank"]
    g, at, fb = P["gemm"], P["attn"], P["bwd"]
    lin = lambda x, W, b=None: QMM.apply(x, W, g, fb) + (0 if b is None else b)
    x = O.patchify(img.float(), p)
    x = lin(x, st["patch_to_embedding.weight"], st["patch_to_embedding.bias"])
    b, n, _ = x.shape
    x = torch.cat((st["cls_token"].expand(b, -1, -1), x), dim=1)
    x = RoundST.apply(x + st["pos_embedding"][:, : n + 1], P["stream"], "f32")
    scale = d ** -0.5
    for i in range(cfg["depth"]):
        a = f"transformer.layers.{i}.0.fn"
        f = f"transformer.layers.{i}.1.fn"
        xn = F.layer_norm(x, (d,), st[f"{a}.norm.weight"], st[f"{a}.norm.bias"], 1e-5)
        qkv = RoundST.apply(lin(xn, st[f"{a}.fn.to_qkv.weight"]), at, fb)      # stored in the attention kernels' operand format
        q, k, v = qkv.chunk(3, dim=-1)
        sp = lambda t: t.reshape(b, n + 1, hds, -1).permute(0, 2, 1, 3)
        q, k, v = sp(q), sp(k), sp(v)
        dots = QBMM.apply(q, k.transpose(-1, -2), at, fb) * scale
        attn = dots.softmax(dim=-1)
        o = QBMM.apply(attn, v, at, fb).permute(0, 2, 1, 3).reshape(b, n + 1, -1)
        o = RoundST.apply(o, P.get("o", at), fb)
        x = RoundST.apply(lin(o, st[f"{a}.fn.to_out.0.weight"], st[f"{a}.fn.to_out.0.bias"]) + x, P["stream"], P["gstream"])
        xn = F.layer_norm(x, (d,), st[f"{f}.norm.weight"], st[f"{f}.norm.bias"], 1e-5)
        A1, B1 = st[f"{f}.fn.net.0.lora_A"], st[f"{f}.fn.net.0.lora_B"]
        A2, B2 = st[f"{f}.fn.net.3.lora_A"], st[f"{f}.fn.net.3.lora_B"]
        u1 = lin(xn, A1) * (1.0 / r)
        h = lin(xn, st[f"{f}.fn.net.0.weight"], st[f"{f}.fn.net.0.bias"]) + lin(u1, B1)
        h = F.gelu(h)
        u2 = lin(h, A2) * (1.0 / r)
        y = lin(h, st[f"{f}.fn.net.3.weight"], st[f"{f}.fn.net.3.bias"]) + lin(u2, B2)
        x = RoundST.apply(y + x, P["stream"], P["gstream"])
    emb = F.layer_norm(x[:, 0], (d,), st["mlp_head.0.weight"], st["mlp_head.0.bias"], 1e-5)
    return O.cosface(emb, st["loss.weight"], label), emb


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from oracle import gslora_oracle as O
    torch.manual_seed(0)
    cfg = dict(patch_size=8, dim=512, heads=8, depth=6, mlp_dim=2048, lora_rank=8, num_class=100, image_size=112)
    from vit_pytorch_face import ViT_face
    m = ViT_face(loss_type="CosFace", GPU_ID=None, num_class=100, image_size=112, patch_size=8, dim=512, depth=6, heads=8, mlp_dim=2048,
                 dropout=0.0, emb_dropout=0.0, lora_rank=8)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.normal_(0, 0.02)
    st0 = {k: v.detach().clone().float() for k, v in m.state_dict().items()}
    x = torch.rand(2 * B, 3, 112, 112)
    y = torch.randint(0, 100, (2 * B,))
    proto = torch.randn(100, 512)

    def run(P):
        st = {k: v.clone().requires_grad_("lora_" in k) for k, v in st0.items()}
        lo, em = forward(st, x, y, cfg, P)
        ce_r = F.cross_entropy(lo[:B], y[:B])
        ce_f = F.cross_entropy(lo[B:], y[B:])
        kl = O.prototype_kl(em[:B], y[:B], proto)
        total = 0.15 * torch.relu(105.0 - ce_f) + ce_r + 1e-4 * O.structure_loss(st, cfg) + 0.05 * kl
        total.backward()
        grads = {k: v.grad.reshape(-1).clone() for k, v in st.items() if v.requires_grad}
        return lo.detach(), em.detach(), grads, float(total)

    ref = run(dict(gemm="f32", attn="f32", bwd="f32", stream="f32", gstream="f32"))
    CONFIGS = [
        ("today: bf16 operands everywhere, fp16 stream, bf16 gradient stream", dict(gemm="bf16", attn="bf16", bwd="bf16", stream="f16", gstream="bf16")),
        ("fp16 forward operands (GEMMs + attention), bf16 backward", dict(gemm="f16", attn="f16", bwd="bf16", stream="f16", gstream="bf16")),
        ("fp16 forward GEMM operands, attention bf16 (q/k/v/p/o), bf16 backward", dict(gemm="f16", attn="bf16", bwd="bf16", stream="f16", gstream="bf16")),
        ("fp16 forward GEMMs + attention, o stored bf16", dict(gemm="f16", attn="f16", o="bf16", bwd="bf16", stream="f16", gstream="bf16")),
        ("fp16 forward operands, f32 backward (what the forward alone leaves)", dict(gemm="f16", attn="f16", bwd="f32", stream="f16", gstream="f32")),
        ("bf16 forward operands, f32 backward (the backward's share today)", dict(gemm="bf16", attn="bf16", bwd="f32", stream="f16", gstream="f32")),
        ("f32 forward, bf16 backward", dict(gemm="f32", attn="f32", bwd="bf16", stream="f32", gstream="bf16")),
        ("fp16 forward AND fp16 backward operands (no loss scale)", dict(gemm="f16", attn="f16", bwd="f16", stream="f16", gstream="f16")),
    ]
    print(f"CPU emulation, FULL ViT-P8S8, batch {B}+{B}, against f32 (loss {ref[3]:.5f})\n")
    print("| configuration | logits max abs (scale 64) | emb max abs | loss | LoRA grad rel. Frobenius | cosine | worst tensor rel. | top-1 flips |")
    print("|---|---|---|---|---|---|---|---|")
    for name, P in CONFIGS:
        lo, em, gr, tot = run(P)
        a = torch.cat([gr[k] for k in sorted(gr)])
        b = torch.cat([ref[2][k] for k in sorted(gr)])
        rel = float((a - b).norm() / b.norm())
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        worst = max(float((gr[k] - ref[2][k]).norm() / ref[2][k].norm().clamp_min(1e-30)) for k in gr)
        flips = int((lo.argmax(1) != ref[0].argmax(1)).sum())
        print(f"| {name} | {float((lo - ref[0]).abs().max()):.4f} | {float((em - ref[1]).abs().max()):.4f} | {tot:.5f} | {rel:.5f} | {cos:.6f} | {worst:.4f} | {flips} |",
              flush=True)


if __name__ == "__main__":
    main()
