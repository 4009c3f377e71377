"""Loss terms of the GS-LoRA step as autograd nodes over the HIP kernels.

  ce_sum_top1      — nn.CrossEntropyLoss (sum form) + top-1 count     (engine_cl.py:65-78)
  proto_kl_sum     — get_prototype_loss 'kl' (sum form)                (engine_cl.py:571-603)
  structure_loss   — group-lasso over LoRA groups                      (engine_cl.py:349-432)
Upstream gradients arrive as 0-dim device tensors and are handed to the kernels as device
pointers, so the whole loss graph runs without a host sync.
"""
import torch

from . import ops


class _CESum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.contiguous().float()
        labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        out = ops.ce_fwd(logits, labels)
        ctx.save_for_backward(logits, labels)
        ctx.mark_non_differentiable(out[1])
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_sum, _g_cnt):
        logits, labels = ctx.saved_tensors
        coef = g_sum.reshape(1).float().contiguous()
        return ops.ce_bwd(logits, labels, coef, 1.0), None


def ce_sum_top1(logits, labels):
    """-> (sum_i CE_i, number of top-1 hits), both 0-dim f32 device tensors."""
    return _CESum.apply(logits, labels)


class _CESumSplit(torch.autograd.Function):
    """ce_sum_top1 of the remain rows [0, nr) and the forget rows [nr, N) of ONE logits tensor (the step runs both batches as one
    forward). Backward writes the two row ranges of a single dlogits buffer — slicing the logits in Python instead would make
    autograd build two zero-filled full-size gradients and add them (10 extra one-off kernels per step)."""

    @staticmethod
    def forward(ctx, logits, labels, nr):
        logits = logits.contiguous().float()
        labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        out_r, out_f = ops.ce_fwd(logits[:nr], labels[:nr]), ops.ce_fwd(logits[nr:], labels[nr:])
        ctx.save_for_backward(logits, labels)
        ctx.nr = nr
        res = (out_r[0], out_r[1], out_f[0], out_f[1])
        ctx.mark_non_differentiable(res[1], res[3])
        return res

    @staticmethod
    def backward(ctx, g_r, _c_r, g_f, _c_f):
        logits, labels = ctx.saved_tensors
        nr = ctx.nr
        dl = torch.empty_like(logits)
        for g, sl in ((g_r, slice(0, nr)), (g_f, slice(nr, None))):
            if g is None:
                dl[sl].zero_()
            else:
                ops.ce_bwd(logits[sl], labels[sl], g.reshape(1).float().contiguous(), 1.0, dlogits=dl[sl], accumulate=False)
        return dl, None, None


def ce_sum_top1_split(logits, labels, nr):
    """-> (CE sum, top-1 hits) of rows [0, nr) and of rows [nr, N): four 0-dim f32 device tensors."""
    return _CESumSplit.apply(logits, labels, int(nr))


class _ProtoKLSumSplit(torch.autograd.Function):
    """proto_kl_sum of rows [nr, N) (forget) and rows [0, nr) (remain) of one embedding tensor; one demb buffer in backward."""

    @staticmethod
    def forward(ctx, emb, labels, table, nr):
        emb = emb.contiguous().float()
        labels = labels.to(device=emb.device, dtype=torch.int64).contiguous()
        ctx.save_for_backward(emb, labels, table)
        ctx.nr = nr
        return ops.proto_kl_fwd(emb[nr:], labels[nr:], table)[0], ops.proto_kl_fwd(emb[:nr], labels[:nr], table)[0]

    @staticmethod
    def backward(ctx, g_f, g_r):
        emb, labels, table = ctx.saved_tensors
        nr = ctx.nr
        de = torch.empty_like(emb)
        for g, sl in ((g_r, slice(0, nr)), (g_f, slice(nr, None))):
            if g is None:
                de[sl].zero_()
            else:
                ops.proto_kl_bwd(emb[sl], labels[sl], table, g.reshape(1).float().contiguous(), 1.0, demb=de[sl], accumulate=False)
        return de, None, None, None


def proto_kl_sum_split(emb, labels, table, nr):
    """-> (KL sum of the forget rows [nr, N), KL sum of the remain rows [0, nr))."""
    return _ProtoKLSumSplit.apply(emb, labels, table, int(nr))


class _ProtoKLSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, labels, table):
        emb = emb.contiguous().float()
        labels = labels.to(device=emb.device, dtype=torch.int64).contiguous()
        ctx.save_for_backward(emb, labels, table)
        return ops.proto_kl_fwd(emb, labels, table)[0]

    @staticmethod
    def backward(ctx, g):
        emb, labels, table = ctx.saved_tensors
        return ops.proto_kl_bwd(emb, labels, table, g.reshape(1).float().contiguous(), 1.0), None, None


def proto_kl_sum(emb, labels, table):
    return _ProtoKLSum.apply(emb, labels, table)


class _Combine(torch.autograd.Function):
    """total loss of the step from the five batch sums; one kernel forward, one multiply backward (see gsl_loss_combine)."""

    @staticmethod
    def forward(ctx, ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r, hit_f, n_r, n_f, beta, BND, alpha, w_f, w_r, BND_pro):
        f = lambda t: None if t is None else t.detach().float().contiguous()
        total, meters, coefs = ops.loss_combine(f(ce_r_sum), f(ce_f_sum), f(kl_f_sum), f(kl_r_sum), f(structure), f(hit_r), f(hit_f),
                                                n_r, n_f, beta, BND, alpha, w_f, w_r, BND_pro)
        ctx.save_for_backward(coefs)
        ctx.has = (kl_f_sum is not None, kl_r_sum is not None, structure is not None)
        ctx.mark_non_differentiable(meters)
        return total, meters

    @staticmethod
    def backward(ctx, g, _gm):
        (coefs,) = ctx.saved_tensors
        gc = coefs * g                      # one 5-element kernel; the entries below are views
        has_f, has_r, has_s = ctx.has
        return (gc[0], gc[1], gc[2] if has_f else None, gc[3] if has_r else None, gc[4] if has_s else None) + (None,) * 10


def combine(ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r, hit_f, n_r, n_f, beta, BND, alpha, w_f, w_r, BND_pro):
    return _Combine.apply(ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r, hit_f, n_r, n_f, beta, BND, alpha, w_f, w_r,
                          BND_pro)


class _CombinePack(torch.autograd.Function):
    """Data-parallel scalar tail: value from the all-reduced pack (global batch sums and sizes), gradient with respect to this rank's
    local sums (d global sum / d local sum = 1): one kernel forward, one 5-element multiply backward (gsl_loss_combine_pack)."""

    @staticmethod
    def forward(ctx, pack, ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, beta, BND, alpha, w_f, w_r, BND_pro):
        has_proto = kl_f_sum is not None
        total, meters, coefs = ops.loss_combine_pack(pack.detach().float().contiguous(),
                                                     None if structure is None else structure.detach().float().contiguous(),
                                                     has_proto, beta, BND, alpha, w_f, w_r, BND_pro)
        ctx.save_for_backward(coefs)
        ctx.has = (has_proto, structure is not None)
        ctx.mark_non_differentiable(meters)
        return total, meters

    @staticmethod
    def backward(ctx, g, _gm):
        (coefs,) = ctx.saved_tensors
        gc = coefs * g
        has_p, has_s = ctx.has
        return (None, gc[0], gc[1], gc[2] if has_p else None, gc[3] if has_p else None, gc[4] if has_s else None) + (None,) * 6


def combine_pack(pack, ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, beta, BND, alpha, w_f, w_r, BND_pro):
    return _CombinePack.apply(pack, ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, beta, BND, alpha, w_f, w_r, BND_pro)


_table_cache = {}


def prototype_table(prototype_dict, device, dim=None):
    """dict{int -> [D] tensor} (util.utils.calculate_prototypes) -> dense [C, D] f32 device table."""
    key = (id(prototype_dict), str(device))
    ent = _table_cache.get(key)
    if ent is not None and ent[0] is prototype_dict:
        return ent[1]
    C = max(int(k) for k in prototype_dict) + 1
    any_v = next(iter(prototype_dict.values()))
    # a class without a prototype holds NaN: the reference raises KeyError when a batch label misses (engine_cl.py:587-589); here the
    # look-up happens on the device without a host sync, so the loss (and the meters) turn NaN instead of silently using zeros
    table = torch.full((C, any_v.numel()), float("nan"), dtype=torch.float32)
    for k, v in prototype_dict.items():
        table[int(k)] = v.detach().float().cpu().reshape(-1)
    table = table.to(device)
    _table_cache.clear()
    _table_cache[key] = (prototype_dict, table)
    return table


class _StructureLoss(torch.autograd.Function):
    """sum_g ||group g||_2 over the flat LoRA bucket; backward adds coef * t/||g|| straight into the
    flat gradient bucket (the parameters' .grad views)."""

    @staticmethod
    def forward(ctx, bucket, tgroup, ngroups, grad_scale, *params):
        out = ops.group_norms_fwd(bucket.flat, bucket.toff, bucket.tnumel, tgroup, ngroups)
        ctx.bucket, ctx.tgroup, ctx.norm, ctx.grad_scale, ctx.n = bucket, tgroup, out["group_norm"], grad_scale, len(params)
        return out["loss"][0]

    @staticmethod
    def backward(ctx, g):
        b = ctx.bucket
        b.attach_grads()
        ops.group_norms_bwd(b.flat, b.toff, b.tnumel, ctx.tgroup, ctx.norm, g.reshape(1).float().contiguous(),
                            ctx.grad_scale, b.grad)
        return (None, None, None, None) + (None,) * ctx.n


def structure_loss(model, group_type="block", grad_scale=1.0):
    """Differentiable group-lasso term. grad_scale lets a data-parallel engine pre-divide the
    (rank-identical) parameter-only gradient by the world size before the gradient all-reduce."""
    bucket = model.lora_bucket()
    if bucket is None:
        raise RuntimeError("structure loss needs a model with LoRA parameters (lora_rank > 0)")
    tgroup, ng = bucket.group_table(group_type)
    trainable = [p for p in bucket.params if p.requires_grad]
    if torch.is_grad_enabled() and trainable:
        return _StructureLoss.apply(bucket, tgroup, ng, grad_scale, *trainable)
    return ops.group_norms_fwd(bucket.flat, bucket.toff, bucket.tnumel, tgroup, ng)["loss"][0]


def group_report(model, group_type="block", tau=0.0):
    """All K12 observables in one launch: group-lasso norms, cal_norm (sum of Frobenius norms),
    the selection mask (norm > tau) and the loss."""
    bucket = model.lora_bucket()
    tgroup, ng = bucket.group_table(group_type)
    return ops.group_norms_fwd(bucket.flat, bucket.toff, bucket.tnumel, tgroup, ng, tau=tau)
