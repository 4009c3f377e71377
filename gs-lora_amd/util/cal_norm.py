"""get_norm_of_lora — per-group LoRA norm report (reference util/cal_norm.py:4-146), one HIP launch.
Note the reference's definition: SUM of the member tensors' norms, not the group-lasso norm."""
import torch


def get_norm_of_lora(model, type="L2", group_num=6, group_type: str = "block", group_pos: str = "FFN",
                     imagenet: bool = False):
    if group_pos not in ("FFN", "Attention"):
        raise ValueError("group_pos must be 'FFN' or 'Attention'")
    if group_pos != getattr(model, "lora_pos", "FFN"):
        raise ValueError(f"group_pos={group_pos!r} but the model was built with lora_pos={getattr(model, 'lora_pos', 'FFN')!r}")
    if group_pos == "Attention":      # reference :108-120: one (to_qkv.lora_A, to_qkv.lora_B) group per block
        group_type = "block"
    if imagenet:      # reference :91-107: always the 12 per-block groups of ViT-B/16, group_num / group_type ignored
        group_type, group_num = "block", len(model.hip_spec().blocks)
    if type not in ("L2", "L1"):
        raise ValueError("type should be L1 or L2")
    bucket = model.lora_bucket()
    with torch.no_grad():
        if type == "L1":   # never used by the drivers; tiny, so plain reductions over the bucket views
            tg, ng = bucket.group_table(group_type)
            vals = [torch.zeros((), device=bucket.flat.device) for _ in range(ng)]
            for p, g in zip(bucket.params, tg.tolist()):
                vals[g] = vals[g] + p.detach().abs().sum()
            return vals
        from gslora_hip.losses import group_report
        cn = group_report(model, group_type)["cal_norm"]
        ngroups = cn.numel()
        if group_type == "block" and group_num != ngroups:
            cn = cn[:group_num]
        return [cn[i] for i in range(cn.numel())]
