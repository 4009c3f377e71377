"""Deterministic weight / input recipe shared by the oracle, the golden-vector
generator and the parity tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing in the product
package imports this.

Both sides of every parity test (reference-in-container, oracle, HIP path)
rebuild identical tensors from (name, seed) with a counter-based splitmix64
hash, so no weights have to be committed (SURVEY.md §8c "Fixture policy").
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(name, shape, seed=1337, lo=-1.0, hi=1.0):
    """float32 U[lo,hi) array, a pure function of (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + (key << np.uint64(32))
        bits = _splitmix64(_splitmix64(ctr))
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24-bit mantissa
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(name, shape, seed=1337, std=1.0):
    """Irwin-Hall(4) pseudo-normal, float32."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += uniform(f"{name}#{k}", shape, seed, 0.0, 1.0)
    return ((acc - 2.0) * (std / np.sqrt(4.0 / 12.0))).astype(np.float32)


# ---------------------------------------------------------------------------
# model configurations
# ---------------------------------------------------------------------------
def cfg_full(lora_rank=8, num_class=100):
    """ViT-P8S8 depth 6 (reference train/train_own_forget_cl.py:207-221)."""
    return dict(image_size=112, patch_size=8, dim=512, depth=6, heads=8, dim_head=64,
                mlp_dim=2048, num_class=num_class, lora_rank=lora_rank, channels=3)


def cfg_small(lora_rank=4, num_class=10):
    """Shrunken model for fast unit tests: 40px / patch 8 -> 25 patches (+cls = 26 tokens)."""
    return dict(image_size=40, patch_size=8, dim=64, depth=2, heads=1, dim_head=64,
                mlp_dim=128, num_class=num_class, lora_rank=lora_rank, channels=3)


def cfg_small_attn(lora_rank=8, num_class=12):
    """cfg_small2 with the adapters on the QKV projection instead of the FFN (--lora_pos Attention)."""
    c = cfg_small2(lora_rank, num_class)
    c["lora_pos"] = "Attention"
    return c


def cfg_small2(lora_rank=8, num_class=12):
    """Second small model: 2 heads, 3 layers, ragged token count (48px/8 -> 37 tokens)."""
    return dict(image_size=48, patch_size=8, dim=128, depth=3, heads=2, dim_head=64,
                mlp_dim=256, num_class=num_class, lora_rank=lora_rank, channels=3)


def cfg_small6(lora_rank=8, num_class=12):
    """cfg_small2 with SIX layers: the reference's engine_cl.get_structure_loss hard-codes six per-block groups (engine_cl.py:390-396),
    so every scenario that runs the real continual engine needs depth 6."""
    c = cfg_small2(lora_rank, num_class)
    c["depth"] = 6
    return c


def param_shapes(cfg):
    """Ordered {name: shape} exactly as the reference module tree names them
    (probe of vit_pytorch_face/vit_face.py:449-521)."""
    d, h, dh, mlp, r = cfg["dim"], cfg["heads"], cfg["dim_head"], cfg["mlp_dim"], cfg["lora_rank"]
    inner = h * dh
    attn_lora = cfg.get("lora_pos", "FFN") == "Attention"
    npatch = (cfg["image_size"] // cfg["patch_size"]) ** 2
    pdim = cfg["channels"] * cfg["patch_size"] ** 2
    sh = {}
    sh["pos_embedding"] = (1, npatch + 1, d)
    sh["cls_token"] = (1, 1, d)
    sh["patch_to_embedding.weight"] = (d, pdim)
    sh["patch_to_embedding.bias"] = (d,)
    for i in range(cfg["depth"]):
        a = f"transformer.layers.{i}.0.fn"
        f = f"transformer.layers.{i}.1.fn"
        sh[f"{a}.norm.weight"] = (d,)
        sh[f"{a}.norm.bias"] = (d,)
        sh[f"{a}.fn.to_qkv.weight"] = (3 * inner, d)
        if attn_lora:      # --lora_pos Attention: loralib.MergedLinear(enable_lora=[True]*3) on to_qkv (vit_face.py:349-355)
            sh[f"{a}.fn.to_qkv.lora_A"] = (3 * r, d)
            sh[f"{a}.fn.to_qkv.lora_B"] = (3 * inner, r)
        sh[f"{a}.fn.to_out.0.weight"] = (d, inner)
        sh[f"{a}.fn.to_out.0.bias"] = (d,)
        sh[f"{f}.norm.weight"] = (d,)
        sh[f"{f}.norm.bias"] = (d,)
        sh[f"{f}.fn.net.0.weight"] = (mlp, d)
        sh[f"{f}.fn.net.0.bias"] = (mlp,)
        if not attn_lora:
            sh[f"{f}.fn.net.0.lora_A"] = (r, d)
            sh[f"{f}.fn.net.0.lora_B"] = (mlp, r)
        sh[f"{f}.fn.net.3.weight"] = (d, mlp)
        sh[f"{f}.fn.net.3.bias"] = (d,)
        if not attn_lora:
            sh[f"{f}.fn.net.3.lora_A"] = (r, mlp)
            sh[f"{f}.fn.net.3.lora_B"] = (d, r)
    sh["mlp_head.0.weight"] = (d,)
    sh["mlp_head.0.bias"] = (d,)
    sh["loss.weight"] = (cfg["num_class"], d)
    return sh


def make_state(cfg, seed=1337, lora_b_std=0.02):
    """Deterministic fp32 state dict (numpy). Scales mimic a trained net closely enough
    that logits are O(10) and every code path (LoRA-B != 0) is non-trivial."""
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("norm.weight") or name == "mlp_head.0.weight":
            v = 1.0 + uniform(name, shape, seed, -0.2, 0.2)
        elif name.endswith("norm.bias") or name == "mlp_head.0.bias":
            v = uniform(name, shape, seed, -0.1, 0.1)
        elif name in ("pos_embedding", "cls_token"):
            v = normalish(name, shape, seed, 0.5)
        elif name.endswith("lora_A"):
            bound = float(np.sqrt(6.0 / ((1 + 5.0) * shape[1])))  # kaiming_uniform(a=sqrt5)
            v = uniform(name, shape, seed, -bound, bound)
        elif name.endswith("lora_B"):
            v = normalish(name, shape, seed, lora_b_std)
        elif name.endswith(".bias"):
            v = uniform(name, shape, seed, -0.05, 0.05)
        elif name == "loss.weight":
            bound = float(np.sqrt(6.0 / (shape[0] + shape[1])))  # xavier_uniform
            v = uniform(name, shape, seed, -bound, bound)
        else:  # dense weights: U(+-1/sqrt(fan_in)) like nn.Linear default
            bound = 1.0 / float(np.sqrt(shape[1]))
            v = uniform(name, shape, seed, -bound, bound)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def make_images(cfg, batch, seed=7, tag="img"):
    """u8 pattern / 255 == torchvision ToTensor() range (train_own_forget_cl.py:131-135)."""
    shape = (batch, cfg["channels"], cfg["image_size"], cfg["image_size"])
    u = uniform(f"{tag}", shape, seed, 0.0, 256.0)
    return (np.floor(u).clip(0, 255) / 255.0).astype(np.float32)


def make_labels(cfg, batch, seed=7, tag="lab", lo=0, hi=None):
    hi = cfg["num_class"] if hi is None else hi
    u = uniform(f"{tag}", (batch,), seed, float(lo), float(hi))
    return np.floor(u).clip(lo, hi - 1).astype(np.int64)


def make_prototypes(cfg, seed=11):
    """[num_class, dim] table standing in for calculate_prototypes output."""
    return normalish("prototypes", (cfg["num_class"], cfg["dim"]), seed, 1.0)


# ---------------------------------------------------------------------------
# ViT-B/16 family (torchvision VisionTransformer naming; reference vit_pytorch_face/modified_VIT.py)
# ---------------------------------------------------------------------------
def cfg_vitb(lora_rank=16, num_class=1000):
    """torchvision vit_b_16: 224 px, patch 16, 12 layers x 12 heads x 64, mlp 3072 (train_own_forget_cl.py:238-241)."""
    return dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072, num_class=num_class,
                lora_rank=lora_rank, channels=3)


def cfg_vitb_small(lora_rank=4, num_class=20):
    """12 layers (the reference hard-codes 12 groups, engine_cl.py:395-403) but 64 px / dim 64 / 1 head: 17 tokens."""
    return dict(image_size=64, patch_size=16, dim=64, depth=12, heads=1, mlp_dim=128, num_class=num_class,
                lora_rank=lora_rank, channels=3)


def cfg_vitb_small2(lora_rank=16, num_class=16):
    """2 heads, rank 16 (the ImageNet100 rank), 3 layers, 96 px -> 37 tokens."""
    return dict(image_size=96, patch_size=16, dim=128, depth=3, heads=2, mlp_dim=256, num_class=num_class,
                lora_rank=lora_rank, channels=3)


def tv_param_shapes(cfg):
    """Ordered {name: shape} of ModifiedViT(vit) after replace_ffn_with_lora (named_parameters() order)."""
    d, mlp, r, p = cfg["dim"], cfg["mlp_dim"], cfg["lora_rank"], cfg["patch_size"]
    ntok = (cfg["image_size"] // p) ** 2 + 1
    sh = {"class_token": (1, 1, d), "conv_proj.weight": (d, cfg["channels"], p, p), "conv_proj.bias": (d,),
          "encoder.pos_embedding": (1, ntok, d)}
    for i in range(cfg["depth"]):
        e = f"encoder.layers.encoder_layer_{i}"
        sh[f"{e}.ln_1.weight"] = (d,)
        sh[f"{e}.ln_1.bias"] = (d,)
        sh[f"{e}.self_attention.in_proj_weight"] = (3 * d, d)
        sh[f"{e}.self_attention.in_proj_bias"] = (3 * d,)
        sh[f"{e}.self_attention.out_proj.weight"] = (d, d)
        sh[f"{e}.self_attention.out_proj.bias"] = (d,)
        sh[f"{e}.ln_2.weight"] = (d,)
        sh[f"{e}.ln_2.bias"] = (d,)
        for j, (o, k) in ((0, (mlp, d)), (3, (d, mlp))):
            sh[f"{e}.mlp.{j}.weight"] = (o, k)
            sh[f"{e}.mlp.{j}.bias"] = (o,)
            if r > 0:
                sh[f"{e}.mlp.{j}.lora_A"] = (r, k)
                sh[f"{e}.mlp.{j}.lora_B"] = (o, r)
    sh["encoder.ln.weight"] = (d,)
    sh["encoder.ln.bias"] = (d,)
    sh["heads.head.weight"] = (cfg["num_class"], d)
    sh["heads.head.bias"] = (cfg["num_class"],)
    return sh


def make_tv_state(cfg, seed=4242, lora_b_std=0.02):
    out = {}
    for name, shape in tv_param_shapes(cfg).items():
        if name.endswith(("ln_1.weight", "ln_2.weight", "ln.weight")):
            v = 1.0 + uniform(name, shape, seed, -0.2, 0.2)
        elif name.endswith(("ln_1.bias", "ln_2.bias", "ln.bias")):
            v = uniform(name, shape, seed, -0.1, 0.1)
        elif name in ("encoder.pos_embedding", "class_token"):
            v = normalish(name, shape, seed, 0.5)
        elif name.endswith("lora_A"):
            bound = float(np.sqrt(6.0 / ((1 + 5.0) * shape[1])))
            v = uniform(name, shape, seed, -bound, bound)
        elif name.endswith("lora_B"):
            v = normalish(name, shape, seed, lora_b_std)
        elif name.endswith("bias"):
            v = uniform(name, shape, seed, -0.05, 0.05)
        elif name == "heads.head.weight":
            v = uniform(name, shape, seed, -0.3, 0.3)      # logits O(1..5): CE and top-1 are non-trivial
        else:
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / float(np.sqrt(fan_in))
            v = uniform(name, shape, seed, -bound, bound)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out
