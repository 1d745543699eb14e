"""Synthetic inputs and deterministic parameter initialisation (CPU generators, seed-stable).

The batch layout follows SURVEY.md 8(d): ``image ~ randn[B,3,H,W]`` (post-normalisation image
statistics) and ``text`` laid out as the reference tokenizer does
(``src/open_clip/tokenizer.py:277-291``: SOT 49406 first, EOT = the largest id once, zero padding)
so that the 'argmax' pooling of ``text_global_pool`` (transformer.py:941-944) is unambiguous.
"""
import math

import torch


def synthetic_batch(cfg: dict, batch_size: int, seed: int = 1234, rank: int = 0, device="cpu", image_dtype=torch.float32):
    v, t = cfg["vision_cfg"], cfg["text_cfg"]
    g = torch.Generator().manual_seed(seed + rank)
    H = v["image_size"]
    image = torch.randn(batch_size, 3, H, H, generator=g)
    ctx, vocab = t["context_length"], t["vocab_size"]
    sot, eot = vocab - 2, vocab - 1
    text = torch.zeros(batch_size, ctx, dtype=torch.int64)
    lengths = torch.randint(min(8, ctx - 1), ctx, (batch_size,), generator=g)  # EOT position in [8, ctx-1]
    body = torch.randint(1, vocab - 2, (batch_size, ctx), generator=g)
    pos = torch.arange(ctx).unsqueeze(0)
    text = torch.where(pos < lengths.unsqueeze(1), body, text)
    text[:, 0] = sot
    text[torch.arange(batch_size), lengths] = eot
    return {"image": image.to(device=device, dtype=image_dtype), "text": text.to(device)}


def init_state_dict(cfg: dict, seed: int = 0, perturb: bool = False, siglip: bool = False) -> dict:
    """Random parameters with the reference's names/shapes (SURVEY.md 8a 'State-dict layout') and
    the reference's init *distributions* (transformer.py:632-645,714 vision; :1664-1685 text;
    model.py:326,363 logit_scale).  Not bit-equal to the reference's init (RNG order differs) --
    parity tests copy one state dict into both sides instead (SURVEY.md 8c 'Init parity').

    ``perturb=True`` additionally randomises LayerNorm affine params and all biases so that parity
    tests exercise them (the reference initialises them to 1/0).
    """
    g = torch.Generator().manual_seed(seed)
    v, t, e = cfg["vision_cfg"], cfg["text_cfg"], cfg["embed_dim"]
    sd = {}

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    def uniform(*shape, bound):
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound

    def ln(prefix, width):
        sd[prefix + ".weight"] = 1 + (randn(width, std=0.1) if perturb else torch.zeros(width))
        sd[prefix + ".bias"] = randn(width, std=0.1) if perturb else torch.zeros(width)

    def block(prefix, width, mlp, attn_std=None, proj_std=None, fc_std=None):
        ln(prefix + "ln_1", width)
        if attn_std is None:  # vision: xavier_uniform on in_proj, default Linear init elsewhere
            sd[prefix + "attn.in_proj_weight"] = uniform(3 * width, width, bound=math.sqrt(6.0 / (4 * width)))
            sd[prefix + "attn.out_proj.weight"] = uniform(width, width, bound=1 / math.sqrt(width))
            sd[prefix + "mlp.c_fc.weight"] = uniform(mlp, width, bound=1 / math.sqrt(width))
            sd[prefix + "mlp.c_proj.weight"] = uniform(width, mlp, bound=1 / math.sqrt(mlp))
        else:
            sd[prefix + "attn.in_proj_weight"] = randn(3 * width, width, std=attn_std)
            sd[prefix + "attn.out_proj.weight"] = randn(width, width, std=proj_std)
            sd[prefix + "mlp.c_fc.weight"] = randn(mlp, width, std=fc_std)
            sd[prefix + "mlp.c_proj.weight"] = randn(width, mlp, std=proj_std)
        bstd = 0.02 if perturb else 0.0
        sd[prefix + "attn.in_proj_bias"] = randn(3 * width, std=bstd)
        sd[prefix + "attn.out_proj.bias"] = randn(width, std=bstd)
        ln(prefix + "ln_2", width)
        sd[prefix + "mlp.c_fc.bias"] = randn(mlp, std=bstd) if perturb else uniform(mlp, bound=1 / math.sqrt(width))
        sd[prefix + "mlp.c_proj.bias"] = randn(width, std=bstd) if perturb else uniform(width, bound=1 / math.sqrt(mlp))

    # --- text side (CLIP unpacks the text tower onto itself: model.py:351-360) ---
    tw = t["width"]
    sd["positional_embedding"] = randn(t["context_length"], tw, std=0.01)
    sd["text_projection"] = randn(tw, e, std=tw ** -0.5)
    sd["logit_scale"] = torch.tensor(math.log(10.0) if siglip else math.log(1 / 0.07))
    if siglip:
        sd["logit_bias"] = torch.tensor(-10.0)  # main.py:259-261
    # --- vision ---
    vw, ps = v["width"], v["patch_size"]
    scale = vw ** -0.5
    ntok = (v["image_size"] // ps) ** 2 + 1
    sd["visual.class_embedding"] = randn(vw, std=scale)
    sd["visual.positional_embedding"] = randn(ntok, vw, std=scale)
    sd["visual.proj"] = randn(vw, e, std=scale)
    sd["visual.conv1.weight"] = uniform(vw, 3, ps, ps, bound=1 / math.sqrt(3 * ps * ps))
    ln("visual.ln_pre", vw)
    for i in range(v["layers"]):
        block(f"visual.transformer.resblocks.{i}.", vw, int(vw * v.get("mlp_ratio", 4.0)))
    ln("visual.ln_post", vw)
    # --- text blocks ---
    proj_std = (tw ** -0.5) * ((2 * t["layers"]) ** -0.5)
    for i in range(t["layers"]):
        block(f"transformer.resblocks.{i}.", tw, int(tw * t.get("mlp_ratio", 4.0)),
              attn_std=tw ** -0.5, proj_std=proj_std, fc_std=(2 * tw) ** -0.5)
    sd["token_embedding.weight"] = randn(t["vocab_size"], tw, std=0.02)
    ln("ln_final", tw)
    return sd
