"""CPU restatement of the reference CLIP training hot path -- TEST INFRASTRUCTURE ONLY.

A functional, fp32, plain-``torch`` (CPU) restatement of what ``/root/reference`` computes on
the path named in SURVEY.md section 8(a): ViT image tower + causal text transformer +
``ClipLoss`` / ``SigLipLoss`` (forward; backward comes from ``torch.autograd`` applied to this
restatement).  It takes a *state dict with the reference's key names* and a config dict, so the
same weights drive the reference, this oracle and the HIP path.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which ``oracle/make_golden.py`` produced by running the reference's own
``open_clip.model.CLIP`` / ``open_clip.loss.*`` (imported from ``/root/reference/src``) on the same
seeded inputs.  The reference stores no golden vectors of its own for this path.

All math is spelled out with elementary tensor ops (no ``F.layer_norm`` / ``F.gelu`` / SDPA /
``F.cross_entropy``) so the oracle is an independent statement of the algorithm.  Citations are
``file:line`` into ``/root/reference/src/open_clip``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """layers.py:20-26 (``LayerNorm``: F.layer_norm over the last dim, eps=1e-5, biased var)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """transformer.py:295-299: ``act_layer=nn.GELU`` -> exact erf GELU."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def quick_gelu(x: Tensor) -> Tensor:
    """layers.py:29-32 (``QuickGELU``): x * sigmoid(1.702 x); selected by ``quick_gelu`` in the model config (model.py:172-176, :262)."""
    return x * (1.0 / (1.0 + torch.exp(-1.702 * x)))


def attention(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int, causal: bool) -> Tensor:
    """transformer.py:157-248 (``Attention.forward``, self-attention fast path).

    qkv = x @ in_proj_weight.T + in_proj_bias, rows ordered [Wq;Wk;Wv] (:169); heads split as
    [N, H, L, hd] (:199-201); softmax(q k^T * hd^-0.5 + mask) v (:223-228); out_proj (:246).
    ``causal`` stands for the additive -inf upper-triangular mask of transformer.py:1716-1722.
    """
    N, L, C = x.shape
    hd = C // heads
    qkv = x @ p[pre + "attn.in_proj_weight"].t() + p[pre + "attn.in_proj_bias"]
    q, k, v = qkv.split(C, dim=-1)
    q = q.reshape(N, L, heads, hd).permute(0, 2, 1, 3)
    k = k.reshape(N, L, heads, hd).permute(0, 2, 1, 3)
    v = v.reshape(N, L, heads, hd).permute(0, 2, 1, 3)
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    if causal:
        mask = torch.full((L, L), float("-inf"), dtype=s.dtype).triu_(1)
        s = s + mask
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    a = e / e.sum(dim=-1, keepdim=True)
    o = (a @ v).permute(0, 2, 1, 3).reshape(N, L, C)
    return o @ p[pre + "attn.out_proj.weight"].t() + p[pre + "attn.out_proj.bias"]


def resblock(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int, causal: bool, quick: bool = False) -> Tensor:
    """transformer.py:319-330 (``ResidualAttentionBlock.forward``; ls_1/ls_2 = Identity)."""
    x = x + attention(layer_norm(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"]), p, pre, heads, causal)
    h = layer_norm(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
    h = h @ p[pre + "mlp.c_fc.weight"].t() + p[pre + "mlp.c_fc.bias"]
    h = quick_gelu(h) if quick else gelu_erf(h)
    h = h @ p[pre + "mlp.c_proj.weight"].t() + p[pre + "mlp.c_proj.bias"]
    return x + h


def transformer(x: Tensor, p: Dict[str, Tensor], pre: str, layers: int, heads: int, causal: bool, quick: bool = False) -> Tensor:
    """transformer.py:577-585 (``Transformer.forward``: for r in resblocks)."""
    for i in range(layers):
        x = resblock(x, p, f"{pre}resblocks.{i}.", heads, causal, quick)
    return x


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """model.py:391,411: ``F.normalize(x, dim=-1)`` = x / max(||x||_2, 1e-12)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


# ----------------------------------------------------------------------------------------------
# towers
# ----------------------------------------------------------------------------------------------
def encode_image(image: Tensor, p: Dict[str, Tensor], cfg: dict, normalize: bool = True) -> Tensor:
    """model.py:389-391 -> transformer.py:917-928 (``VisionTransformer.forward``).

    _embeds (:793-808): conv1 (kernel=stride=patch, no bias) == per-patch dot product with the
    flattened [3*ps*ps] filter; prepend class_embedding; + positional_embedding; ln_pre.
    _pool (:829-831) default branch: ln_post on all tokens then take token 0 ('tok').
    then ``pooled @ proj`` (:923).
    """
    v = cfg["vision_cfg"]
    ps, width = v["patch_size"], v["width"]
    B, Cin, H, W = image.shape
    gh, gw = H // ps, W // ps
    w = p["visual.conv1.weight"].reshape(width, Cin * ps * ps)
    patches = image.reshape(B, Cin, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cin * ps * ps)
    x = patches @ w.t()
    cls = p["visual.class_embedding"].reshape(1, 1, width).expand(B, 1, width)
    x = torch.cat([cls, x], dim=1) + p["visual.positional_embedding"]
    x = layer_norm(x, p["visual.ln_pre.weight"], p["visual.ln_pre.bias"])
    heads = width // v.get("head_width", 64)
    x = transformer(x, p, "visual.transformer.", v["layers"], heads, causal=False, quick=bool(cfg.get("quick_gelu", False)))
    x = layer_norm(x, p["visual.ln_post.weight"], p["visual.ln_post.bias"])
    pooled = x[:, 0] @ p["visual.proj"]
    return l2_normalize(pooled) if normalize else pooled


def encode_text(text: Tensor, p: Dict[str, Tensor], cfg: dict, normalize: bool = True) -> Tensor:
    """model.py:396-411 (``CLIP._encode_text``) + transformer.py:931-954 (``text_global_pool`` 'argmax')."""
    t = cfg["text_cfg"]
    x = p["token_embedding.weight"][text] + p["positional_embedding"]
    x = transformer(x, p, "transformer.", t["layers"], t["heads"], causal=True, quick=bool(cfg.get("quick_gelu", False)))
    x = layer_norm(x, p["ln_final.weight"], p["ln_final.bias"])
    pooled = x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ p["text_projection"]
    return l2_normalize(pooled) if normalize else pooled


def clip_forward(image: Tensor, text: Tensor, p: Dict[str, Tensor], cfg: dict) -> Dict[str, Tensor]:
    """model.py:528-548 (``CLIP.forward``): normalized features + ``logit_scale.exp()`` (+ bias clone)."""
    out = {
        "image_features": encode_image(image, p, cfg, True),
        "text_features": encode_text(text, p, cfg, True),
        "logit_scale": p["logit_scale"].exp(),
    }
    if "logit_bias" in p:
        out["logit_bias"] = p["logit_bias"] + 0.0
    return out


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def cross_entropy_arange(logits: Tensor, offset: int = 0) -> Tensor:
    """F.cross_entropy(logits, arange(n)+offset), mean reduction (loss.py:136-139, :78-89)."""
    n = logits.shape[0]
    m = logits.max(dim=-1, keepdim=True).values
    lse = (logits - m).exp().sum(dim=-1).log() + m.squeeze(-1)
    idx = torch.arange(n) + offset
    return (lse - logits[torch.arange(n), idx]).mean()


def clip_loss(
    image_features: Tensor,
    text_features: Tensor,
    logit_scale: Tensor,
    all_image_features: Optional[Tensor] = None,
    all_text_features: Optional[Tensor] = None,
    local_loss: bool = False,
    rank: int = 0,
) -> Tensor:
    """loss.py:91-141 (``ClipLoss.get_logits`` + ``forward``).

    world_size == 1 (all_* None):  li = s*I@T^T, lt = s*T@I^T                      (:109-110)
    global loss:                   li = s*I_all@T_all^T, lt = li^T, labels arange(N) (:106-107)
    local loss:                    li = s*I@T_all^T, lt = s*T@I_all^T, labels + B*rank (:103-104, :82-83)
    ``all_*`` are what ``gather_features`` (loss.py:29-54) returned on this rank.
    """
    if all_image_features is None:
        li = (logit_scale * image_features) @ text_features.t()
        lt = (logit_scale * text_features) @ image_features.t()
        off = 0
    elif local_loss:
        li = (logit_scale * image_features) @ all_text_features.t()
        lt = (logit_scale * text_features) @ all_image_features.t()
        off = image_features.shape[0] * rank
    else:
        li = (logit_scale * all_image_features) @ all_text_features.t()
        lt = li.t()
        off = 0
    return (cross_entropy_arange(li, off) + cross_entropy_arange(lt, off)) / 2


def siglip_pair_loss(image_features: Tensor, text_features: Tensor, logit_scale: Tensor, logit_bias: Tensor,
                     negative_only: bool = False) -> Tensor:
    """loss.py:356-367 (``SigLipLoss._loss``): -sum(logsigmoid(labels * (s*I@T^T + b))) / B,
    labels = 2*eye - 1 (or all -1 when ``negative_only``) (:344-348)."""
    logits = (logit_scale * image_features) @ text_features.t() + logit_bias
    n = logits.shape[0]
    labels = -torch.ones_like(logits)
    if not negative_only:
        labels = labels + 2 * torch.eye(n, dtype=logits.dtype)
    z = labels * logits
    logsig = torch.where(z >= 0, -torch.log1p(torch.exp(-z)), z - torch.log1p(torch.exp(z)))
    return -logsig.sum() / n


def siglip_loss(image_features: Tensor, text_features_by_rank, logit_scale: Tensor, logit_bias: Tensor, rank: int = 0) -> Tensor:
    """loss.py:406-489 (``SigLipLoss.forward``): the local positive+negative term plus one
    ``negative_only`` term per *other* rank's text chunk (order-independent sum; not averaged)."""
    loss = siglip_pair_loss(image_features, text_features_by_rank[rank], logit_scale, logit_bias)
    for r, tf in enumerate(text_features_by_rank):
        if r != rank:
            loss = loss + siglip_pair_loss(image_features, tf, logit_scale, logit_bias, negative_only=True)
    return loss


# ----------------------------------------------------------------------------------------------
# one training step: loss + grads (autograd over the restatement above)
# ----------------------------------------------------------------------------------------------
def train_forward_backward(image: Tensor, text: Tensor, state: Dict[str, Tensor], cfg: dict, siglip: bool = False
                           ) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """clip_task.py:41-46 (``CLIPTask.training_forward``) followed by ``loss.backward()``
    (train.py:179), world_size == 1.  Returns (outputs, grads-by-state-dict-key)."""
    p = {k: v.detach().clone().float().requires_grad_(True) for k, v in state.items()}
    out = clip_forward(image.float(), text, p, cfg)
    if siglip:
        loss = siglip_loss(out["image_features"], [out["text_features"]], out["logit_scale"], out["logit_bias"], 0)
    else:
        loss = clip_loss(out["image_features"], out["text_features"], out["logit_scale"])
    loss.backward()
    grads = {k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    outs = {k: v.detach() for k, v in out.items()}
    outs["loss"] = loss.detach()
    outs["logits_per_image"] = (out["logit_scale"] * out["image_features"] @ out["text_features"].t()).detach()
    return outs, grads
