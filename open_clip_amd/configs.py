"""Model-config registry for the native CLIP path.

Mirrors the reference's JSON registry (``src/open_clip/model_configs/*.json``, looked up by
``factory.get_model_config`` factory.py:154-169): same keys (``embed_dim``, ``vision_cfg``,
``text_cfg``) and the same defaults the reference dataclasses apply
(``CLIPVisionCfg.head_width = 64`` model.py:37-60, ``CLIPTextCfg`` model.py:108-150).
Registered: every ``ViT-*`` config of the reference whose towers are the plain pre-LN ViT + causal text transformer this path implements
(learnable position embedding, class-token pooling, argmax text pooling, nn.GELU or QuickGELU; head_dim a multiple of 8 up to 128) -- the BASELINE
models (ViT-B-32, ViT-L-14, ViT-H-14) and their size / resolution variants -- plus tiny ones for parity tests.  SigLIP / CoCa / timm / HF
configs need other blocks (attentional pooling, no class token, other tokenisers) and are rejected by ``NativeCLIP._check_cfg`` if added.
"""
import copy
import json
import os


def _vit(embed, image, layers, width, patch, t_width, t_heads, t_layers, head_width=None, mlp_ratio=None, quick_gelu=False):
    v = {"image_size": image, "layers": layers, "width": width, "patch_size": patch}
    if head_width is not None:
        v["head_width"] = head_width
    if mlp_ratio is not None:
        v["mlp_ratio"] = mlp_ratio
    cfg = {"embed_dim": embed, "vision_cfg": v,
           "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": t_width, "heads": t_heads, "layers": t_layers}}
    if quick_gelu:
        cfg["quick_gelu"] = True
    return cfg


# name: (embed_dim, image_size, vision layers, vision width, patch, text width, text heads, text layers[, head_width[, mlp_ratio]])
# -- the values of src/open_clip/model_configs/<name>.json
_REFERENCE_VITS = {
    "ViT-S-32": (384, 224, 12, 384, 32, 384, 6, 12), "ViT-S-32-alt": (256, 224, 12, 384, 32, 256, 4, 10),
    "ViT-S-16": (384, 224, 12, 384, 16, 384, 6, 12), "ViT-S-16-alt": (256, 224, 12, 384, 16, 256, 4, 10),
    "ViT-M-32": (512, 224, 12, 512, 32, 512, 8, 12), "ViT-M-32-alt": (384, 224, 12, 512, 32, 384, 6, 12),
    "ViT-M-16": (512, 224, 12, 512, 16, 512, 8, 12),  # (ViT-M-16-alt has LayerScale: not this path)
    "ViT-B-32-256": (512, 256, 12, 768, 32, 512, 8, 12), "ViT-B-32-plus-256": (640, 256, 12, 896, 32, 640, 10, 12),
    "ViT-B-16": (512, 224, 12, 768, 16, 512, 8, 12), "ViT-B-16-plus": (640, 224, 12, 896, 16, 640, 10, 12),
    "ViT-B-16-plus-240": (640, 240, 12, 896, 16, 640, 10, 12),
    "ViT-L-14-280": (768, 280, 24, 1024, 14, 768, 12, 12), "ViT-L-14-336": (768, 336, 24, 1024, 14, 768, 12, 12),
    "ViT-L-16": (768, 224, 24, 1024, 16, 768, 12, 12), "ViT-L-16-320": (768, 320, 24, 1024, 16, 768, 12, 12),
    "ViT-H-14-378": (1024, 378, 32, 1280, 14, 1024, 16, 24, 80), "ViT-H-16": (1024, 224, 32, 1280, 16, 1024, 16, 24, 80),
    # head_dim 88 / 104 / 112 and the MLP widths 6144 / 8192 / 15360 (int(width * mlp_ratio), transformer.py:283)
    "ViT-g-14": (1024, 224, 40, 1408, 14, 1024, 16, 24, 88, 4.3637), "ViT-bigG-14": (1280, 224, 48, 1664, 14, 1280, 20, 32, 104, 4.9231),
    "ViT-e-14": (1280, 224, 56, 1792, 14, 1280, 20, 36, 112, 8.5715),
}
_REFERENCE_QUICKGELU = ("ViT-B-16", "ViT-L-14", "ViT-L-14-336", "ViT-H-14", "ViT-H-14-378", "ViT-bigG-14")  # <name>-quickgelu.json twins

_MODEL_CONFIGS = {
    # src/open_clip/model_configs/ViT-B-32.json
    "ViT-B-32": {
        "embed_dim": 512,
        "vision_cfg": {"image_size": 224, "layers": 12, "width": 768, "patch_size": 32},
        "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 512, "heads": 8, "layers": 12},
    },
    # src/open_clip/model_configs/ViT-B-32-quickgelu.json (the OpenAI / LAION-400M ViT-B/32 checkpoints: QuickGELU in both towers)
    "ViT-B-32-quickgelu": {
        "embed_dim": 512,
        "quick_gelu": True,
        "vision_cfg": {"image_size": 224, "layers": 12, "width": 768, "patch_size": 32},
        "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 512, "heads": 8, "layers": 12},
    },
    # src/open_clip/model_configs/ViT-L-14.json
    "ViT-L-14": {
        "embed_dim": 768,
        "vision_cfg": {"image_size": 224, "layers": 24, "width": 1024, "patch_size": 14},
        "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 768, "heads": 12, "layers": 12},
    },
    # src/open_clip/model_configs/ViT-H-14.json
    "ViT-H-14": {
        "embed_dim": 1024,
        "vision_cfg": {"image_size": 224, "layers": 32, "width": 1280, "head_width": 80, "patch_size": 14},
        "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 1024, "heads": 16, "layers": 24},
    },
    # parity-test twins (not in the reference registry; same code path, small shapes)
    "tiny-test": {
        "embed_dim": 64,
        "vision_cfg": {"image_size": 64, "layers": 2, "width": 128, "patch_size": 32},
        "text_cfg": {"context_length": 16, "vocab_size": 512, "width": 128, "heads": 2, "layers": 2},
    },
    # ViT-H-14's shape class in miniature: image head_dim 80 (head_width 80), patch 14, odd token count (26), text head_dim 64
    "hd80-test": {
        "embed_dim": 96,
        "vision_cfg": {"image_size": 70, "layers": 2, "width": 160, "head_width": 80, "patch_size": 14},
        "text_cfg": {"context_length": 77, "vocab_size": 1024, "width": 128, "heads": 2, "layers": 2},
    },
    # ViT-g-14's shape class in miniature: image head_dim 88 (contraction zero-padded to 96 in the attention kernels), MLP width
    # int(352 * 4.3637) = 1536, patch 14, 26 tokens (widths are multiples of 32: the GEMM kernels' K granularity)
    "hd88-test": {
        "embed_dim": 96,
        "vision_cfg": {"image_size": 70, "layers": 1, "width": 352, "head_width": 88, "mlp_ratio": 4.3637, "patch_size": 14},
        "text_cfg": {"context_length": 77, "vocab_size": 1024, "width": 128, "heads": 2, "layers": 2},
    },
    "small-test": {
        "embed_dim": 128,
        "vision_cfg": {"image_size": 96, "layers": 2, "width": 256, "patch_size": 16},
        "text_cfg": {"context_length": 77, "vocab_size": 1024, "width": 192, "heads": 3, "layers": 2},
    },
}


for _name, _dims in _REFERENCE_VITS.items():
    _MODEL_CONFIGS[_name] = _vit(*_dims)
for _name in _REFERENCE_QUICKGELU:
    _MODEL_CONFIGS[_name + "-quickgelu"] = dict(copy.deepcopy(_MODEL_CONFIGS[_name]), quick_gelu=True)


def list_models():
    return sorted(_MODEL_CONFIGS)


def add_model_config(name, cfg: dict = None):
    """Counterpart of ``open_clip.factory.add_model_config`` (factory.py:80-85).  ``add_model_config(name, cfg_dict)`` registers a dict;
    ``add_model_config(path)`` -- the reference's form -- registers a ``.json`` file or every ``*.json`` of a directory under the file
    stems (files without ``embed_dim`` / ``vision_cfg`` / ``text_cfg`` are skipped as the reference does, factory.py:60-66)."""
    if cfg is not None:
        _MODEL_CONFIGS[name] = copy.deepcopy(cfg)
        return
    path = os.fspath(name)
    files = [path] if os.path.isfile(path) else sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith(".json"))
    for f in files:
        with open(f) as fh:
            d = json.load(fh)
        if all(k in d for k in ("embed_dim", "vision_cfg", "text_cfg")):
            _MODEL_CONFIGS[os.path.splitext(os.path.basename(f))[0]] = d


def get_model_config(name: str) -> dict:
    """factory.py:154-169: returns a deep copy, None-like KeyError if unknown."""
    if name not in _MODEL_CONFIGS:
        raise RuntimeError(f"Model config for {name} not found; available: {list_models()}")
    cfg = copy.deepcopy(_MODEL_CONFIGS[name])
    cfg["vision_cfg"].setdefault("head_width", 64)  # CLIPVisionCfg default (model.py:41)
    cfg["vision_cfg"].setdefault("mlp_ratio", 4.0)
    cfg["text_cfg"].setdefault("mlp_ratio", 4.0)
    return cfg


def vision_tokens(cfg: dict) -> int:
    v = cfg["vision_cfg"]
    g = v["image_size"] // v["patch_size"]
    return g * g + 1


def count_params(cfg: dict) -> int:
    v, t, e = cfg["vision_cfg"], cfg["text_cfg"], cfg["embed_dim"]

    def tower(width, layers):
        per = 4 * width + 3 * width * width + 3 * width + width * width + width + 8 * width * width + 5 * width
        return layers * per

    n = tower(v["width"], v["layers"]) + tower(t["width"], t["layers"])
    n += v["width"] * 3 * v["patch_size"] ** 2 + v["width"] + vision_tokens(cfg) * v["width"] + 4 * v["width"] + v["width"] * e
    n += t["vocab_size"] * t["width"] + t["context_length"] * t["width"] + 2 * t["width"] + t["width"] * e + 1
    return n


def forward_gflops_per_pair(cfg: dict) -> float:
    """algorithmic forward GFLOPs (2 x multiply-accumulates) of one image-text pair as the reference executes it (every caption padded to
    context_length): per block and token (4 + 2 r) C^2 multiply-accumulates for QKV, out-projection and the MLP of ratio r, + 2 L C for the
    two attention products; the patch embedding; the two projections.  Reproduces the `gflops` column of the reference's
    docs/model_profile.csv for all 28 registered configs it lists to 0.1 % (tests/test_reference_dropin.py)."""
    v, t, e = cfg["vision_cfg"], cfg["text_cfg"], cfg["embed_dim"]

    def tower(width, layers, tokens, ratio):
        per_token = (4 + 2 * ratio) * width * width + 2 * tokens * width
        return layers * tokens * per_token

    lv = vision_tokens(cfg)
    macs = tower(v["width"], v["layers"], lv, int(v["width"] * v.get("mlp_ratio", 4.0)) / v["width"])
    macs += (lv - 1) * v["width"] * 3 * v["patch_size"] ** 2 + v["width"] * e
    macs += tower(t["width"], t["layers"], t["context_length"], int(t["width"] * t.get("mlp_ratio", 4.0)) / t["width"]) + t["width"] * e
    return 2 * macs / 1e9
