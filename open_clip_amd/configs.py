"""Model-config registry for the native CLIP path.

Mirrors the reference's JSON registry (``src/open_clip/model_configs/*.json``, looked up by
``factory.get_model_config`` factory.py:154-169): same keys (``embed_dim``, ``vision_cfg``,
``text_cfg``) and the same defaults the reference dataclasses apply
(``CLIPVisionCfg.head_width = 64`` model.py:37-60, ``CLIPTextCfg`` model.py:108-150).
Only the configs BASELINE.json names are registered, plus tiny ones for parity tests.
"""
import copy

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
    "small-test": {
        "embed_dim": 128,
        "vision_cfg": {"image_size": 96, "layers": 2, "width": 256, "patch_size": 16},
        "text_cfg": {"context_length": 77, "vocab_size": 1024, "width": 192, "heads": 3, "layers": 2},
    },
}


def list_models():
    return sorted(_MODEL_CONFIGS)


def add_model_config(name: str, cfg: dict):
    """Counterpart of ``open_clip.factory.add_model_config`` (factory.py:80-85) for dict configs."""
    _MODEL_CONFIGS[name] = copy.deepcopy(cfg)


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
