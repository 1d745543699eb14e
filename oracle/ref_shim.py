"""Import shim that makes the *reference* (``/root/reference/src``) importable in the
build container (torchvision / ftfy are absent there).  TEST INFRASTRUCTURE ONLY.

Used exclusively by ``oracle/make_golden.py`` (fixture generation) and by CPU tests that are
skipped when ``/root/reference`` does not exist (it never exists on the GPU box).

Recipe validated in SURVEY.md Appendix A: ``transformers`` must be imported before the
``torchvision`` stub exists (its ``find_spec('torchvision')`` raises on a spec-less stub).
"""
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "open_clip"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the reference ``open_clip`` package (imported from /root/reference/src)."""
    if "open_clip" in sys.modules and getattr(sys.modules["open_clip"], "__file__", "").startswith(REFERENCE_SRC):
        return sys.modules["open_clip"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_SRC)
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    import torch

    try:
        import transformers  # noqa: F401  (must precede the torchvision stub)
    except Exception:  # pragma: no cover
        pass
    if "torchvision" not in sys.modules:

        class _Dummy:
            def __init__(self, *a, **k):
                pass

            def __call__(self, x):
                return x

        class FrozenBatchNorm2d(torch.nn.Module):
            def __init__(self, n):
                super().__init__()

        class InterpolationMode:
            BICUBIC = "bicubic"
            BILINEAR = "bilinear"
            NEAREST = "nearest"
            LANCZOS = "lanczos"
            BOX = "box"
            HAMMING = "hamming"

        tv = _mod("torchvision")
        ops = _mod("torchvision.ops")
        misc = _mod("torchvision.ops.misc", FrozenBatchNorm2d=FrozenBatchNorm2d)
        names = ["Normalize", "Compose", "RandomResizedCrop", "ToTensor", "Resize", "CenterCrop",
                 "ColorJitter", "Grayscale", "RandomApply"]
        tr = _mod("torchvision.transforms", InterpolationMode=InterpolationMode, **{n: _Dummy for n in names})
        trf = _mod("torchvision.transforms.functional")
        tv.ops, ops.misc, tv.transforms, tr.functional = ops, misc, tr, trf
    if "ftfy" not in sys.modules:
        _mod("ftfy", fix_text=lambda s: s)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import open_clip  # noqa: E402

    return open_clip
