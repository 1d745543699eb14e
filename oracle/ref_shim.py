"""Import shim that makes the *reference* (``/root/reference/src``) importable in the
build container (torchvision / ftfy are absent there).  TEST INFRASTRUCTURE ONLY.

Used by ``oracle/make_golden.py`` (fixture generation), by CPU tests that are skipped when ``/root/reference`` does not
exist, and by ``oracle/ref_cpu_baseline.py`` -- the reference's own train loop timed on the host cores.  On the GPU box
/root/reference does not exist: there the packages come out of ``oracle/_ref/reference_src.zip`` (packed by
``oracle/fetch_ref.py`` in the build container, git-ignored, shipped with the snapshot like the built ``.so`` files),
unpacked into a temporary directory for the life of the process.

Recipe validated in SURVEY.md Appendix A: ``transformers`` must be imported before the
``torchvision`` stub exists (its ``find_spec('torchvision')`` raises on a spec-less stub).
"""
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"
ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference_src.zip")
_unpacked = None


def reference_available() -> bool:
    """the reference TREE (sources, docs, model_profile.csv): only in the build container"""
    return os.path.isdir(os.path.join(REFERENCE_SRC, "open_clip"))


def reference_importable() -> bool:
    """the tree, or the archive of its two packages that travels to the GPU box"""
    return reference_available() or os.path.exists(ARCHIVE)


def reference_src() -> str:
    """directory to put on sys.path: /root/reference/src, or the archive unpacked into a temporary directory (removed at exit)"""
    global _unpacked
    if reference_available():
        return REFERENCE_SRC
    if _unpacked is None:
        if not os.path.exists(ARCHIVE):
            raise RuntimeError("reference not present: neither %s nor %s (python -m oracle.fetch_ref packs it in the build container)" % (REFERENCE_SRC, ARCHIVE))
        import atexit
        import shutil
        import tempfile
        import zipfile
        _unpacked = tempfile.mkdtemp(prefix="ocn_reference_src_")
        atexit.register(shutil.rmtree, _unpacked, True)
        with zipfile.ZipFile(ARCHIVE) as z:
            z.extractall(_unpacked)
    return _unpacked


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the reference ``open_clip`` package (imported from /root/reference/src)."""
    src = reference_src()
    if "open_clip" in sys.modules and getattr(sys.modules["open_clip"], "__file__", "").startswith(src):
        return sys.modules["open_clip"]
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
    if src not in sys.path:
        sys.path.insert(0, src)
    import open_clip  # noqa: E402

    return open_clip
