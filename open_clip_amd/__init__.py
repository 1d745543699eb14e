"""open_clip_amd: MI355X-native (gfx950) CLIP training hot path behind open_clip's model / loss API.

Public surface (mirrors the reference names):
    create_model, NativeCLIP            <- open_clip.factory.create_model / open_clip.model.CLIP
    NativeClipLoss, NativeSigLipLoss    <- open_clip.loss.ClipLoss / SigLipLoss
    get_model_config, add_model_config  <- open_clip.factory.get_model_config / add_model_config
    create_task, create_loss            <- open_clip.factory.create_task / create_loss (the reference's task classes around the native losses)
"""
from .configs import add_model_config, get_model_config, list_models  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require torch.cuda / the shared library
    if name in ("create_model", "NativeCLIP"):
        from . import model
        return getattr(model, name)
    if name in ("NativeClipLoss", "NativeSigLipLoss"):
        from . import loss
        return getattr(loss, name)
    if name in ("create_task", "create_loss"):
        from . import factory
        return getattr(factory, name)
    if name in ("NativeAdamW",):
        from . import optim
        return getattr(optim, name)
    raise AttributeError(name)
