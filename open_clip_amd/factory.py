"""The reference's loss / task dispatch seam for the native path.

``open_clip.factory.create_task(args, model)`` (factory.py:975-1043, called from open_clip_train/main.py:364) builds the task AND its loss from the
training arguments -- and for a ``NativeCLIP`` it would build the ATen ``ClipLoss`` / ``SigLipLoss``.  ``create_task`` / ``create_loss`` here take the
same arguments and make the same choices (``args.siglip`` -> SigLIPTask + sigmoid loss, else CLIPTask + softmax loss; ``args.local_loss``,
``args.gather_with_grad``, ``args.rank``, ``args.world_size``, ``args.loss_dist_impl``) with the NATIVE losses inside the REFERENCE's own task
classes, so ``main.py`` needs one changed import and nothing else.  The task classes come from the ``open_clip`` package the caller has installed
(imported here, at call time: this package never depends on it otherwise).  What the native path does not implement raises instead of silently
building something else: distillation, CoCa, GenLIP / GenLAP, CLAP (SURVEY.md section 8 marks them out of scope).
"""
from typing import Optional


def _unsupported(args):
    name = str(getattr(args, "model", "")).lower()
    if getattr(args, "distill", False):
        return "--distill (DistillClipLoss / DistillCLIPTask)"
    for key, what in (("coca", "CoCa (CoCaLoss / CoCaTask)"), ("genlap", "GenLAP"), ("genlip", "GenLIP")):
        if key in name:
            return what
    return None


def create_loss(args, comm=None, deterministic: bool = False, row_sharded: Optional[bool] = None):
    """``open_clip.factory.create_loss`` (factory.py:930-972) for the native losses.  ``cache_labels`` has no counterpart (labels are never
    materialised: the kernels compare against ``label_offset + row``).  ``comm``: a ``NativeComm`` (RCCL through the C ABI) for the loss collectives
    instead of torch.distributed's process group.  ``row_sharded`` (ClipLoss with world_size > 1 and ``local_loss=False``): every rank evaluates only
    its own rows of the global logits both ways -- same value and gradients as the reference's redundant form; default on when distributed."""
    from .loss import NativeClipLoss, NativeSigLipLoss
    bad = _unsupported(args)
    if bad is not None:
        raise NotImplementedError(f"open_clip_amd.create_loss: {bad} is not implemented by the native path; use open_clip.factory.create_loss")
    rank, world = int(getattr(args, "rank", 0)), int(getattr(args, "world_size", 1))
    if getattr(args, "siglip", False):
        return NativeSigLipLoss(rank=rank, world_size=world, dist_impl=getattr(args, "loss_dist_impl", None), comm=comm, deterministic=deterministic)
    local_loss = bool(getattr(args, "local_loss", False))
    if row_sharded is None:
        row_sharded = world > 1 and not local_loss
    return NativeClipLoss(local_loss=local_loss, gather_with_grad=bool(getattr(args, "gather_with_grad", False)), rank=rank, world_size=world,
                          row_sharded=bool(row_sharded) and world > 1 and not local_loss, comm=comm, deterministic=deterministic)


def create_task(args, model, dist_model=None, naflex_data_config=None, comm=None, deterministic: bool = False, **task_kwargs):
    """``open_clip.factory.create_task`` (factory.py:975-1043): the reference's ``CLIPTask`` / ``SigLIPTask`` around ``model`` with the native loss that
    ``create_loss(args)`` selects.  ``task_kwargs`` go to the task constructor (``device=``, ``dtype=``, ``verbose=``)."""
    from .model import NativeCLIP
    if not isinstance(model, NativeCLIP):
        raise TypeError(f"open_clip_amd.create_task wraps a NativeCLIP (got {type(model).__name__}); other models go through open_clip.factory.create_task")
    if dist_model is not None:
        raise NotImplementedError("open_clip_amd.create_task: distillation (dist_model) is not implemented by the native path")
    loss = create_loss(args, comm=comm, deterministic=deterministic)
    try:
        from open_clip.task import CLIPTask, SigLIPTask
    except ImportError as e:  # the task classes are the reference's own: this package restates none of them
        raise ImportError("open_clip_amd.create_task needs the reference's `open_clip` package (open_clip.task.CLIPTask / SigLIPTask) on sys.path") from e
    shared = dict(rank=int(getattr(args, "rank", 0)), world_size=int(getattr(args, "world_size", 1)))
    cls = SigLIPTask if getattr(args, "siglip", False) else CLIPTask
    task = cls(model, loss=loss, **shared, **task_kwargs)
    if naflex_data_config is not None:
        task.set_naflex_data_config(naflex_data_config)
    return task
