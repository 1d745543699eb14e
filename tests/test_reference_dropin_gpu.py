"""The REFERENCE's own objects driving the native path ON THE MI355X: ``open_clip.task.CLIPTask`` (clip_task.py:26-46) wrapped around NativeCLIP +
NativeClipLoss, stepped by the unmodified ``open_clip_train.train.train_one_epoch`` (train.py:337: prepare_batch -> autocast(bf16) -> training_forward
-> backward -> clip_grad_norm_ -> optimizer.step -> clamp_logit_scale) with NativeAdamW, against the CPU oracle stepped with torch.optim.AdamW on the
same batches.  The reference's packages come from /root/reference (build container) or from ``oracle/_ref/reference_src.zip`` (the GPU box:
oracle/fetch_ref.py); without either the test is skipped and tests/test_model_gpu.py::test_reference_train_loop_under_autocast_tracks_the_cpu_oracle
-- the same loop restated line by line -- stands in.  The reference is test infrastructure here: the product never imports it."""
import logging
import math
import re

import pytest
import torch

from open_clip_amd.configs import get_model_config
from open_clip_amd.synth import init_state_dict, synthetic_batch
from oracle.ref_shim import import_reference, reference_importable
from tests.test_kernels_gpu import _report

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not reference_importable(), reason="reference packages not available (no /root/reference, no oracle/_ref archive)")]


def test_reference_train_one_epoch_trains_the_native_model_on_the_gpu():
    import_reference()
    from open_clip.task import CLIPTask
    from open_clip_train.distributed import init_distributed_device
    from open_clip_train.params import parse_args
    from open_clip_train.train import TrainState, train_one_epoch
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    from oracle import clip_oracle as O
    from tests.test_model_gpu import LOSS_TOL, _build
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=61, perturb=True)
    lr, wd, clip, steps, bs = 2e-3, 0.2, 1.0, 3, 8
    args = parse_args(["--model", "ViT-B-32", "--precision", "amp_bf16", "--batch-size", str(bs), "--device", "cuda", "--lr", str(lr), "--wd", str(wd),
                       "--warmup", "0", "--epochs", "1", "--log-every-n-steps", "1", "--skip-scheduler", "--grad-clip-norm", str(clip)])
    device = init_distributed_device(args)
    args.wandb = args.trackio = args.tensorboard = False
    args.distill = False
    model = _build(cfg, state)
    task = CLIPTask(model, loss=NativeClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=0, world_size=1), rank=0, world_size=1,
                    device=torch.device(device), verbose=False)
    assert task.trainable_module is model
    opt = NativeAdamW(param_groups_like_reference(model, wd), lr=lr, betas=(args.beta1, args.beta2), eps=args.eps, weight_caches=weight_caches_of(model))
    host = [{k: v.pin_memory() for k, v in synthetic_batch(cfg, bs, seed=400 + i).items()} for i in range(steps)]

    class Loader(list):
        num_batches, num_samples = steps, steps * bs

    class Data:
        dataloader = Loader(host)

        def set_epoch(self, e):
            pass

    losses = []

    class Lines(logging.Handler):
        def emit(self, record):
            m = re.search(r"^Train Epoch: .* Contrastive_loss: ([0-9.]+)", record.getMessage())  # (not the "End epoch ... Avg Contrastive_loss" summary)
            if m:
                losses.append(float(m.group(1)))

    h = Lines(level=logging.INFO)
    logging.getLogger().addHandler(h)
    old_level = logging.getLogger().level
    logging.getLogger().setLevel(logging.INFO)
    try:
        st = TrainState(task=task, optimizer=opt)
        train_one_epoch(st, {"train": Data()}, args)
    finally:
        logging.getLogger().removeHandler(h)
        logging.getLogger().setLevel(old_level)
    torch.cuda.synchronize()
    assert st.global_step == steps and st.samples_seen == steps * bs
    # the CPU oracle stepped with torch.optim.AdamW on the same batches (reference grouping: optim.py:67-77)
    ref_params = {k: torch.nn.Parameter(v.clone().float()) for k, v in state.items()}
    skip = {"positional_embedding", "visual.positional_embedding", "visual.class_embedding"}
    ref_opt = torch.optim.AdamW([{"params": [p for k, p in ref_params.items() if p.ndim <= 1 or k in skip], "weight_decay": 0.0},
                                 {"params": [p for k, p in ref_params.items() if not (p.ndim <= 1 or k in skip)], "weight_decay": wd}],
                                lr=lr, betas=(args.beta1, args.beta2), eps=args.eps)
    ref_losses = []
    for b in host:
        outs, grads = O.train_forward_backward(b["image"], b["text"], {k: p.detach() for k, p in ref_params.items()}, cfg)
        for k, p in ref_params.items():
            p.grad = grads[k]
        torch.nn.utils.clip_grad_norm_(list(ref_params.values()), clip)
        ref_opt.step()
        with torch.no_grad():
            ref_params["logit_scale"].clamp_(0, math.log(100))
        ref_losses.append(float(outs["loss"]))
    _report(f"reference train_one_epoch over the native model: losses {losses} oracle {[round(v, 5) for v in ref_losses]}")
    assert len(losses) == steps, losses  # one console line per step (--log-every-n-steps 1)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= LOSS_TOL + 5e-5 * abs(b), (losses, ref_losses)  # (the console prints 5 significant digits)
    worst = (0.0, "")
    for k, p in model.named_parameters():
        ref, got, start = ref_params[k].detach(), p.detach().float().cpu(), state[k].float()
        if k.endswith("attn.in_proj_bias"):  # the K third has an exactly-zero gradient: Adam moves it by rounding noise alone
            c = ref.numel() // 3
            keep = torch.cat([torch.arange(0, c), torch.arange(2 * c, 3 * c)])
            ref, got, start = ref[keep], got[keep], start[keep]
        ratio = float((got - ref).norm()) / (float((ref - start).norm()) + 1e-30)
        worst = max(worst, (ratio, k))
        assert ratio <= 0.25 + 1e-6, (k, ratio)
    _report(f"reference train_one_epoch over the native model: error / movement of the parameters after {steps} steps: worst {worst[0]:.3f} ({worst[1]}); bound 0.25")


@pytest.mark.parametrize("siglip", [False, True])
def test_create_task_from_parsed_args_steps_the_native_path(siglip):
    """the reference's dispatch seam end to end -- ``parse_args`` -> ``open_clip_amd.create_task(args, model)`` (the counterpart of
    open_clip/factory.py:975-1043 that open_clip_train/main.py:364 would call) -> ``task.training_forward`` under autocast -> backward, on the MI355X:
    the task is the reference's own class, the loss inside it the native one, and loss + gradients are those of the hand-constructed objects"""
    import_reference()
    import open_clip_amd
    from open_clip.task import CLIPTask, SigLIPTask
    from open_clip_train.params import parse_args
    from open_clip_amd.loss import NativeClipLoss, NativeSigLipLoss
    from tests.test_model_gpu import _build, _step
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=62, perturb=True, siglip=siglip)
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, 12, seed=500).items()}
    args = parse_args(["--model", "ViT-B-32", "--precision", "amp_bf16", "--local-loss", "--gather-with-grad"] + (["--siglip", "--loss-dist-impl", "gather"] if siglip else []))
    args.rank, args.world_size, args.distill = 0, 1, False  # (main.py:247 derives args.distill before it calls create_task)
    model = _build(cfg, state, siglip=siglip)
    task = open_clip_amd.create_task(args, model, device=torch.device("cuda"), verbose=False)
    assert type(task) is (SigLIPTask if siglip else CLIPTask) and isinstance(task.loss, NativeSigLipLoss if siglip else NativeClipLoss)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        losses, _ = task.training_forward(batch)
    losses["loss"].backward()
    torch.cuda.synchronize()
    ref_model = _build(cfg, state, siglip=siglip)
    _, ref_loss = _step(ref_model, batch, siglip=siglip)
    assert abs(float(losses["loss"]) - float(ref_loss)) <= 1e-6 * max(1.0, abs(float(ref_loss)))
    worst = max(float((p.grad - q.grad).norm() / q.grad.norm().clamp_min(1e-30)) for p, q in zip(model.parameters(), ref_model.parameters()))
    _report(f"create_task[{'siglip' if siglip else 'clip'}]: loss {float(losses['loss']):.6f} vs hand-built {float(ref_loss):.6f}, worst gradient rel_l2 {worst:.2e}")
    assert worst <= 2e-6
