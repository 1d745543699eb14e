"""CPU, only where /root/reference exists (never on the GPU box): the native modules plug into the REFERENCE's own objects.
  * ``native.load_state_dict(reference_model.state_dict())`` is strict-clean (same names and shapes);
  * ``open_clip.task.CLIPTask(native_model, loss=NativeClipLoss())`` constructs, exposes the native module as its trainable
    module and drives it with the reference's call convention (``image=``, ``text=``) -- on CPU the call must stop at the first
    kernel with the 'no CPU path' error, i.e. nothing in between silently computes;
  * the reference's weight-decay grouping (open_clip_train/optim.py:136-208) applied to the native model partitions the
    parameters exactly like ``param_groups_like_reference``.
The reference is test infrastructure here (imported through oracle/ref_shim.py); the product never imports it."""
import pytest
import torch

from oracle.ref_shim import import_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present")


def _pair():
    open_clip = import_reference()
    from open_clip_amd.model import NativeCLIP
    ref = open_clip.create_model("ViT-B-32", output_dict=True)
    cfg = open_clip.get_model_config("ViT-B-32")
    with torch.device("meta"):
        native = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=True)
    return open_clip, ref, native.to_empty(device="cpu")


def test_native_model_is_a_drop_in_for_the_reference_task_and_optimizer_grouping():
    open_clip, ref, native = _pair()
    missing, unexpected = native.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    for (k, p), (k2, q) in zip(native.named_parameters(), ref.named_parameters()):
        assert k == k2 and torch.equal(p, q)

    from open_clip.task import CLIPTask
    from open_clip_amd.loss import NativeClipLoss
    task = CLIPTask(native, loss=NativeClipLoss(), rank=0, world_size=1, device=torch.device("cpu"), verbose=False)
    assert task.trainable_module is native and isinstance(task.loss, NativeClipLoss)
    batch = {"image": torch.randn(2, 3, 224, 224), "text": torch.randint(0, 49408, (2, 77))}
    with pytest.raises(RuntimeError, match="no CPU path"):
        task.training_forward(batch)

    from open_clip_train.optim import wd_param_groups
    from open_clip_amd.optim import param_groups_like_reference
    ref_groups = wd_param_groups(native, 0.2)
    mine = param_groups_like_reference(native, 0.2)
    by_wd = lambda groups: {g["weight_decay"]: {id(p) for p in g["params"]} for g in groups}
    assert by_wd(ref_groups) == by_wd(mine)
    # and the same partition as the reference model itself gets
    names = {id(p): n for n, p in native.named_parameters()}
    ref_names = {id(p): n for n, p in ref.named_parameters()}
    ref_own = {g["weight_decay"]: {ref_names[id(p)] for p in g["params"]} for g in wd_param_groups(ref, 0.2)}
    assert {wd: {names[i] for i in ids} for wd, ids in by_wd(mine).items()} == ref_own


def test_reference_train_one_epoch_drives_the_native_task():
    """The reference's unmodified ``train_one_epoch`` (open_clip_train/train.py:337) accepts a TrainState built from the native
    model, loss and optimizer and reaches the native forward with the batch it prepared (on CPU: the 'no CPU path' error of
    the first kernel -- proof that nothing between the loop and the kernels computes on its own)."""
    open_clip, ref, native = _pair()
    native.load_state_dict(ref.state_dict(), strict=True)
    from open_clip.task import CLIPTask
    from open_clip_train.distributed import init_distributed_device
    from open_clip_train.params import parse_args
    from open_clip_train.train import TrainState, train_one_epoch
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference
    args = parse_args(["--model", "ViT-B-32", "--precision", "fp32", "--batch-size", "2", "--device", "cpu", "--lr", "5e-4",
                       "--warmup", "2", "--epochs", "1", "--log-every-n-steps", "1", "--skip-scheduler"])
    init_distributed_device(args)
    args.wandb = args.trackio = args.tensorboard = False
    args.distill = False
    task = CLIPTask(native, loss=NativeClipLoss(), rank=0, world_size=1, device=torch.device("cpu"), verbose=False)
    opt = NativeAdamW(param_groups_like_reference(native, args.wd), lr=args.lr, betas=(args.beta1, args.beta2), eps=args.eps)

    class Loader(list):
        num_batches, num_samples = 1, 2

    class Data:
        dataloader = Loader([{"image": torch.randn(2, 3, 224, 224), "text": torch.randint(0, 49408, (2, 77))}])

        def set_epoch(self, e):
            pass

    with pytest.raises(RuntimeError, match="no CPU path"):
        train_one_epoch(TrainState(task=task, optimizer=opt), {"train": Data()}, args)


def test_layer_groups_and_partial_locks_match_the_reference():
    """``visual.layer_groups()`` / the text groups name the same partitions as the reference's (transformer.py:718-743, :1999-2031) and
    ``lock_image_tower(unlocked_groups=k)`` / ``lock_text_tower(unlocked_layers=k)`` freeze exactly the same parameters"""
    open_clip, ref, native = _pair()
    from open_clip.transformer import _text_layer_groups

    def names_by_group(model, groups, prefix=""):
        ids = {id(p): n for n, p in model.named_parameters()}
        out = []
        for gname, members in groups:
            ps = []
            for m in members:
                ps += [m] if isinstance(m, torch.nn.Parameter) else list(m.parameters())
            out.append((gname, sorted(ids[id(p)] for p in ps)))
        return out
    assert names_by_group(native, native.visual.layer_groups()) == names_by_group(ref, ref.visual.layer_groups())
    assert names_by_group(native, native.text_layer_groups()) == names_by_group(ref, _text_layer_groups(ref))
    for k in (0, 1, 3, 14):
        ref.lock_image_tower(unlocked_groups=k)
        native.lock_image_tower(unlocked_groups=k)
        ref.lock_text_tower(unlocked_layers=k)
        native.lock_text_tower(unlocked_layers=k)
        assert {n for n, p in ref.named_parameters() if p.requires_grad} == {n for n, p in native.named_parameters() if p.requires_grad}, k


def test_registered_configs_equal_the_reference_jsons(tmp_path):
    """every reference-named config of the native registry carries exactly the reference JSON's values; every plain-ViT JSON the native
    path can run is registered; add_model_config(path) (the reference's form, factory.py:80-85) loads JSON files"""
    import glob
    import json
    import os
    from open_clip_amd import configs
    from open_clip_amd.model import NativeCLIP
    ref_dir = "/root/reference/src/open_clip/model_configs"
    names = [n for n in configs.list_models() if os.path.exists(os.path.join(ref_dir, n + ".json"))]  # (other tests register their own)
    assert len(names) >= 31
    for n in names:
        with open(os.path.join(ref_dir, n + ".json")) as fh:
            assert json.load(fh) == configs._MODEL_CONFIGS[n], n
    # the other way round: a reference ViT config that passes the native path's own option check must be registered
    missing = []
    for f in sorted(glob.glob(os.path.join(ref_dir, "ViT-*.json"))):
        d = json.load(open(f))
        name = os.path.basename(f)[:-5]
        kw = {k: v for k, v in d.items() if k not in ("embed_dim", "vision_cfg", "text_cfg", "quick_gelu")}
        try:
            NativeCLIP._check_cfg(d["vision_cfg"], d["text_cfg"], kw)
        except NotImplementedError:
            continue
        hw = d["vision_cfg"].get("head_width", 64)
        if hw % 8 == 0 and hw <= 128 and name not in names:
            missing.append(name)
    assert not missing, missing
    # add_model_config(path)
    p = tmp_path / "My-ViT.json"
    p.write_text(json.dumps(configs._MODEL_CONFIGS["ViT-S-32"]))
    (tmp_path / "notes.json").write_text(json.dumps({"comment": "not a model"}))
    configs.add_model_config(str(tmp_path))
    try:
        assert configs.get_model_config("My-ViT")["vision_cfg"]["width"] == 384 and "notes" not in configs.list_models()
    finally:
        configs._MODEL_CONFIGS.pop("My-ViT", None)


def test_analytic_forward_flops_match_the_reference_profile_table():
    """bench.py prices every model against the MFMA roof with configs.forward_gflops_per_pair: it must agree with the reference's own
    docs/model_profile.csv (the figure SURVEY.md 8d quotes) for every registered config the table lists"""
    import csv
    from open_clip_amd import configs
    with open("/root/reference/docs/model_profile.csv") as fh:
        table = {r["model"]: float(r["gflops"]) for r in csv.DictReader(fh)}
    seen = 0
    for name in configs.list_models():
        if name in table:
            got = configs.forward_gflops_per_pair(configs.get_model_config(name))
            assert abs(got / table[name] - 1) < 2e-3, (name, got, table[name])
            seen += 1
    assert seen >= 28


def test_reference_stream_dtypes_under_autocast():
    """What ``NativeCLIP(image_stream="bf16")`` mirrors, read off the REFERENCE itself (round 6): under ``--precision amp_bf16`` (torch.autocast,
    precision.py:6-17) the image tower's residual stream is bf16 from ln_pre on -- conv1 runs under autocast (transformer.py:794), LayerNorm hands its
    result back in its input's dtype (layers.py:20-26), `q_x + attention(...)` adds two bf16 tensors (transformer.py:328-329) -- while the text tower's
    stays fp32 (the token embedding is fp32, model.py:399-401; fp32 + bf16 -> fp32).  CPU autocast applies the same rules as the GPU's."""
    open_clip, ref, _ = _pair()
    seen = {}

    def hook(name):
        def f(mod, inp, out):
            seen[name] = (inp[0].dtype, (out[0] if isinstance(out, tuple) else out).dtype)
        return f
    ref.visual.ln_pre.register_forward_hook(hook("ln_pre"))
    ref.visual.transformer.resblocks[5].register_forward_hook(hook("image block"))
    ref.transformer.resblocks[5].register_forward_hook(hook("text block"))
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ref(image=torch.randn(2, 3, 224, 224), text=torch.randint(0, 49408, (2, 77)))
    assert seen["ln_pre"] == (torch.bfloat16, torch.bfloat16)
    assert seen["image block"] == (torch.bfloat16, torch.bfloat16)
    assert seen["text block"] == (torch.float32, torch.float32)


def test_create_task_mirrors_the_reference_dispatch():
    """open_clip_amd.create_task(args, model) makes the choices of open_clip.factory.create_task (factory.py:975-1043) from the reference's own parsed
    arguments -- task class, loss kind, local_loss / gather_with_grad / rank / world_size / loss_dist_impl -- with the native losses inside the
    reference's task classes; what the native path does not implement raises"""
    open_clip, ref, native = _pair()
    import open_clip_amd
    from open_clip.task import CLIPTask, SigLIPTask
    from open_clip_train.params import parse_args
    from open_clip_amd.loss import NativeClipLoss, NativeSigLipLoss
    args = parse_args(["--model", "ViT-B-32", "--local-loss", "--gather-with-grad"])
    args.rank, args.world_size, args.distill = 3, 8, False  # (main.py:247 derives args.distill before it calls create_task)
    task = open_clip_amd.create_task(args, native, device=torch.device("cpu"), verbose=False)
    ref_task = open_clip.factory.create_task(args, ref)
    assert type(task) is type(ref_task) is CLIPTask and task.trainable_module is native
    assert isinstance(task.loss, NativeClipLoss) and not isinstance(ref_task.loss, NativeClipLoss)
    for k in ("local_loss", "gather_with_grad", "rank", "world_size"):
        assert getattr(task.loss, k) == getattr(ref_task.loss, k), k
    assert not task.loss.row_sharded
    args = parse_args(["--model", "ViT-B-32"])  # global loss over 8 ranks: evaluated row-sharded (same value and gradients as the redundant form)
    args.rank, args.world_size, args.distill = 0, 8, False
    assert open_clip_amd.create_loss(args).row_sharded and not open_clip_amd.create_loss(args, row_sharded=False).row_sharded
    args = parse_args(["--model", "ViT-B-32", "--siglip", "--loss-dist-impl", "shift"])
    args.rank, args.world_size, args.distill = 1, 2, False
    task = open_clip_amd.create_task(args, native, device=torch.device("cpu"), verbose=False)
    ref_task = open_clip.factory.create_task(args, ref)
    assert type(task) is type(ref_task) is SigLIPTask and isinstance(task.loss, NativeSigLipLoss)
    assert (task.loss.dist_impl, task.loss.rank, task.loss.world_size) == (ref_task.loss.dist_impl, 1, 2) == ("shift", 1, 2)
    args.loss_dist_impl = "ring"
    with pytest.raises(ValueError, match="dist_impl"):
        open_clip_amd.create_task(args, native)
    args = parse_args(["--model", "ViT-B-32", "--distill-model", "ViT-B-32", "--distill-pretrained", "x"])
    args.rank, args.world_size, args.distill = 0, 1, True
    with pytest.raises(NotImplementedError, match="distill"):
        open_clip_amd.create_task(args, native)
    args = parse_args(["--model", "coca_ViT-B-32"])
    args.rank, args.world_size, args.distill = 0, 1, False
    with pytest.raises(NotImplementedError, match="CoCa"):
        open_clip_amd.create_loss(args)
    args = parse_args(["--model", "ViT-B-32"])
    args.rank, args.world_size, args.distill = 0, 1, False
    with pytest.raises(TypeError, match="NativeCLIP"):
        open_clip_amd.create_task(args, ref)
