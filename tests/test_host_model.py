"""CPU: host-side mirror of the reference interface (names, shapes, state-dict layout, API surface) and the
'no fallback' rule.  Compute is never executed here."""
import numpy as np
import pytest
import torch

from open_clip_amd.configs import count_params, get_model_config
from open_clip_amd.model import NativeCLIP, create_model
from open_clip_amd.synth import init_state_dict, synthetic_batch
from tests.golden_util import load


def _tiny(**kw):
    cfg = get_model_config("tiny-test")
    return NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], **kw), cfg


def test_state_dict_layout_matches_reference_fixture():
    """keys/shapes == what the reference's CLIP produced (fixture written by oracle/make_golden.py)"""
    g = load("tiny_clip.npz")
    ref = {k[2:]: tuple(v.shape) for k, v in g.items() if k.startswith("w/")}
    model, _ = _tiny()
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert mine == ref
    assert "attn_mask" not in mine and model.attn_mask.shape == (16, 16)  # non-persistent buffer (model.py:360)
    names = [n for n, _ in model.named_parameters()]
    assert sorted(names) == sorted(ref)


def test_vitb32_parameter_count_and_keys():
    cfg = get_model_config("ViT-B-32")
    sd = init_state_dict(cfg, 0)
    assert len(sd) == 302
    assert sum(v.numel() for v in sd.values()) == 151277313 == count_params(cfg)
    with torch.device("meta"):
        m = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}


def test_load_state_dict_roundtrip_and_siglip_bias():
    model, cfg = _tiny(init_logit_bias=-10.0, init_logit_scale=float(np.log(10)))
    sd = init_state_dict(cfg, 3, perturb=True, siglip=True)
    model.load_state_dict(sd, strict=True)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert model.logit_bias is not None and model.logit_scale.ndim == 0


def test_reference_api_surface():
    model, cfg = _tiny(output_dict=True)
    assert model.no_weight_decay() == {"positional_embedding", "visual.positional_embedding", "visual.class_embedding"}
    assert model.visual.image_size == (64, 64) and model.visual.grid_size == (2, 2)
    assert model.context_length == 16 and model.vocab_size == 512 and model.embed_dim == 64
    assert model.transformer.get_cast_dtype() == torch.float32
    model.set_grad_checkpointing(True)
    assert model.visual.transformer.grad_checkpointing and model.transformer.grad_checkpointing
    assert model.visual.transformer.keep_last_blocks == 0 and model.transformer.keep_last_blocks == 0  # the reference recomputes every block
    model.lock_image_tower()
    assert not any(p.requires_grad for p in model.visual.parameters())
    assert model.logit_scale.requires_grad and model.text_projection.requires_grad
    for blk in model.transformer.resblocks:  # discoverable block type (FSDP shard units, base_task.py:245-254)
        assert type(blk).__name__ == "ResidualAttentionBlock"


def test_forward_on_cpu_fails_loudly():
    model, cfg = _tiny(output_dict=True)
    batch = synthetic_batch(cfg, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(image=batch["image"], text=batch["text"])
    from open_clip_amd.loss import NativeClipLoss
    with pytest.raises(RuntimeError, match="no CPU path"):
        NativeClipLoss()(torch.randn(4, 64), torch.randn(4, 64), torch.tensor(10.0))


def test_create_model_rejects_unsupported():
    with pytest.raises(RuntimeError, match="not found"):
        create_model("RN50", device="cpu")
    m = create_model("ViT-H-14", device="meta")  # head_dim 80: the generic attention kernels (csrc/attention_generic.hip)
    assert m.visual.transformer.resblocks[0].attn.head_dim == 80 and m.visual.transformer.resblocks[0].n_head == 16
    from open_clip_amd.configs import add_model_config
    odd = get_model_config("tiny-test")
    odd["vision_cfg"].update(width=144, head_width=72)
    add_model_config("odd-head", odd)
    with pytest.raises(NotImplementedError, match="head_dim 64 / 80 / 88 / 96 / 104 / 112 / 128"):
        create_model("odd-head", device="meta")
    with pytest.raises(ValueError, match="precision"):
        create_model("tiny-test", precision="fp16", device="cpu")


def test_synthetic_batch_layout():
    cfg = get_model_config("ViT-B-32")
    b = synthetic_batch(cfg, 16, seed=1234)
    t = b["text"]
    assert t.shape == (16, 77) and t.dtype == torch.int64 and b["image"].shape == (16, 3, 224, 224)
    assert (t[:, 0] == 49406).all() and ((t == 49407).sum(-1) == 1).all()
    eot = t.argmax(-1)
    assert (eot >= 8).all() and all((t[i, eot[i] + 1:] == 0).all() for i in range(16))


def test_checkpoint_file_roundtrip_and_custom_text_key_layout(tmp_path):
    """8f-4: a checkpoint written to disk loads back bit-equal through ``create_model(pretrained=...)`` in the layouts the reference
    produces: bare ``CLIP`` keys, a ``{'state_dict': ...}`` wrapper with DDP's ``module.`` prefix, and ``CustomTextCLIP``'s
    ``text.*`` keys (model.py:772-787 run backwards); the forward converter reproduces the reference's mapping."""
    from open_clip_amd.configs import add_model_config
    from open_clip_amd.model import convert_from_custom_text_state_dict, convert_to_custom_text_state_dict
    cfg = get_model_config("tiny-test")
    add_model_config("tiny-test-ckpt", cfg)
    sd = init_state_dict(cfg, 5, perturb=True)
    custom = convert_to_custom_text_state_dict(sd)
    assert "text.token_embedding.weight" in custom and "text.transformer.resblocks.0.ln_1.weight" in custom
    assert "visual.conv1.weight" in custom and "logit_scale" in custom and "text_projection" not in custom
    back = convert_from_custom_text_state_dict(dict(custom, **{"text.attn_mask": torch.zeros(16, 16)}))
    assert back.keys() == sd.keys() and all(torch.equal(back[k], sd[k]) for k in sd)
    variants = {"plain.pt": sd, "wrapped.pt": {"state_dict": {"module." + k: v for k, v in sd.items()}, "epoch": 3}, "custom.pt": custom}
    for name, obj in variants.items():
        path = str(tmp_path / name)
        torch.save(obj, path)
        m = create_model("tiny-test-ckpt", pretrained=path, device="cpu")
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd[k]), (name, k)
    with pytest.raises(KeyError, match="unexpected text-tower key"):
        convert_from_custom_text_state_dict({"text.proj.weight": torch.zeros(1)})


def test_optimizer_state_dict_roundtrip():
    """NativeAdamW is a torch.optim.Optimizer: its state (step, exp_avg, exp_avg_sq) and param groups survive
    state_dict() / load_state_dict() (what task/checkpoint.py stores)."""
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference
    model, _ = _tiny()
    opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=1e-3)
    for p in model.parameters():
        st = opt.state[p]
        st["step"], st["exp_avg"], st["exp_avg_sq"] = 7, torch.full_like(p, 0.5), torch.full_like(p, 0.25)
    blob = opt.state_dict()
    opt2 = NativeAdamW(param_groups_like_reference(model, 0.2), lr=5e-4)
    opt2.load_state_dict(blob)
    assert opt2.param_groups[0]["lr"] == 1e-3 and opt2.param_groups[1]["weight_decay"] == 0.2
    for p in model.parameters():
        assert opt2.state[p]["step"] == 7 and torch.equal(opt2.state[p]["exp_avg"], torch.full_like(p, 0.5))


def test_layer_groups_partial_unlock_and_fsdp_units():
    """transformer.py:718-753, :1999-2053, base_task.py:234-250 on the native containers: complete ordered partitions, top-down unlock
    counts (projection head first), idempotent re-locking, one FSDP shard unit per residual block"""
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.model import NativeCLIP
    cfg = get_model_config("tiny-test")
    m = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"])
    vg, tg = m.visual.layer_groups(), m.text_layer_groups()
    assert [n for n, _ in vg] == ["embeddings", "layer.0", "layer.1", "proj"] and [n for n, _ in tg] == ["embeddings", "layer.0", "layer.1", "proj"]

    def params_of(groups):
        out = []
        for _, members in groups:
            for x in members:
                out += [x] if isinstance(x, torch.nn.Parameter) else list(x.parameters())
        return out
    assert {id(p) for p in params_of(vg)} == {id(p) for p in m.visual.parameters()}
    text_params = {id(p) for n, p in m.named_parameters() if not n.startswith("visual.") and n not in ("logit_scale", "logit_bias")}
    assert {id(p) for p in params_of(tg)} == text_params
    m.lock_image_tower(unlocked_groups=2)  # proj + the last block (with ln_post) stay trainable
    train = {n for n, p in m.visual.named_parameters() if p.requires_grad}
    assert "proj" in train and "ln_post.weight" in train and "transformer.resblocks.1.mlp.c_fc.weight" in train
    assert not any(n.startswith(("conv1", "class_embedding", "positional_embedding", "ln_pre", "transformer.resblocks.0.")) for n in train)
    m.lock_image_tower(unlocked_groups=0)
    assert not any(p.requires_grad for p in m.visual.parameters())
    m.lock_text_tower(unlocked_layers=1)
    assert m.text_projection.requires_grad and not m.ln_final.weight.requires_grad and not m.token_embedding.weight.requires_grad
    assert m.logit_scale.requires_grad
    units = m.fsdp_shard_modules()
    assert len(units) == 4 and all(n.endswith(("resblocks.0", "resblocks.1")) for n, _ in units)
    assert m.visual.preprocess_cfg["mean"] == m.visual.image_mean and m.visual.preprocess_cfg["size"] == (64, 64)


def test_unsupported_options_fail_loudly_and_init_follows_the_reference_distributions():
    import math
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.model import NativeCLIP, create_model
    cfg = get_model_config("tiny-test")
    with pytest.raises(NotImplementedError, match="nonscalar_logit_scale"):
        NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], nonscalar_logit_scale=True)
    from open_clip_amd import ops
    q = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], quick_gelu=True)  # the `*-quickgelu` configs (layers.py:29-32)
    assert q.quick_gelu and all(b.act_epilogue == ops.EPI_BIAS_QUICKGELU for b in list(q.visual.transformer.resblocks) + list(q.transformer.resblocks))
    assert create_model("ViT-B-32-quickgelu", device="meta").quick_gelu and not create_model("ViT-B-32", device="meta").quick_gelu
    assert create_model("ViT-B-32", device="meta", force_quick_gelu=True).quick_gelu
    with pytest.raises(NotImplementedError, match="pool_type"):
        NativeCLIP(cfg["embed_dim"], dict(cfg["vision_cfg"], pool_type="avg"), cfg["text_cfg"])
    with pytest.raises(NotImplementedError, match="no_causal_mask"):
        NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], dict(cfg["text_cfg"], no_causal_mask=True))
    NativeCLIP(cfg["embed_dim"], dict(cfg["vision_cfg"], pool_type="tok", patch_dropout=0.0), dict(cfg["text_cfg"], pool_type="argmax"), quick_gelu=False)
    with pytest.raises(ValueError, match="amp_bf16"):
        create_model("tiny-test", precision="fp32", device="cpu")
    torch.manual_seed(0)
    big = get_model_config("ViT-B-32")
    m = NativeCLIP(big["embed_dim"], big["vision_cfg"], big["text_cfg"])
    tw, tl = 512, 12
    blk = m.transformer.resblocks[3]
    assert abs(float(blk.attn.in_proj_weight.std()) - tw ** -0.5) < 0.03 * tw ** -0.5
    assert abs(float(blk.mlp.c_fc.weight.std()) - (2 * tw) ** -0.5) < 0.03 * (2 * tw) ** -0.5
    assert abs(float(blk.attn.out_proj.weight.std()) - tw ** -0.5 * (2 * tl) ** -0.5) < 0.03 * tw ** -0.5
    assert float(blk.attn.out_proj.bias.abs().max()) == 0.0 and float(blk.attn.in_proj_bias.abs().max()) == 0.0
    assert abs(float(m.token_embedding.weight.std()) - 0.02) < 1e-3 and abs(float(m.logit_scale) - math.log(1 / 0.07)) < 1e-6
    vb = m.visual.transformer.resblocks[0]
    assert float(vb.attn.in_proj_weight.abs().max()) <= math.sqrt(6.0 / (4 * 768)) + 1e-6 and float(vb.attn.out_proj.bias.abs().max()) == 0.0


def test_load_state_dict_drops_the_cached_operand_copies():
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.model import NativeCLIP
    cfg = get_model_config("tiny-test")
    m = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"])
    m._cache._d[("x", "n")] = (0, (1,), None)
    m.visual._cache._d[("y", "n")] = (0, (1,), None)
    m.load_state_dict(m.state_dict())
    assert not m._cache._d and not m.visual._cache._d


def test_execution_switches_are_constructor_arguments_and_copies_start_with_empty_caches():
    """the switches of the native execution are keyword arguments / attributes (no environment variables are read); a deep copy of the
    model (the EMA twin of base_task.py:171) does not duplicate the cached bf16 operand copies"""
    import copy
    import inspect
    import open_clip_amd.model as M
    assert "environ" not in inspect.getsource(M), "model.py must not read environment switches"
    cfg = get_model_config("tiny-test")
    m = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], pack_text=False, tower_streams=False, pooled_last_block=False,
                   attn_buckets=False, pair_wgrad=False, deterministic=True)
    assert (m.pack_text, m.tower_streams, m.pooled_last_block, m.visual.pooled_last_block, m.attn_buckets, m.pair_wgrad, m.deterministic) == \
        (False, False, False, False, False, False, True)
    d = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"])
    assert (d.pack_text, d.tower_streams, d.pooled_last_block, d.visual.pooled_last_block, d.attn_buckets, d.pair_wgrad, d.deterministic) == \
        (True, True, True, True, True, True, False)
    d._cache._d[("x", "n")] = (0, (1,), torch.zeros(3))
    d.visual._cache.deterministic = True
    twin = copy.deepcopy(d)
    assert twin._cache._d == {} and twin.visual._cache._d == {} and twin.visual._cache.deterministic is True
    assert ("x", "n") in d._cache._d  # the original keeps its copies


def test_create_model_explicit_arguments_win_over_config_values():
    """a registered config may carry init_logit_scale / init_logit_bias / output_dict itself (the reference's SigLIP JSONs do): the explicit
    create_model argument overrides it, as in the reference (factory.py:547-556), instead of a duplicate-keyword TypeError"""
    import math
    from open_clip_amd.configs import add_model_config
    cfg = get_model_config("tiny-test")
    cfg = dict(cfg, init_logit_bias=-10.0, init_logit_scale=math.log(10), output_dict=True)
    add_model_config("tiny-siglip-cfg", cfg)
    m = create_model("tiny-siglip-cfg", device="meta")
    assert m.logit_bias is not None and m.output_dict is True
    m = create_model("tiny-siglip-cfg", device="meta", init_logit_bias=-3.0, init_logit_scale=1.5, output_dict=False)
    assert m.output_dict is False and m.logit_bias is not None


def test_last_block_hooks_fire_in_the_pooled_form(monkeypatch):
    """the pooled form of the last block goes through Module.__call__ (FSDP2's unshard / user hooks sit there); the kernels are replaced
    by stand-ins here, only the module plumbing is under test"""
    import open_clip_amd.model as M

    class FakeBlock:
        @staticmethod
        def apply(x, *a):
            return x

    class FakePooled:
        @staticmethod
        def apply(x, *a):
            rows = a[12]
            return x[rows.long()]

    monkeypatch.setattr(M, "_BlockFn", FakeBlock)
    monkeypatch.setattr(M, "_PooledBlockFn", FakePooled)
    t = M.Transformer(64, 3, 1)
    fired = []
    t.resblocks[-1].register_forward_pre_hook(lambda mod, args: fired.append("pre"))
    t.resblocks[-1].register_forward_hook(lambda mod, args, out: fired.append("post"))
    x = torch.randn(10, 64)
    y = t(x, None, 2, 5, False, None, torch.tensor([0, 5], dtype=torch.int32))
    assert fired == ["pre", "post"] and y.shape == (2, 64)


def test_partial_recompute_plan():
    """keep_last / plan_grad_checkpointing (native extension of set_grad_checkpointing): pure host arithmetic"""
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.model import NativeCLIP
    cfg = get_model_config("ViT-H-14")
    with torch.device("meta"):
        model = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"])
    model.set_grad_checkpointing(True, keep_last=(3, 5))
    assert (model.visual.transformer.keep_last_blocks, model.transformer.keep_last_blocks) == (3, 5)
    model.set_grad_checkpointing(True, keep_last=2)
    assert (model.visual.transformer.keep_last_blocks, model.transformer.keep_last_blocks) == (2, 2)
    bv, bt = model.activation_bytes_per_block(1024, text_rows=44000)
    assert abs(bv / (1024 * 257 * 1280 * 32) - 1) < 0.01 and abs(bt / (44000 * 1024 * 32) - 1) < 0.01
    nv, nt = 32, 24
    assert model.plan_grad_checkpointing(1024, 10 ** 9, text_rows=44000) == (0, 0)          # nothing fits: recompute every block
    assert model.plan_grad_checkpointing(1024, 10 ** 13, text_rows=44000) == (nv, nt)      # everything fits
    kv, kt = model.plan_grad_checkpointing(1024, 228 * 10 ** 9, text_rows=44000)
    assert kt == nt and 8 <= kv <= 16
    # the plan's own accounting stays inside the budget: kept blocks in full, block inputs of the others, two blocks of transients
    used = kv * bv + (nv - kv) * bv // 8 + kt * bt + (nt - kt) * bt // 8 + 2 * max(bv, bt)
    assert used <= 228 * 10 ** 9
    assert model.visual.transformer.grad_checkpointing and model.visual.transformer.keep_last_blocks == kv


def test_host_text_plan_layout():
    """input_pipeline.HostTextPlan (the packed text layout computed next to the tokenizer) against a plain loop: rows up to the FIRST
    maximum of every caption (text_global_pool's argmax, transformer.py:941-944), their positions, the cumulative offsets, the bucket order
    (stable by 32-row block count) and the count of out-of-vocabulary ids"""
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.input_pipeline import HostTextPlan
    from open_clip_amd.synth import synthetic_batch
    cfg = get_model_config("ViT-B-32")
    text = synthetic_batch(cfg, 37, seed=3)["text"]
    text[5, 3] = text[5].max()          # a tie: the first maximum wins
    text[7, 2] = 60000                  # an id outside the vocabulary (it is also the row's maximum)
    plan = HostTextPlan(text, cfg["text_cfg"]["vocab_size"], buckets=True)
    B, L = text.shape
    eot, toks, pos, off = [], [], [], [0]
    for b in range(B):
        row = text[b].tolist()
        e = row.index(max(row))
        eot.append(e)
        toks += row[:e + 1]
        pos += list(range(e + 1))
        off.append(off[-1] + e + 1)
    assert eot[5] == 3 and eot[7] == 2 and plan.n_bad == 1 and plan.M == off[-1]
    v = plan.device_views(plan.ints, plan.tokens)     # host tensors stand in for the device copies
    assert v["eot"].tolist() == eot and v["seq_off"].tolist() == off and v["last_row"].tolist() == [o - 1 for o in off[1:]]
    assert v["tokens"].tolist() == toks and v["posidx"].tolist() == pos
    nb = [(e + 1 + 31) // 32 for e in eot]
    order = v["order"].tolist()
    assert sorted(order) == list(range(B)) and [nb[i] for i in order] == sorted(nb)
    assert all(a < b for a, b in zip(order, order[1:]) if nb[a] == nb[b])           # stable inside a bucket
    assert v["counts"] == [nb.count(k) for k in range(1, (L + 31) // 32 + 1)] and sum(v["counts"]) == B
    assert plan.ints.numel() <= HostTextPlan.capacity(B, L)
    flat = HostTextPlan(text, None, buckets=False)
    assert flat.device_views(flat.ints, flat.tokens)["order"] is None and flat.counts is None and flat.n_bad == 0
