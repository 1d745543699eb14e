"""End-to-end parity on a real MI355X: the native model + loss (HIP kernels through the C ABI) against
  (1) the golden vectors the REFERENCE produced (tests/golden/*.npz) and
  (2) the CPU oracle (oracle/clip_oracle.py) on fresh seeded inputs.
Stated tolerance (bf16 operands / fp32 accumulation, fp32 residual stream, vs the fp32 CPU reference):
  features max-abs <= 4e-3 (unit-norm vectors), loss |d| <= 2e-2, logits max-abs <= 0.15 (scale up to 100),
  parameter-gradient rel-L2 <= 3.5e-2 for weight matrices / embeddings (measured <= 2.6e-2, profiles/r02_parity_report.txt), <= 5e-2 for
  1-D tensors (biases, LayerNorm affine: column sums with cancellation, measured <= 3.9e-2), <= 0.12 for tensors whose gradient norm
  is < 1e-3 of the largest.
Measured values are appended to gpurun_out/parity_report.txt."""
import numpy as np
import pytest
import torch

from open_clip_amd.configs import get_model_config
from open_clip_amd.synth import init_state_dict, synthetic_batch
from tests.golden_util import check_grad, grad_keys, load, state_from_golden
from tests.test_kernels_gpu import _report

pytestmark = pytest.mark.gpu

FEAT_TOL, LOSS_TOL, GRAD_TOL, GRAD_TOL_SMALL = 4e-3, 2e-2, 5e-2, 0.12
GRAD_TOL_MATRIX = 3.5e-2


def _grad_tol(ref_norm, gmax, ndim):
    if ref_norm < 1e-3 * gmax:
        return GRAD_TOL_SMALL
    return GRAD_TOL_MATRIX if ndim >= 2 else GRAD_TOL


def _build(cfg, state, siglip=False, **kw):
    from open_clip_amd.model import NativeCLIP
    extra = dict(init_logit_scale=float(np.log(10)), init_logit_bias=-10.0) if siglip else {}
    m = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=True, **extra, **kw)
    m.load_state_dict(state, strict=True)
    return m.cuda().train()


def _step(model, batch, siglip=False):
    from open_clip_amd.loss import NativeClipLoss, NativeSigLipLoss
    out = model(image=batch["image"].cuda(), text=batch["text"].cuda())
    loss_fn = NativeSigLipLoss() if siglip else NativeClipLoss()
    loss = loss_fn(**out)
    loss.backward()
    torch.cuda.synchronize()
    return out, loss


def _compare(tag, g, model, out, loss):
    fi = float((out["image_features"].float().cpu() - torch.from_numpy(g["out/image_features"])).abs().max())
    ft = float((out["text_features"].float().cpu() - torch.from_numpy(g["out/text_features"])).abs().max())
    dl = abs(float(loss) - float(g["out/loss"]))
    _report(f"{tag}: image_features max_abs={fi:.3e} text_features max_abs={ft:.3e} loss={float(loss):.6f} ref={float(g['out/loss']):.6f}")
    grads = {k: p.grad for k, p in model.named_parameters()}
    gmax = max(float(g["gnorm/" + k]) for k in grad_keys(g))
    worst = []
    for k in grad_keys(g):
        assert grads[k] is not None, f"no gradient for {k}"
        rel, nrel = check_grad(g, k, grads[k], 0)
        worst.append((rel, k, float(g["gnorm/" + k])))
    worst.sort(reverse=True)
    for rel, k, n in worst[:12]:
        _report(f"{tag}:   grad rel_l2={rel:.3e} |g|={n:.3e} {k}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL, (fi, ft)
    assert dl <= LOSS_TOL, dl
    for rel, k, n in worst:
        assert rel <= _grad_tol(n, gmax, grads[k].ndim), (k, rel, n)


@pytest.mark.parametrize("name,siglip", [("tiny_clip.npz", False), ("tiny_siglip.npz", True), ("tiny_quickgelu.npz", False), ("tiny_hd88.npz", False)])
def test_tiny_against_reference_golden(name, siglip):
    g = load(name)
    cfg = get_model_config("hd88-test" if "hd88" in name else "tiny-test")  # hd88: head_width 88 + mlp_ratio 4.3637 (ViT-g-14's shape class)
    # tiny_quickgelu: the reference's CLIP(quick_gelu=True) -- the `*-quickgelu` configs of the OpenAI / LAION-400M checkpoints
    model = _build(cfg, state_from_golden(g), siglip, **({"quick_gelu": True} if "quickgelu" in name else {}))
    batch = {"image": torch.from_numpy(g["image"].astype(np.float32)), "text": torch.from_numpy(g["text"])}
    out, loss = _step(model, batch, siglip)
    _compare(name, g, model, out, loss)


def test_vitb32_b8_against_reference_golden():
    g = load("vitb32_b8.npz")
    cfg = get_model_config("ViT-B-32")
    state = init_state_dict(cfg, seed=0, perturb=True)
    for k in ("visual.conv1.weight", "token_embedding.weight", "visual.proj"):
        if abs(float(state[k].double().sum()) - float(g["wsum/" + k])) > 1e-6 * state[k].numel() ** 0.5:
            pytest.skip("torch CPU RNG stream differs from the one that generated the fixture")
    batch = synthetic_batch(cfg, 8, seed=1234)
    model = _build(cfg, state)
    out, loss = _step(model, batch)
    _compare("vitb32_b8", g, model, out, loss)


@pytest.mark.parametrize("cfg_name,B,ckpt", [("small-test", 9, False), ("small-test", 16, True), ("small-test", 16, (1, 2))])
def test_against_cpu_oracle(cfg_name, B, ckpt):
    """fresh seeded inputs (odd batch, patch 16, 3 text heads, ragged EOT positions), oracle on the host CPU;
    also with block recompute (set_grad_checkpointing) and with recompute of all but the last 1 image / 2 text blocks (keep_last), which
    must not change anything"""
    from oracle import clip_oracle as O
    cfg = get_model_config(cfg_name)
    state = init_state_dict(cfg, seed=21, perturb=True)
    batch = synthetic_batch(cfg, B, seed=77)
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg)
    model = _build(cfg, state)
    if isinstance(ckpt, tuple):
        model.set_grad_checkpointing(True, keep_last=ckpt)
        ckpt = 2
    else:
        model.set_grad_checkpointing(ckpt)
    out, loss = _step(model, batch)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"oracle[{cfg_name},B{B},ckpt{int(ckpt)}]: feat max_abs {fi:.3e}/{ft:.3e} loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    for k, p in model.named_parameters():
        ref = grads[k]
        rel = float((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        tol = _grad_tol(float(ref.norm()), gmax, ref.ndim)
        if rel > tol * 0.5:
            _report(f"oracle[{cfg_name}]:   grad rel_l2={rel:.3e} |g|={float(ref.norm()):.3e} {k}")
        assert rel <= tol, (k, rel)


@pytest.mark.parametrize("name", ["ViT-S-16-alt", "ViT-M-32-alt", "ViT-B-16-plus-240", "ViT-B-32-plus-256", "ViT-L-14-336", "ViT-L-16-320",
                                  "ViT-H-14-378-quickgelu", "ViT-H-16", "ViT-g-14", "ViT-bigG-14", "ViT-e-14"])
def test_reference_config_shape_classes(name):
    """the registered reference configs at their own widths / head counts / head dims / patch and image sizes / MLP ratios (two blocks per
    tower instead of 10-56, batch 8 -- at batch 3 the contrastive loss sits at ln 3 with gradients that are small differences of large terms,
    and the bf16 noise of an unchanged absolute size reads as 5-60 % of them): features, loss and every gradient against the CPU oracle.  Covers the token counts 65 ... 730, widths
    384 ... 1792 (N = 896 / 640 / 384: ragged tiles of the 256-wide GEMM kernels), head_dim 64 / 80 / 88 / 104 / 112, MLP widths 6144 /
    8192 / 15360, text towers of 256 ... 1280."""
    from oracle import clip_oracle as O
    cfg = get_model_config(name)
    cfg["vision_cfg"]["layers"] = 2
    cfg["text_cfg"]["layers"] = 2
    B = 8
    state = init_state_dict(cfg, seed=5, perturb=True)
    batch = synthetic_batch(cfg, B, seed=91)
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg)
    model = _build(cfg, state, **({"quick_gelu": True} if cfg.get("quick_gelu") else {}))
    out, loss = _step(model, batch)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"config[{name}, 2+2 blocks, B{B}]: feat max_abs {fi:.3e}/{ft:.3e} loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    rows = []
    for k, p in model.named_parameters():
        ref = grads[k]
        rel = float((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        # the bias-like gradients of the pooled text rows (ln_final, the last block's c_proj.bias) are sums of 8 rows that nearly cancel
        # (the text-feature gradients of a contrastive batch sum to ~0): below 1e-2 of the largest gradient norm the tolerance is 0.2
        tol = 0.2 if float(ref.norm()) < 1e-2 * gmax else _grad_tol(float(ref.norm()), gmax, ref.ndim)
        rows.append((rel / tol, rel, float(ref.norm()), k))
    rows.sort(reverse=True)
    for frac, rel, n, k in rows[:4]:
        _report(f"config[{name}]:   grad rel_l2={rel:.3e} ({frac:.2f} of its tolerance) |g|={n:.3e} (largest |g| {gmax:.2e}) {k}")
    assert rows[0][0] <= 1.0, rows[:4]


def test_full_size_properties_vitb32():
    """BASELINE size-independent checks at a bench-like batch: finite outputs, unit-norm features, loss near ln(B)
    at init, every parameter receives a finite gradient, and a repeated step is bit-identical in the forward."""
    cfg = get_model_config("ViT-B-32")
    model = _build(cfg, init_state_dict(cfg, seed=0))
    B = 256
    batch = synthetic_batch(cfg, B, seed=5)
    out, loss = _step(model, batch)
    for k in ("image_features", "text_features"):
        f = out[k].float()
        assert torch.isfinite(f).all()
        assert float((f.norm(dim=-1) - 1).abs().max()) < 1e-3
    assert abs(float(loss) - np.log(B)) < 1.0
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    with torch.no_grad():
        o2 = model(image=batch["image"].cuda(), text=batch["text"].cuda())
    assert torch.equal(o2["image_features"], out["image_features"]) and torch.equal(o2["text_features"], out["text_features"])


def test_gradient_accumulation_path_matches_full_batch():
    """8f-3: the reference's --accum-freq loop (train.py:236-311) over the native modules: features of every micro-batch
    are cached under no_grad, then each micro-batch is re-run with gradient and the loss is taken over the concatenation
    (cached features as negatives).  Tower gradients must equal the full-batch gradients; logit_scale receives one full
    gradient PER micro-batch pass (the reference's behaviour, reproduced -- not normalised)."""
    from open_clip_amd.loss import NativeClipLoss
    from oracle import clip_oracle as O
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=31, perturb=True)
    accum, Bm = 3, 4
    batch = synthetic_batch(cfg, accum * Bm, seed=91)
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg)
    model = _build(cfg, state)
    loss_fn = NativeClipLoss()
    micro = [{"image": batch["image"][j * Bm:(j + 1) * Bm].cuda(), "text": batch["text"][j * Bm:(j + 1) * Bm].cuda()} for j in range(accum)]
    feats = {"image_features": [], "text_features": []}
    with torch.no_grad():
        for mb in micro:
            o = model(**mb)
            for k in feats:
                feats[k].append(o[k])
    model.zero_grad(set_to_none=True)
    for j, mb in enumerate(micro):
        o = model(**mb)
        inputs = {k: torch.cat(feats[k][:j] + [o[k]] + feats[k][j + 1:]) for k in feats}
        losses = loss_fn(**inputs, logit_scale=o["logit_scale"], output_dict=True)
        total = sum(v for k, v in losses.items() if k.endswith("_loss"))
        total.backward()
    torch.cuda.synchronize()
    assert abs(float(total.detach()) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    for k, p in model.named_parameters():
        ref = grads[k] * (accum if k == "logit_scale" else 1)
        rel = float((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        tol = GRAD_TOL if float(grads[k].norm()) >= 1e-3 * gmax else GRAD_TOL_SMALL
        assert rel <= tol, (k, rel)


def test_uint8_image_input_equals_normalised_float_input():
    """8f-4 at the model boundary: NativeCLIP.encode_image(uint8 pixels) == encode_image(ToTensor+Normalize(pixels))."""
    cfg = get_model_config("small-test")
    model = _build(cfg, init_state_dict(cfg, seed=4, perturb=True))
    S = cfg["vision_cfg"]["image_size"]
    g = torch.Generator().manual_seed(8)
    u8 = torch.randint(0, 256, (5, S, S, 3), generator=g, dtype=torch.uint8)  # decoder layout [B,H,W,3]
    mean = torch.tensor(model.visual.image_mean).view(1, 3, 1, 1)
    std = torch.tensor(model.visual.image_std).view(1, 3, 1, 1)
    f = (u8.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
    with torch.no_grad():
        a = model.encode_image(u8.cuda(), normalize=True)
        b = model.encode_image(u8.permute(0, 3, 1, 2).contiguous().cuda(), normalize=True)
        c = model.encode_image(f.cuda(), normalize=True)
    assert torch.equal(a, b)
    assert float((a - c).abs().max()) <= 2e-3


def test_vitl14_grad_checkpointed_step_against_cpu_oracle():
    """BASELINE config 4's model (ViT-L-14: patch 14 -> K padded 588 -> 640, 257 image tokens -> the 9-wave attention
    kernels, 24 + 12 blocks, embed 768) with set_grad_checkpointing, small batch, against the CPU oracle."""
    from oracle import clip_oracle as O
    cfg = get_model_config("ViT-L-14")
    state = init_state_dict(cfg, seed=2, perturb=True)
    batch = synthetic_batch(cfg, 3, seed=13)
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg)
    model = _build(cfg, state)
    model.set_grad_checkpointing(True)
    out, loss = _step(model, batch)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"oracle[ViT-L-14,B3,ckpt]: feat max_abs {fi:.3e}/{ft:.3e} loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    keys = ["visual.conv1.weight", "visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.23.mlp.c_fc.weight",
            "visual.transformer.resblocks.11.ln_1.weight", "transformer.resblocks.0.mlp.c_proj.weight", "token_embedding.weight",
            "text_projection", "visual.proj", "visual.positional_embedding", "logit_scale"]
    named = dict(model.named_parameters())
    for k in keys:
        ref = grads[k]
        rel = float((named[k].grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        tol = _grad_tol(float(ref.norm()), gmax, ref.ndim)
        _report(f"oracle[ViT-L-14]:   grad rel_l2={rel:.3e} |g|={float(ref.norm()):.3e} {k}")
        assert rel <= tol, (k, rel)


def test_state_carried_between_steps_is_not_stale():
    """The per-step caches (bf16 weight copies, the bf16 hand-off of the residual-stream gradient between blocks, allocator-
    recycled addresses) must not leak from one step into the next: gradients of a second step on NEW data equal those of a
    fresh model given only that data (fp32 atomics order aside)."""
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=17, perturb=True)
    a, b = synthetic_batch(cfg, 8, seed=1), synthetic_batch(cfg, 8, seed=2)
    m1 = _build(cfg, state)
    _step(m1, a)
    m1.zero_grad(set_to_none=True)
    _, l1 = _step(m1, b)
    m2 = _build(cfg, state)
    stats = [m2._cache.twin_stats, m2.visual._cache.twin_stats]
    for st in stats:
        st.update(hit=0, miss=0)
    _, l2 = _step(m2, b)
    # the bf16 twin of the residual-stream gradient rides on the gradient tensor from block to block: every block backward of both
    # towers (2 + 2 here) must have found it (a miss is only a cast kernel, but then the optimisation would be silently gone)
    assert sum(st["hit"] for st in stats) >= 4 and sum(st["miss"] for st in stats) == 0, stats
    assert abs(float(l1.detach()) - float(l2.detach())) < 1e-6
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        rel = float((p.grad - q.grad).norm() / q.grad.norm().clamp_min(1e-30))
        assert rel < 1e-4, (k, rel)


@pytest.mark.parametrize("siglip", [True, False])
def test_head_dim_80_tower_against_cpu_oracle(siglip):
    """BASELINE config 5's shape class (ViT-H-14 + SigLIP: image head_dim 80, patch 14, sigmoid pairwise loss with logit_bias) in
    miniature: the image tower runs the generic attention kernels, the text tower the specialised ones."""
    from oracle import clip_oracle as O
    cfg = get_model_config("hd80-test")
    state = init_state_dict(cfg, seed=9, perturb=True, siglip=siglip)
    batch = synthetic_batch(cfg, 6, seed=23)
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg, siglip=siglip)
    model = _build(cfg, state, siglip)
    out, loss = _step(model, batch, siglip)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"oracle[hd80-test,siglip{int(siglip)}]: feat max_abs {fi:.3e}/{ft:.3e} loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    for k, p in model.named_parameters():
        ref = grads[k]
        rel = float((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        tol = GRAD_TOL if float(ref.norm()) >= 1e-3 * gmax else GRAD_TOL_SMALL
        assert rel <= tol, (k, rel)


def test_vith14_siglip_full_size_step_runs():
    """BASELINE config 5 at full model size (ViT-H-14: 32 image blocks of width 1280 / head_dim 80 / 257 tokens, 24 text blocks of
    width 1024; SigLIP loss): one grad-checkpointed step at a small batch -- finite loss, unit-norm features, a finite gradient
    on every parameter.  (Numerical parity of this shape class: test_head_dim_80_tower_against_cpu_oracle.)"""
    cfg = get_model_config("ViT-H-14")
    model = _build(cfg, init_state_dict(cfg, seed=0, siglip=True), siglip=True)
    model.set_grad_checkpointing(True)
    batch = synthetic_batch(cfg, 4, seed=3)
    out, loss = _step(model, batch, siglip=True)
    assert torch.isfinite(loss.detach())
    for k in ("image_features", "text_features"):
        f = out[k].float()
        assert torch.isfinite(f).all() and float((f.norm(dim=-1) - 1).abs().max()) < 1e-3
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_training_steps_with_fused_optimizer_track_the_cpu_oracle():
    """Four full training steps -- forward, ClipLoss, backward, fused NativeAdamW (clip + update + in-place refresh of the cached
    bf16 operand copies), logit_scale clamp -- against the CPU oracle stepped with torch.optim.AdamW + clip_grad_norm_ on the
    same data: the loss of every step and the final parameters must agree (a stale operand copy or a missed version bump would
    show from the second step on)."""
    import math
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    from oracle import clip_oracle as O
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=41, perturb=True)
    batches = [synthetic_batch(cfg, 8, seed=100 + i) for i in range(4)]
    lr, wd, clip = 2e-3, 0.2, 1.0
    model = _build(cfg, state)
    opt = NativeAdamW(param_groups_like_reference(model, wd), lr=lr, betas=(0.9, 0.98), eps=1e-6, grad_clip_norm=clip,
                      weight_caches=weight_caches_of(model))
    loss_fn = NativeClipLoss()
    ref_params = {k: torch.nn.Parameter(v.clone().float()) for k, v in state.items()}
    skip = {"positional_embedding", "visual.positional_embedding", "visual.class_embedding"}
    ref_opt = torch.optim.AdamW([{"params": [p for k, p in ref_params.items() if p.ndim <= 1 or k in skip], "weight_decay": 0.0},
                                 {"params": [p for k, p in ref_params.items() if not (p.ndim <= 1 or k in skip)], "weight_decay": wd}],
                                lr=lr, betas=(0.9, 0.98), eps=1e-6)
    for i, b in enumerate(batches):
        opt.zero_grad(set_to_none=True)
        out = model(image=b["image"].cuda(), text=b["text"].cuda())
        loss = loss_fn(**out)
        loss.backward()
        opt.step()
        with torch.no_grad():
            model.logit_scale.clamp_(0, math.log(100))
        outs, grads = O.train_forward_backward(b["image"], b["text"], {k: p.detach() for k, p in ref_params.items()}, cfg)
        for k, p in ref_params.items():
            p.grad = grads[k]
        torch.nn.utils.clip_grad_norm_(list(ref_params.values()), clip)
        ref_opt.step()
        with torch.no_grad():
            ref_params["logit_scale"].clamp_(0, math.log(100))
        _report(f"train-steps[{i}]: loss {float(loss.detach()):.5f} oracle {float(outs['loss']):.5f}")
        assert abs(float(loss.detach()) - float(outs["loss"])) <= LOSS_TOL, i
    torch.cuda.synchronize()
    worst_any, worst_mat = (0.0, ""), (0.0, "")
    for k, p in model.named_parameters():
        ref, got, start = ref_params[k].detach(), p.detach().float().cpu(), state[k].float()
        if k.endswith("attn.in_proj_bias"):  # drop the K third: its exact gradient is zero, Adam moves it by rounding noise alone
            c = ref.numel() // 3
            keep = torch.cat([torch.arange(0, c), torch.arange(2 * c, 3 * c)])
            ref, got, start = ref[keep], got[keep], start[keep]
        moved = (ref - start).norm()
        err = (got - ref).norm()
        # Adam turns a gradient into a step of ~lr per element whatever its size, so elements whose gradient is (nearly) rounding
        # noise -- the K third of in_proj_bias (softmax is invariant to it: exact gradient zero), embedding rows of rare tokens --
        # move differently in any two implementations; the bound only has to catch a wrong update (error ~ the movement itself)
        bound = 0.20  # measured worst: 0.11 (token_embedding.weight: rows of rare tokens)
        assert float(err) <= bound * float(moved) + 1e-6, (k, float(err), float(moved))
        ratio = float(err) / (float(moved) + 1e-30)
        worst_any = max(worst_any, (ratio, k))
        if ref.dim() >= 2 and ref.numel() >= 4096 and "embedding" not in k:
            worst_mat = max(worst_mat, (ratio, k))
    _report(f"train-steps: error / movement of the parameters after the steps: worst {worst_any[0]:.3f} ({worst_any[1]}), worst weight matrix {worst_mat[0]:.3f} "
            f"({worst_mat[1]}); bounds 0.20 / 0.10")
    assert worst_mat[0] <= 0.10, worst_mat  # weight matrices (every element has a real gradient; measured worst 0.051): tighter than the noisy 1-D tensors


def test_reference_train_loop_under_autocast_tracks_the_cpu_oracle():
    """The step exactly as the reference's loop runs it on a GPU (the reference itself is not on the GPU box, so its few lines of
    glue are restated here with their sources): batches arrive as pinned HOST tensors and go through ``prepare_batch``
    (base_task.py:135-157: ``.to(device, non_blocking=True)``, floats keep their dtype under amp), the forward runs inside
    ``torch.amp.autocast('cuda', dtype=torch.bfloat16)`` (precision.py:6-17, train.py:228-231), ``training_forward`` sums the
    ``*_loss`` entries (clip_task.py:41-46), then ``backward`` OUTSIDE the autocast region, ``torch.nn.utils.clip_grad_norm_`` on
    ``.grad`` (train.py:205-214), ``optimizer.step()`` and ``clamp_logit_scale`` (image_text_task.py:91-101).  Three steps against
    the CPU oracle stepped with torch.optim.AdamW: per-step losses and the final parameters."""
    import math
    from functools import partial
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    from oracle import clip_oracle as O
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=51, perturb=True)
    device = torch.device("cuda:0")
    lr, wd, clip = 2e-3, 0.2, 1.0
    model = _build(cfg, state)
    loss_mod = NativeClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=0, world_size=1)  # create_task's arguments
    optimizer = NativeAdamW(param_groups_like_reference(model, wd), lr=lr, betas=(0.9, 0.98), eps=1e-6, weight_caches=weight_caches_of(model))
    autocast = partial(torch.amp.autocast, device_type="cuda", dtype=torch.bfloat16)

    def prepare_batch(batch, input_dtype=None):  # base_task.py:135-157
        return {k: (v.to(device=device, dtype=input_dtype, non_blocking=True) if v.is_floating_point() else v.to(device=device, non_blocking=True))
                for k, v in batch.items()}

    def training_forward(batch):  # clip_task.py:41-46
        model_out = model(image=batch["image"], text=batch["text"])
        losses = loss_mod(**model_out, output_dict=True)
        losses["loss"] = sum(v for k, v in losses.items() if k.endswith("_loss"))
        return losses, {k: model_out[k] for k in ("logit_scale", "logit_bias") if k in model_out}

    ref_params = {k: torch.nn.Parameter(v.clone().float()) for k, v in state.items()}
    skip = {"positional_embedding", "visual.positional_embedding", "visual.class_embedding"}
    ref_opt = torch.optim.AdamW([{"params": [p for k, p in ref_params.items() if p.ndim <= 1 or k in skip], "weight_decay": 0.0},
                                 {"params": [p for k, p in ref_params.items() if not (p.ndim <= 1 or k in skip)], "weight_decay": wd}],
                                lr=lr, betas=(0.9, 0.98), eps=1e-6)
    for i in range(3):
        host = {k: v.pin_memory() for k, v in synthetic_batch(cfg, 8, seed=300 + i).items()}
        batch = prepare_batch(host)
        optimizer.zero_grad()
        with autocast():
            losses, report = training_forward(batch)
            total_loss = losses["loss"]
        total_loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip, norm_type=2.0)
        optimizer.step()
        with torch.no_grad():
            model.logit_scale.clamp_(0, math.log(100))
        assert total_loss.dtype == torch.float32 and report["logit_scale"].dtype == torch.float32
        outs, grads = O.train_forward_backward(host["image"], host["text"], {k: p.detach() for k, p in ref_params.items()}, cfg)
        for k, p in ref_params.items():
            p.grad = grads[k]
        torch.nn.utils.clip_grad_norm_(list(ref_params.values()), clip)
        ref_opt.step()
        with torch.no_grad():
            ref_params["logit_scale"].clamp_(0, math.log(100))
        _report(f"autocast-loop[{i}]: loss {float(total_loss.detach()):.5f} oracle {float(outs['loss']):.5f}")
        assert abs(float(total_loss.detach()) - float(outs["loss"])) <= LOSS_TOL, i
    torch.cuda.synchronize()
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
    _report(f"autocast-loop: error / movement of the parameters after the steps: worst {worst[0]:.3f} ({worst[1]}); bound 0.25")


def test_vith14_siglip_full_size_against_cpu_oracle_and_loss_after_one_update():
    """BASELINE config 5 at FULL model size (ViT-H-14 + SigLIP loss, block recompute), B = 2, against the CPU oracle: loss, features and
    a spread of gradients.  Then ONE AdamW update at the bench's learning rate on both sides and the loss again: round 1's sanity
    bench line ended at final_loss 178.8 after three updates (init ~10).  If the native loss after the update agrees with the
    oracle's, that excursion is the optimizer's (lr 5e-4 from step 0 on a 1 B-parameter model, where the reference warms up over
    10 000 steps, params.py:288) and not the kernels'."""
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    from oracle import clip_oracle as O
    cfg = get_model_config("ViT-H-14")
    state = init_state_dict(cfg, seed=5, siglip=True)
    batch = synthetic_batch(cfg, 2, seed=29)
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg, siglip=True)
    model = _build(cfg, state, siglip=True)
    model.set_grad_checkpointing(True)
    opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_caches=weight_caches_of(model))
    out, loss = _step(model, batch, siglip=True)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"oracle[ViT-H-14 SigLIP,B2,ckpt]: feat max_abs {fi:.3e}/{ft:.3e} loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    keys = ["visual.conv1.weight", "visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.31.mlp.c_fc.weight",
            "visual.transformer.resblocks.15.attn.out_proj.weight", "transformer.resblocks.0.mlp.c_proj.weight", "transformer.resblocks.23.attn.in_proj_weight",
            "text_projection", "visual.proj", "visual.positional_embedding", "logit_scale", "logit_bias"]
    named = dict(model.named_parameters())
    for k in keys:
        ref = grads[k]
        rel = float((named[k].grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        tol = _grad_tol(float(ref.norm()), gmax, ref.ndim)
        _report(f"oracle[ViT-H-14]:   grad rel_l2={rel:.3e} |g|={float(ref.norm()):.3e} {k}")
        assert rel <= tol, (k, rel)
    # one update on both sides, then the loss on the same batch
    opt.step()
    ref_params = {k: torch.nn.Parameter(v.clone().float()) for k, v in state.items()}
    skip = {"positional_embedding", "visual.positional_embedding", "visual.class_embedding"}
    ref_opt = torch.optim.AdamW([{"params": [p for k, p in ref_params.items() if p.ndim <= 1 or k in skip], "weight_decay": 0.0},
                                 {"params": [p for k, p in ref_params.items() if not (p.ndim <= 1 or k in skip)], "weight_decay": 0.2}],
                                lr=5e-4, betas=(0.9, 0.98), eps=1e-6)
    for k, p in ref_params.items():
        p.grad = grads[k]
    ref_opt.step()
    with torch.no_grad():
        o2 = model(image=batch["image"].cuda(), text=batch["text"].cuda())
        from open_clip_amd.loss import NativeSigLipLoss
        l2 = float(NativeSigLipLoss()(**o2))
        ro = O.clip_forward(batch["image"].float(), batch["text"], {k: p.detach() for k, p in ref_params.items()}, cfg)
        r2 = float(O.siglip_loss(ro["image_features"], [ro["text_features"]], ro["logit_scale"], ro["logit_bias"], 0))
    _report(f"oracle[ViT-H-14]: loss after one AdamW update at lr 5e-4: native {l2:.4f} oracle {r2:.4f} (before: {float(outs['loss']):.4f})")
    assert abs(l2 - r2) <= max(5 * LOSS_TOL, 0.03 * abs(r2))


def test_packed_text_tower_equals_dense_text_tower():
    """``pack_text`` (only the tokens up to the pooled EOT exist in the text tower) against the same model running all 77 positions
    like the reference: identical text features BIT FOR BIT (every kernel works row- / sequence-locally and the causal mask hides
    the dropped positions), the same loss (its row sums are reduced with fp32 atomics, so two runs of the SAME tower already differ
    in the last bits: |diff| <= 2e-6 relative), and gradients equal up to the fp32 summation order of the weight-gradient GEMMs
    (the dense tower adds the dropped rows' exact zeros in a different split).  tolerance: rel-L2 <= 2e-5 per parameter."""
    cfg = get_model_config("ViT-B-32")
    state = init_state_dict(cfg, seed=1, perturb=True)
    batch = synthetic_batch(cfg, 96, seed=11)
    batch["text"][0, :] = 0
    batch["text"][0, 0] = cfg["text_cfg"]["vocab_size"] - 1          # EOT first: a one-token sequence
    batch["text"][1, -1] = cfg["text_cfg"]["vocab_size"] - 1 + 0     # a second maximum at the end: argmax keeps the FIRST one
    res = {}
    for mode in (True, False):
        model = _build(cfg, state)
        assert model.pack_text, "the packed text tower must be the default on the ViT-B-32 text tower (head_dim 64, L = 77)"
        model.pack_text = mode
        out, loss = _step(model, batch)
        res[mode] = (out, float(loss), {k: p.grad.clone() for k, p in model.named_parameters()})
    (o1, l1, g1), (o0, l0, g0) = res[True], res[False]
    assert torch.equal(o1["text_features"], o0["text_features"]), float((o1["text_features"] - o0["text_features"]).abs().max())
    assert torch.equal(o1["image_features"], o0["image_features"])
    assert abs(l1 - l0) <= 2e-6 * abs(l0), (l1, l0)
    worst = max((float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)), k) for k in g0)
    _report(f"packed vs dense text tower (ViT-B-32, B=96): features bit-identical, loss {l1:.7f} vs {l0:.7f}; worst gradient rel_l2 = {worst[0]:.3e} ({worst[1]})")
    assert worst[0] <= 2e-5, worst


def test_overlapped_towers_equal_one_stream():
    """``tower_streams`` (image tower on a stream of its own next to the text tower, forward and backward) against the same model on
    one stream: bit-identical features, and over four optimizer steps -- allocator reuse across the two streams, the bf16 weight
    copies the optimizer rewrites, the packed-text plan -- the same loss trajectory (fp32 atomics reorder the weight-gradient sums:
    rel <= 2e-5 on the first step's gradients, 2e-4 on the loss of the later steps) and final parameters (rel <= 3e-3: Adam turns the rounding noise of gradients that are
    mathematically zero, e.g. the key bias of every attention layer, into +-lr steps)."""
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    cfg = get_model_config("ViT-B-32")
    state = init_state_dict(cfg, seed=2, perturb=True)
    batches = [synthetic_batch(cfg, 128, seed=20 + i, device="cuda") for i in range(4)]
    res = {}
    for mode in (True, "serial", False):
        model = _build(cfg, state)
        assert model.tower_streams is True, "overlapped towers must be the default"
        model.tower_streams = mode
        opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_caches=weight_caches_of(model))
        loss_fn, losses, first = NativeClipLoss(), [], None
        for b in batches:
            opt.zero_grad(set_to_none=True)
            out = model(image=b["image"], text=b["text"])
            loss = loss_fn(**out)
            loss.backward()
            if first is None:
                first = (out["image_features"].clone(), out["text_features"].clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
            opt.step()
            losses.append(float(loss.detach()))
            del out, loss  # the next step reuses this step's memory while the other stream may still be draining
        torch.cuda.synchronize()
        res[mode] = (first, losses, {k: p.detach().clone() for k, p in model.named_parameters()})
    f0, l0, p0 = res[False]
    for mode, what in ((True, "overlapped"), ("serial", "one-at-a-time on two streams (bench.py's event-timed steps)")):
        f1, l1, p1 = res[mode]
        assert torch.equal(f1[0], f0[0]) and torch.equal(f1[1], f0[1])
        gw = max((float((f1[2][k] - f0[2][k]).norm() / (f0[2][k].norm() + 1e-30)), k) for k in f0[2])
        lw = max(abs(a - b) / abs(b) for a, b in zip(l1, l0))
        pw = max((float((p1[k] - p0[k]).norm() / (p0[k].norm() + 1e-30)), k) for k in p0)
        _report(f"{what} vs one-stream towers (ViT-B-32, B=128, 4 AdamW steps): features bit-identical; worst gradient rel_l2 {gw[0]:.2e} ({gw[1]}), "
                f"loss trajectory rel {lw:.2e} {['%.5f' % v for v in l1]}, final parameters rel_l2 {pw[0]:.2e} ({pw[1]})")
        assert gw[0] <= 2e-5 and lw <= 2e-4 and pw[0] <= 3e-3, (mode, gw, lw, pw)


def test_token_ids_outside_the_vocabulary_raise_like_nn_embedding():
    """the reference's ``nn.Embedding`` (model.py:399) raises on an id outside [0, vocab_size); the native embedding kernels clamp (no
    read outside the table) and the packed-text plan counts such ids on the device, so the forward raises IndexError from the 8-byte
    read-back it waits for anyway (ocn_token_range_check); a clean batch passes, and the kernel's count is exact"""
    from open_clip_amd import ops
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=3, perturb=True)
    model = _build(cfg, state)
    batch = synthetic_batch(cfg, 6, seed=5, device="cuda")
    model(image=batch["image"], text=batch["text"])  # clean
    V = cfg["text_cfg"]["vocab_size"]
    assert int(ops.token_range_check(batch["text"], V)) == 0
    bad = batch["text"].clone()
    bad[1, 2], bad[4, 1], bad[5, 3] = V, -1, V + 12345
    assert int(ops.token_range_check(bad, V)) == 3
    with pytest.raises(IndexError, match="3 token id"):
        model(image=batch["image"], text=bad)
    with pytest.raises(IndexError):
        model.encode_text(bad, normalize=True)


@pytest.mark.parametrize("single_query", [True, False], ids=["single_query", "all_queries"])
@pytest.mark.parametrize("variant", ["packed", "dense_text", "recompute", "siglip"])
def test_pooled_last_block_equals_full_block(variant, single_query):
    """the last block of each tower evaluated only on the pooled rows behind its attention (model.py::_PooledBlockFn) against the full
    block: the dropped rows reach neither the features nor any gradient.  With every query through the attention kernel (round 3's form,
    ``pooled_single_query = False``) features, loss and every gradient agree up to fp32 summation order (the weight gradients of the last block
    sum over B rows instead of M, the rest of the graph is the same).  The single-query form (round 4: K, V projection only, one query per
    sequence through csrc/attention_pooled.hip) evaluates the SAME attention row in fp32 where the tile kernel rounds P to bf16 in front of its
    P.V product, so the block's bf16 attention output differs by a rounding here and there: agreement at the level of one bf16 rounding of
    that tensor -- an order of magnitude inside the oracle tolerances -- instead of bit for bit."""
    cfg = get_model_config("ViT-B-32")
    siglip = variant == "siglip"
    state = init_state_dict(cfg, seed=9, perturb=True, siglip=siglip)
    batch = synthetic_batch(cfg, 24, seed=31, device="cuda")
    res = {}
    for pooled in (True, False):
        model = _build(cfg, state, siglip=siglip)
        assert model.pooled_last_block and model.visual.pooled_last_block and model.pooled_single_query, "the pooled single-query last block must be the default"
        model.pooled_last_block = model.visual.pooled_last_block = pooled
        model.pooled_single_query = single_query
        if variant == "dense_text":
            model.pack_text = False
        if variant == "recompute":
            model.set_grad_checkpointing(True)
        out, loss = _step(model, batch, siglip=siglip)
        res[pooled] = (out, float(loss), {k: p.grad.clone() for k, p in model.named_parameters()})
    (o1, l1, g1), (o0, l0, g0) = res[True], res[False]
    fi = float((o1["image_features"] - o0["image_features"]).abs().max())
    ft = float((o1["text_features"] - o0["text_features"]).abs().max())
    worst = max((float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)), k) for k in g0)
    _report(f"pooled vs full last block [{variant}, {'single query' if single_query else 'all queries'}] (ViT-B-32, B=24): features max |diff| {fi:.1e} / {ft:.1e}, "
            f"loss {l1:.7f} vs {l0:.7f}; worst gradient rel_l2 = {worst[0]:.3e} ({worst[1]})")
    if single_query:
        # The attention row differs by one bf16 rounding here and there, and ANY perturbation of that size at the top of the towers moves the
        # gradients below it by what the oracle tolerances allow for: the 1-D gradients are column sums over rows that nearly cancel (a contrastive
        # batch at initialisation), the small-norm tensors likewise -- measured 2e-2 on visual.positional_embedding and 3e-2 on a LayerNorm bias ten
        # blocks below from this ONE change.  So the comparison uses the tolerance classes of the oracle tests; the kernel itself is held to 4e-3 /
        # 1e-2 against fp32 torch in tests/test_kernels_gpu.py::test_attention_pooled_single_query.
        gmax = max(float(v.norm()) for v in g0.values())
        fr = max((float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)) / _grad_tol(float(g0[k].norm()), gmax, g0[k].ndim), k) for k in g0)
        _report(f"pooled vs full last block [{variant}, single query]: worst gradient = {fr[0]:.2f} of its oracle tolerance class ({fr[1]})")
        assert fi <= 4e-4 and ft <= 4e-4 and abs(l1 - l0) <= 2e-3 and fr[0] <= 1.0, (fi, ft, l1, l0, fr)
    else:
        assert fi <= 1e-6 and ft <= 1e-6 and abs(l1 - l0) <= 2e-6 * abs(l0) and worst[0] <= 2e-5, (fi, ft, l1, l0, worst)


@pytest.mark.parametrize("cfg_name,B,siglip", [("ViT-B-32", 48, False), ("small-test", 24, True)])
def test_deterministic_step_is_bit_reproducible(cfg_name, B, siglip):
    """``NativeCLIP(deterministic=True)`` + ``NativeClipLoss(deterministic=True)`` -- the native counterpart of ``torch.use_deterministic_algorithms``
    (which the reference's golden vectors were generated under, oracle/make_golden.py): no fp32 atomic is left in any sum of the step (weight / bias
    gradient GEMMs through per-split slabs, LayerNorm dgamma / dbeta through per-workgroup slabs, token-embedding rows by whole runs of a stable sort,
    positional / class-embedding gradients by single writers, loss and d/d logit_scale by one slot per row), so two runs of the same step on the same
    inputs give the same BITS for the loss and for every one of the gradients -- with the towers on two streams, packed text rows and pooled last blocks
    as shipped.  The default (atomic) form of the same step agrees with it to summation-order noise."""
    from open_clip_amd.loss import NativeClipLoss, NativeSigLipLoss
    cfg = get_model_config(cfg_name)
    state = init_state_dict(cfg, seed=4, perturb=True, siglip=siglip)
    batch = synthetic_batch(cfg, B, seed=12, device="cuda")

    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    clipped = {}

    def run(det):
        model = _build(cfg, state, siglip=siglip, deterministic=det)
        out = model(image=batch["image"], text=batch["text"])
        loss = (NativeSigLipLoss(deterministic=det) if siglip else NativeClipLoss(deterministic=det))(**out)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        # round 5: the clipped update too -- clip_grad_norm_'s squared-norm sum in a fixed order (NativeAdamW(deterministic=True): per-chunk partials
        # folded by one workgroup, ocn_sumsq_multi's chunk workspace) was the one fp32-atomic sum left under deterministic=True
        opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, grad_clip_norm=0.05,
                          weight_caches=weight_caches_of(model), deterministic=det)
        opt.step()
        torch.cuda.synchronize()
        clipped[len(clipped)] = (opt.last_grad_norm_sq.detach().clone(), {k: p.detach().clone() for k, p in model.named_parameters()})
        return loss.detach().clone(), grads

    l1, g1 = run(True)
    l2, g2 = run(True)
    assert torch.equal(l1, l2), (float(l1), float(l2))
    diff = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    assert not diff, f"{len(diff)} gradients differ between two deterministic runs: {diff[:6]}"
    assert torch.equal(clipped[0][0], clipped[1][0]), "squared gradient norm differs between two deterministic runs"
    pdiff = [k for k in clipped[0][1] if not torch.equal(clipped[0][1][k], clipped[1][1][k])]
    assert not pdiff, f"{len(pdiff)} parameters differ after the clipped AdamW update of two deterministic runs: {pdiff[:6]}"
    want = sum(float(v.double().pow(2).sum()) for v in g1.values())
    assert abs(float(clipped[0][0]) / want - 1) <= 1e-5 and float(clipped[0][0]) ** 0.5 > 0.05  # the norm is right, and the clip was active
    l0, g0 = run(False)
    worst = max((float((g0[k] - g1[k]).norm() / (g1[k].norm() + 1e-30)), k) for k in g1)
    _report(f"deterministic step [{cfg_name}, B={B}{', siglip' if siglip else ''}]: two runs bit-identical (loss + {len(g1)} gradients); atomic form differs by "
            f"rel_l2 <= {worst[0]:.2e} ({worst[1]}), loss {float(l0):.7f} vs {float(l1):.7f}")
    assert worst[0] <= 5e-3 and abs(float(l0) - float(l1)) <= 1e-4


@pytest.mark.parametrize("siglip", [False, True])
def test_get_logits_is_differentiable_like_the_reference(siglip):
    """``CLIP.get_logits`` (model.py:413-420) is differentiable in the reference; the native one goes through an autograd node over the
    library's GEMMs (model.py::_LogitsFn).  A caller's own loss on these logits -- here the reference's ClipLoss / SigLipLoss arithmetic written in torch,
    loss.py:103-141 / :356-367 -- must give the gradients the native loss modules give (same products, same bf16 operands; G rounded to bf16 in both)."""
    import torch.nn.functional as F
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=33, perturb=True, siglip=siglip)
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, 24, seed=34).items()}
    model = _build(cfg, state, siglip=siglip)
    li, lt = model.get_logits(batch["image"], batch["text"])
    assert li.requires_grad and lt.shape == li.T.shape
    if siglip:
        labels = 2 * torch.eye(li.shape[0], device=li.device) - 1
        loss = -F.logsigmoid(labels * li).sum() / li.shape[0]
    else:
        labels = torch.arange(li.shape[0], device=li.device)
        loss = (F.cross_entropy(li, labels) + F.cross_entropy(lt, labels)) / 2
    loss.backward()
    ref = _build(cfg, state, siglip=siglip)
    _, ref_loss = _step(ref, batch, siglip=siglip)
    assert abs(float(loss) - float(ref_loss)) <= 2e-3 * max(1.0, abs(float(ref_loss)))
    worst = max((float((p.grad - q.grad).norm() / q.grad.norm().clamp_min(1e-30)), k) for (k, p), q in zip(model.named_parameters(), ref.parameters()))
    _report(f"get_logits[{'siglip' if siglip else 'clip'}] + torch loss vs native loss module: loss {float(loss):.6f} vs {float(ref_loss):.6f}, worst gradient rel_l2 {worst[0]:.3e} ({worst[1]})")
    assert worst[0] <= 2e-2, worst
    with torch.no_grad():
        l2, _ = model.get_logits(batch["image"], batch["text"])
    assert not l2.requires_grad and torch.equal(l2, li.detach())
