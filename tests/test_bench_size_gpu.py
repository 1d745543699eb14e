"""Parity AT THE BENCH'S OWN SIZES for the kernels that are not GEMMs, on a real MI355X through the C ABI:
LayerNorm forward / backward at [204 800 x 768] (image tower, local batch 4096 x 50 tokens) and [177 803 x 512] (packed text tower:
4096 captions with EOT position ~ U[8, 76]), attention forward / backward at B = 4096 (12 heads x 50 tokens dense; 8 heads x 77
tokens packed, causal, un-bucketed and in the three buckets the step launches), each against an fp32 torch statement of the same op
evaluated in chunks; every output buffer is pre-filled with NaN so that a row a launch never wrote shows.  Then one whole ViT-B-32
training step at batch 512 against the CPU oracle (the largest batch the oracle finishes in about a minute of host time), and the
evaluation-mode calls the reference makes after every epoch.  Tolerances as in tests/test_kernels_gpu.py / tests/test_model_gpu.py."""
import numpy as np
import pytest
import torch

from open_clip_amd.configs import get_model_config
from open_clip_amd.synth import init_state_dict, synthetic_batch
from tests.test_kernels_gpu import _attn_ref, _report, bf, check, rel_l2

pytestmark = pytest.mark.gpu

B_BENCH = 4096
MI = B_BENCH * 50


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture()
def nan_outputs(monkeypatch):
    """every buffer the ops layer allocates for a kernel's output arrives full of NaN (floating point) / -1 (integers)"""
    from open_clip_amd import ops

    def empty(shape, dtype, like):
        fill = float("nan") if dtype.is_floating_point else -1
        return torch.full(tuple(shape), fill, dtype=dtype, device=like.device)

    monkeypatch.setattr(ops, "empty", empty)


def _text_lengths(B, L=77, seed=1234):
    """caption lengths with the bench's distribution: EOT position ~ U[8, 76] (open_clip_amd/synth.py) -> eot + 1 tokens kept
    (about 177.8 k of the 315 392 rows at B = 4096)"""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(8, L, (B,), generator=g) + 1


@pytest.mark.parametrize("M,C,tag", [(MI, 768, "image tower"), (None, 512, "packed text tower")])
def test_layernorm_at_bench_size(dev, nan_outputs, M, C, tag):
    from open_clip_amd import ops
    if M is None:
        M = int(_text_lengths(B_BENCH).sum())  # ~177.8 k rows
    g = torch.Generator(device=dev).manual_seed(M + C)
    x = torch.randn(M, C, device=dev, generator=g) * 3 + torch.randn(M, 1, device=dev, generator=g)  # rows with their own mean / scale
    w = 1 + 0.1 * torch.randn(C, device=dev, generator=g)
    b = 0.1 * torch.randn(C, device=dev, generator=g)
    dy = bf(torch.randn(M, C, device=dev, generator=g))
    dres = torch.randn(M, C, device=dev, generator=g)
    y16, y32, mean, rstd = ops.layernorm_fwd(x, w, b, want_bf16=True, want_f32=True)
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx32, dx16 = ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres, want_f32=True, want_bf16=True)
    torch.cuda.synchronize()
    for t in (y16, y32, mean, rstd, dx32, dx16):
        assert torch.isfinite(t.float()).all(), f"layernorm[{M}x{C}]: rows never written"
    ref_dw, ref_db = torch.zeros(C, device=dev, dtype=torch.float64), torch.zeros(C, device=dev, dtype=torch.float64)
    worst = {"y32": 0.0, "y16": 0.0, "dx32": 0.0, "dx16": 0.0, "mean": 0.0, "rstd": 0.0}
    CHUNK = 16384
    for r0 in range(0, M, CHUNK):
        sl = slice(r0, min(M, r0 + CHUNK))
        xr = x[sl].clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5)
        yr.backward(dy[sl].float())
        ref_dw += wr.grad.double()
        ref_db += br.grad.double()
        worst["y32"] = max(worst["y32"], rel_l2(y32[sl], yr.detach()))
        worst["y16"] = max(worst["y16"], rel_l2(y16[sl], yr.detach()))
        worst["dx32"] = max(worst["dx32"], rel_l2(dx32[sl], xr.grad + dres[sl]))
        worst["dx16"] = max(worst["dx16"], rel_l2(dx16[sl], xr.grad + dres[sl]))
        mu = x[sl].mean(-1)
        worst["mean"] = max(worst["mean"], float((mean[sl] - mu).abs().max()))
        worst["rstd"] = max(worst["rstd"], rel_l2(rstd[sl], torch.rsqrt(x[sl].var(-1, unbiased=False) + 1e-5)))
    rdw, rdb = rel_l2(dw, ref_dw.float()), rel_l2(db, ref_db.float())
    _report(f"bench-size layernorm [{M}x{C}] ({tag}): " + " ".join(f"{k}={v:.2e}" for k, v in worst.items()) + f" dgamma={rdw:.2e} dbeta={rdb:.2e}")
    assert worst["y32"] <= 1e-5 and worst["dx32"] <= 1e-5 and worst["rstd"] <= 1e-5 and worst["mean"] <= 1e-5
    assert worst["y16"] <= 3e-3 and worst["dx16"] <= 3e-3                      # one bf16 rounding
    assert rdw <= 1e-4 and rdb <= 1e-4                                          # fp32 atomics over 256 workgroups x M / 256 rows


def test_attention_dense_at_bench_size(dev, nan_outputs):
    """image tower: B = 4096 sequences of 50 tokens, 12 heads (49 152 workgroups per launch)"""
    from open_clip_amd import ops
    B, L, H = B_BENCH, 50, 12
    C = H * 64
    g = torch.Generator(device=dev).manual_seed(11)
    qkv = bf(torch.randn(B * L, 3 * C, device=dev, generator=g) * 1.5)
    dout = bf(torch.randn(B * L, C, device=dev, generator=g))
    out, lse = ops.attn_fwd(qkv, B, L, H, False, 0.125)
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, False, 0.125)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all() and torch.isfinite(dqkv.float()).all(), "rows never written"
    w = [0.0, 0.0, 0.0]
    CB = 256
    for b0 in range(0, B, CB):
        rows = slice(b0 * L, (b0 + CB) * L)
        x = qkv[rows].float().requires_grad_(True)
        ref, ref_lse = _attn_ref(x, CB, L, H, False)
        ref.backward(dout[rows].float())
        w[0] = max(w[0], rel_l2(out[rows], ref.detach()))
        w[1] = max(w[1], rel_l2(lse.reshape(B, H, L)[b0:b0 + CB], ref_lse.detach().reshape(CB, H, L)))
        w[2] = max(w[2], rel_l2(dqkv[rows], x.grad))
    _report(f"bench-size attention dense [B{B} L{L} H{H}]: out={w[0]:.2e} lse={w[1]:.2e} dqkv={w[2]:.2e}")
    assert w[0] <= 6e-3 and w[1] <= 1e-5 and w[2] <= 1.5e-2


@pytest.mark.parametrize("bucketed", [False, True])
def test_attention_packed_text_at_bench_size(dev, nan_outputs, bucketed):
    """text tower: 4096 captions of the bench's length distribution packed into 177 803 rows, 8 heads, causal -- one launch sized for
    77 tokens, and the three launches (1 / 2 / 3 blocks of 32 rows) the training step uses"""
    from open_clip_amd import ops
    B, L, H = B_BENCH, 77, 8
    C = H * 64
    lens = _text_lengths(B)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
    M = int(off[-1])
    seq_off = off.to(torch.int32).to(dev)
    g = torch.Generator(device=dev).manual_seed(12)
    qkv = bf(torch.randn(M, 3 * C, device=dev, generator=g) * 1.5)
    dout = bf(torch.randn(M, C, device=dev, generator=g))
    lay = seq_off
    if bucketed:
        nb = (lens + 31) // 32
        order = torch.sort(nb, stable=True).indices.to(torch.int32).to(dev)
        lay = ops.SeqLayout(seq_off, order, torch.bincount(nb - 1, minlength=(L + 31) // 32).tolist())
    out, lse = ops.attn_fwd(qkv, B, L, H, True, 0.125, seq_off=lay)
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, True, 0.125, seq_off=lay)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all(), "packed rows never written"
    # fp32 reference on the padded batch: under the causal mask a valid query never sees the (zero) rows behind its caption, and those rows
    # receive no gradient because their dout is zero
    pos = torch.arange(L, device=dev)[None, :]
    valid = pos < lens.to(dev)[:, None]                                     # [B, L]
    w = [0.0, 0.0, 0.0]
    CB = 512
    for b0 in range(0, B, CB):
        v = valid[b0:b0 + CB]
        r0, r1 = int(off[b0]), int(off[b0 + CB])
        pad_q = torch.zeros(CB, L, 3 * C, device=dev)
        pad_q[v] = qkv[r0:r1].float()
        pad_d = torch.zeros(CB, L, C, device=dev)
        pad_d[v] = dout[r0:r1].float()
        x = pad_q.reshape(CB * L, 3 * C).requires_grad_(True)
        ref, ref_lse = _attn_ref(x, CB, L, H, True)
        ref.backward(pad_d.reshape(CB * L, C))
        w[0] = max(w[0], rel_l2(out[r0:r1], ref.detach().reshape(CB, L, C)[v]))
        w[2] = max(w[2], rel_l2(dqkv[r0:r1], x.grad.reshape(CB, L, 3 * C)[v]))
        got_lse = lse.reshape(B, H, L)[b0:b0 + CB].permute(0, 2, 1)[v]      # [rows, H]
        w[1] = max(w[1], rel_l2(got_lse, ref_lse.detach().reshape(CB, H, L).permute(0, 2, 1)[v]))
    _report(f"bench-size attention packed [B{B} Lmax{L} H{H} rows {M}, {'3 buckets' if bucketed else 'one launch'}]: out={w[0]:.2e} lse={w[1]:.2e} dqkv={w[2]:.2e}")
    assert w[0] <= 6e-3 and w[1] <= 1e-5 and w[2] <= 1.5e-2


def test_vitb32_step_at_batch_512_against_cpu_oracle():
    """one whole ViT-B-32 training step (both towers, ClipLoss, backward) at batch 512 -- packed text tower, pooled last blocks, two tower
    streams, the persistent GEMM kernels on 25 600 / ~22 000 rows -- against the fp32 CPU oracle on the same inputs and weights"""
    import time
    from oracle import clip_oracle as O
    from tests.test_model_gpu import FEAT_TOL, LOSS_TOL, _build, _grad_tol, _step
    cfg = get_model_config("ViT-B-32")
    B = 512
    state = init_state_dict(cfg, seed=0, perturb=True)
    batch = synthetic_batch(cfg, B, seed=4321)
    import os
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))  # batch 512 scales to many host cores (the bench's batch-32 sample does not)
    t0 = time.time()
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg)
    t_cpu = time.time() - t0
    model = _build(cfg, state)
    out, loss = _step(model, batch)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"oracle[ViT-B-32,B{B}]: feat max_abs {fi:.3e}/{ft:.3e} loss {float(loss):.6f} vs {float(outs['loss']):.6f} (oracle {t_cpu:.0f} s on the host)")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    worst = []
    for k, p in model.named_parameters():
        ref = grads[k]
        rel = float((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        worst.append((rel / _grad_tol(float(ref.norm()), gmax, ref.ndim), rel, k))
    worst.sort(reverse=True)
    for frac, rel, k in worst[:8]:
        _report(f"oracle[ViT-B-32,B{B}]:   grad rel_l2={rel:.3e} ({frac:.2f} of its tolerance) {k}")
    assert worst[0][0] <= 1.0, worst[0]
    # ---- the chunked fp32 GPU reference (oracle/gpu_fp32.py) is PINNED here, against the same CPU-oracle outputs: it is what stands in for
    # the oracle at the bench's own batch (next test), where the host would need the better part of an hour
    from oracle import gpu_fp32
    del model, out, loss
    torch.cuda.empty_cache()
    g_outs, g_grads = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=128)  # 4 chunks: the chunking identity is exercised
    pi = float((g_outs["image_features"].cpu() - outs["image_features"]).abs().max())
    pt = float((g_outs["text_features"].cpu() - outs["text_features"]).abs().max())
    pl = abs(float(g_outs["loss"]) - float(outs["loss"]))
    pg = max((float((g_grads[k].cpu() - grads[k]).norm() / grads[k].norm().clamp_min(1e-30)), k) for k in grads)
    _report(f"fp32 GPU reference vs CPU oracle [ViT-B-32,B{B}]: feat max_abs {pi:.2e}/{pt:.2e} loss |d| {pl:.2e} worst grad rel_l2 {pg[0]:.2e} ({pg[1]})")
    assert set(g_grads) == set(grads)
    assert pi <= 1e-5 and pt <= 1e-5 and pl <= 1e-5 and pg[0] <= 1e-4, (pi, pt, pl, pg)


_BENCH_REF = {}  # fp32 reference + eager yardstick of the bench-batch step: computed once, shared by the three residual-stream modes


@pytest.mark.parametrize("image_stream", ["fp32", "bf16", "bf16-fp32grad"])
def test_vitb32_step_at_the_bench_batch_against_fp32_gpu_reference(image_stream):
    """Round 6: the same bound for every dtype of the image tower's residual stream -- fp32 (the stricter default), bf16 (what the reference's own
    autocast runs: bench.py's headline configuration) and bf16 with an fp32 residual-gradient path.  On the bf16 stream a gradient may exceed 2e-2
    only where the reference's own policy does and no further than that policy's error (``stream_bound``; measured: token_embedding.weight 2.23e-2
    against 2.59e-2 for eager autocast -- the image features carry the stream's rounding, rel-L2 8.3e-3 against eager's 9.3e-3 and 3.4e-3 on the fp32
    stream, into the text tower's gradients through the loss; every other tensor <= 1.8e-2).  The eager yardstick follows the reference's LayerNorm
    (layers.py:23-26: result cast back to the input dtype) since round 6 -- before, its image stream was fp32 from ln_pre on.

    VERDICT r3 #2: the WHOLE step at the bench's own batch (ViT-B-32, 4096 pairs: the launches bench.py times -- 204 800 image rows, the
    packed text rows, the persistent GEMMs' full tile walks and half-tile tails, 49 152-workgroup attention launches, the fused 4096 x 4096
    logits + cross-entropy) against the chunked fp32 GPU reference, which the previous test pins to the CPU oracle at batch 512: features, loss
    and ALL 302 gradients.

    Bound at this batch: EVERY gradient within 2e-2 rel-L2 (measured: worst 1.3e-2, 1-D worst 1.2e-2) -- tighter than the tolerance classes of
    the small-batch tests (3.5e-2 matrices / 5e-2 1-D, tests/test_model_gpu.py), which stay as they are because batches of 8-24 are noisier.
    The same step as plain PyTorch eager operators under torch.amp.autocast(bf16) (the reference's own --precision amp_bf16 on this GPU,
    oracle/torch_eager.py) is measured against the same fp32 reference as a yardstick of what the POLICY costs: the native path must not be
    less accurate than it (medians of the 1-D and of the matrix gradients).

    History (profiles/r04_parity_report.txt): this test is what found the loss kernel's common-mode bias.  Bias / LayerNorm-bias gradients are
    column sums over thousands of rows that nearly cancel (a contrastive batch at initialisation), which amplifies any error that has the SAME
    sign in every row: the label entry (p - 1) * grad_scale of the logit gradient, rounded to bf16, lost its p in every row -- the 1-D gradients of
    the last text block stood at 5.2e-2 (1.04 of their bound; eager amp_bf16: 1.8e-2) and every other gradient at twice eager's error.  With the
    one-hot part applied exactly (loss.py::_PairTerm.dX / dY) they are at 1.1e-2, below eager's.  Taking the column sums from fp32 values
    (LayerNorm backward's dcol) had changed nothing: the error was in the summands, not in the summation."""
    from oracle import gpu_fp32, torch_eager
    from tests.test_model_gpu import FEAT_TOL, LOSS_TOL, _build, _grad_tol, _step
    from tests.test_parity_at_size_gpu import stream_bound
    cfg = get_model_config("ViT-B-32")
    B = 4096
    state = init_state_dict(cfg, seed=0, perturb=True)
    batch = synthetic_batch(cfg, B, seed=1234)  # the bench's own batch
    if "grads" not in _BENCH_REF:
        outs, grads = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=512)
        outs = {k: v.cpu() for k, v in outs.items()}
        grads = {k: v.cpu() for k, v in grads.items()}
        torch.cuda.empty_cache()
        a_outs, a_grads = torch_eager.amp_step_grads(cfg, state, batch["image"].cuda(), batch["text"].cuda())
        amp_rel = {k: float((a_grads[k].cpu() - grads[k]).norm() / grads[k].norm().clamp_min(1e-30)) for k in grads}
        amp_feat = max(float((a_outs[k].cpu() - outs[k]).abs().max()) for k in ("image_features", "text_features"))
        del a_outs, a_grads
        torch.cuda.empty_cache()
        _BENCH_REF.update(outs=outs, grads=grads, amp_rel=amp_rel, amp_feat=amp_feat)
    outs, grads, amp_rel, amp_feat = _BENCH_REF["outs"], _BENCH_REF["grads"], _BENCH_REF["amp_rel"], _BENCH_REF["amp_feat"]
    model = _build(cfg, state, image_stream=image_stream)
    out, loss = _step(model, batch)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"fp32-GPU-reference[ViT-B-32,B{B},stream {image_stream}]: feat max_abs {fi:.3e}/{ft:.3e} (eager amp_bf16: {amp_feat:.3e}) loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    worst = []
    for k, p in model.named_parameters():
        ref = grads[k]
        rel = float((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        tol = stream_bound(min(_grad_tol(float(ref.norm()), gmax, ref.ndim), 2e-2), amp_rel[k], image_stream)
        worst.append((rel / tol, rel, k))
    worst.sort(reverse=True)
    for frac, rel, k in worst[:12]:
        _report(f"fp32-GPU-reference[ViT-B-32,B{B},stream {image_stream}]:   grad rel_l2={rel:.3e} ({frac:.2f} of its bound; eager amp_bf16 {amp_rel[k]:.3e}) {k}")
    med = lambda v: sorted(v)[len(v) // 2]
    nat_1d, nat_2d = [rel for _, rel, k in worst if grads[k].ndim <= 1], [rel for _, rel, k in worst if grads[k].ndim >= 2]
    amp_1d, amp_2d = [amp_rel[k] for k in grads if grads[k].ndim <= 1], [amp_rel[k] for k in grads if grads[k].ndim >= 2]
    _report(f"fp32-GPU-reference[ViT-B-32,B{B},stream {image_stream}]:   1-D gradients: native median rel_l2 {med(nat_1d):.3e} worst {max(nat_1d):.3e}; eager amp_bf16 median {med(amp_1d):.3e} "
            f"worst {max(amp_1d):.3e}; matrices: native median {med(nat_2d):.3e} worst {max(nat_2d):.3e}; eager amp_bf16 median {med(amp_2d):.3e} worst {max(amp_2d):.3e}")
    assert len(worst) == 302 and worst[0][0] <= 1.0, worst[0]
    assert med(nat_1d) <= 1.1 * med(amp_1d) and med(nat_2d) <= 1.1 * med(amp_2d), "the native step is less accurate than eager PyTorch under the same amp_bf16 policy"


def test_eval_and_inference_mode_calls(dev):
    """what the reference does with the model after every epoch (open_clip_train/train.py:536-713 ``evaluate``: model.eval(), autocast,
    ``model(images, texts)`` under no_grad; open_clip/zero_shot_classifier.py:56 / zero_shot.py:105: one modality at a time,
    ``encode_text(texts, normalize=True)`` / ``model(image=images)``; task/clip_task.py:48-50 ``eval_forward``): the autograd Functions,
    the operand cache and the packed-text read-back under ``torch.inference_mode()`` + autocast, against the CPU oracle"""
    from oracle import clip_oracle as O
    from tests.test_model_gpu import FEAT_TOL, _build
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=5, perturb=True)
    batch = synthetic_batch(cfg, 12, seed=8)
    with torch.no_grad():
        ref = O.clip_forward(batch["image"], batch["text"], {k: v.float() for k, v in state.items()}, cfg)
    model = _build(cfg, state).eval()
    img, txt = batch["image"].to(dev), batch["text"].to(dev)

    def close(name, got, want):
        err = float((got.float().cpu() - want).abs().max())
        _report(f"eval-mode {name}: max_abs {err:.3e}")
        assert err <= FEAT_TOL, (name, err)

    with torch.inference_mode(), torch.amp.autocast("cuda", dtype=torch.bfloat16):
        both = model(image=img, text=txt)
        only_i = model(image=img)
        only_t = model(text=txt)
        enc_i = model.encode_image(img, normalize=True)
        enc_t = model.encode_text(txt, normalize=True)
        li, lt = model.get_logits(img, txt)
    close("model(image, text).image_features", both["image_features"], ref["image_features"])
    close("model(image, text).text_features", both["text_features"], ref["text_features"])
    assert only_i["text_features"] is None and only_t["image_features"] is None
    assert torch.equal(only_i["image_features"], both["image_features"]) and torch.equal(only_t["text_features"], both["text_features"])
    assert torch.equal(enc_i, both["image_features"]) and torch.equal(enc_t, both["text_features"])
    scale = float(np.exp(float(state["logit_scale"])))
    want = scale * ref["image_features"] @ ref["text_features"].t()
    assert float((li.float().cpu() - want).abs().max()) <= 0.15 and torch.equal(li.t(), lt)
    assert not both["image_features"].requires_grad
    # no_grad (the accumulation path's feature pass, train.py:246-262) and a following training call see the same weights
    with torch.no_grad():
        ng = model(image=img, text=txt)
    assert torch.equal(ng["image_features"], both["image_features"]) and torch.equal(ng["text_features"], both["text_features"])
    model.train()
    tr = model(image=img, text=txt)
    assert tr["image_features"].requires_grad and torch.equal(tr["image_features"].detach(), both["image_features"])
    # zero-shot classifier shape of use (zero_shot_classifier.py:56-60): class embeddings from text alone, then image @ classifier
    with torch.inference_mode():
        classifier = torch.nn.functional.normalize(model.encode_text(txt, normalize=True).float(), dim=-1).t()
        logits = 100.0 * model.encode_image(img, normalize=True).float() @ classifier
    assert logits.shape == (12, 12) and torch.isfinite(logits).all()


def test_training_step_without_any_host_synchronisation(dev, monkeypatch):
    """a batch that arrives through DeviceBatchPipeline(plan_text_vocab=...) carries its packed text layout
    (computed on the host next to the tokens), so forward + ClipLoss + backward + fused AdamW + clamp enqueue without ONE host
    synchronisation: torch's sync debug mode is set to "error" and Event.synchronize is made to raise for the step.  Same features and
    loss as the step that plans on the device and reads the row count back."""
    import math
    from open_clip_amd.input_pipeline import DeviceBatchPipeline
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    from tests.test_model_gpu import _build
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=2, perturb=True)
    B, S, L = 24, cfg["vision_cfg"]["image_size"], cfg["text_cfg"]["context_length"]
    batch = synthetic_batch(cfg, B, seed=6)
    g = torch.Generator().manual_seed(1)
    pixels = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    model = _build(cfg, state)
    loss_fn = NativeClipLoss()
    opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_caches=weight_caches_of(model))
    pipe = DeviceBatchPipeline(dev, (B, S, S, 3), (B, L), plan_text_vocab=cfg["text_cfg"]["vocab_size"], attn_buckets=model.attn_buckets)

    def step(b):
        opt.zero_grad(set_to_none=True)
        out = model(image=b["image"], text=b["text"])
        loss = loss_fn(**out)
        loss.backward()
        opt.step()
        with torch.no_grad():
            model.logit_scale.clamp_(0, math.log(100))
        return out, loss

    # warm-up step the ordinary way (allocator, operand caches, optimizer state), then the checked step
    pipe.submit(pixels, batch["text"])
    b = pipe.next()
    step(b)
    pipe.release(b)
    torch.cuda.synchronize()
    pipe.submit(pixels, batch["text"])
    b = pipe.next()
    assert hasattr(b["text"], "_ocn_host_plan")

    def no_event_sync(self):
        raise AssertionError("host synchronisation on an event inside the training step")

    monkeypatch.setattr(torch.cuda.Event, "synchronize", no_event_sync)
    torch.cuda.set_sync_debug_mode("error")
    try:
        out, loss = step(b)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    monkeypatch.undo()
    pipe.release(b)
    torch.cuda.synchronize()
    # the same step planned on the device (plain device tensors): identical features, same loss
    model2 = _build(cfg, state)
    ref = model2(image=pixels.to(dev), text=batch["text"].to(dev))
    # (model has taken one optimizer step at lr 1e-4 before the checked one: compare the layouts through the features of a fresh forward instead)
    model3 = _build(cfg, state)
    pipe.submit(pixels, batch["text"])
    b3 = pipe.next()
    planned = model3(image=b3["image"], text=b3["text"])
    assert torch.equal(planned["text_features"], ref["text_features"]) and torch.equal(planned["image_features"], ref["image_features"])
    assert math.isfinite(float(loss))
    # ids outside the vocabulary raise from the host plan as they do from the device plan
    bad = batch["text"].clone()
    bad[0, 1] = cfg["text_cfg"]["vocab_size"] + 5
    pipe.release(b3)
    pipe.submit(pixels, bad)
    bb = pipe.next()
    with pytest.raises(IndexError):
        model3(image=bb["image"], text=bb["text"])
