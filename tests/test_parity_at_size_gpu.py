"""Parity AT SIZE for what round 4's batch-4096 ClipLoss test left open (missing #3 / weak #1).  That test found a defect -- the
bf16-rounded label entry of the logit gradient, a common-mode bias that doubled every parameter gradient's error -- which is INVISIBLE below a
batch of about 512: the small-batch goldens cannot stand in for it, and the fix (the exact rank-1 completion in ``loss.py::_PairTerm.dX / dY``)
was also applied to SigLIP and to the branches with ``label_offset != 0`` without a test at a size that could see it.  Here, on the MI355X
through the C ABI:

  1. the fp32 GPU reference (``oracle/gpu_fp32.py``) is PINNED to the CPU oracle (which is pinned to the reference's own outputs, tests/golden/)
     for the forms the round-4 test did not use: SigLipLoss, ViT-L-14, ViT-H-14 + SigLIP, and ``loss_reference`` (gathered-feature losses);
  2. one whole ViT-B-32 SigLIP step at the bench's batch of 4096 (reference: loss.py:356-367) -- features, loss, all 303 gradients;
  3. the distributed branches of ClipLoss at BASELINE config 3's size -- R = 4096 rows per rank, N = 32768, E = 512: the row-sharded global
     loss (every rank's launch, label_offset = rank * 4096, the reduce-scatter / all-reduce emulated by summing the ranks' own outputs) and the
     reference's redundant global form -- and of SigLipLoss at config 5's size (R = 1024, N = 8192, E = 1024): dI, dT, d logit_scale, d logit_bias
     against fp32 torch on the gathered features (reference: loss.py:91-141, :406-489), the rank-1 completion included;
  4. BASELINE config 4 (ViT-L-14, block recompute) and config 5 (ViT-H-14 + SigLIP, block recompute) as whole steps at batch 512 against the
     chunked fp32 GPU reference, with the same steps as eager operators under autocast(bf16) as the yardstick of what the POLICY costs.

Tolerances: features / loss as everywhere (tests/test_model_gpu.py); gradients <= 2e-2 rel-L2 for EVERY tensor at batch 4096 (the bound of the
round-4 test), the tolerance classes of tests/test_model_gpu.py at batch 512 (measured values in gpurun_out/parity_report.txt), and in every
whole-step case the native medians must not exceed eager amp_bf16's by more than 10 %."""
import os

import pytest
import torch

from open_clip_amd.configs import get_model_config
from open_clip_amd.synth import init_state_dict, synthetic_batch
from tests.test_kernels_gpu import _report

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp_min(1e-30))


def _host_threads():
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))


# ---- 1. pinning ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,B,siglip,chunk", [("ViT-B-32", 96, True, 32), ("ViT-L-14", 4, False, 3), ("ViT-H-14", 4, True, 3)])
def test_fp32_gpu_reference_is_pinned_to_the_cpu_oracle(name, B, siglip, chunk):
    """oracle/gpu_fp32.py::step_reference for the forms used below, against oracle/clip_oracle.py on the same inputs and weights, in ragged
    chunks (the chunking identity is exercised): features, loss and EVERY gradient"""
    from oracle import clip_oracle as O
    from oracle import gpu_fp32
    cfg = get_model_config(name)
    state = init_state_dict(cfg, seed=3, perturb=True, siglip=siglip)
    batch = synthetic_batch(cfg, B, seed=77)
    _host_threads()
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg, siglip=siglip)
    g_outs, g_grads = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=chunk, siglip=siglip)
    pi = float((g_outs["image_features"].cpu() - outs["image_features"]).abs().max())
    pt = float((g_outs["text_features"].cpu() - outs["text_features"]).abs().max())
    pl = abs(float(g_outs["loss"]) - float(outs["loss"]))
    pg = max((_rel(g_grads[k], grads[k]), k) for k in grads)
    _report(f"fp32 GPU reference vs CPU oracle [{name},B{B}{',SigLIP' if siglip else ''}]: feat max_abs {pi:.2e}/{pt:.2e} loss |d| {pl:.2e} "
            f"worst grad rel_l2 {pg[0]:.2e} ({pg[1]})")
    assert set(g_grads) == set(grads) and (("logit_bias" in grads) == siglip)
    assert pi <= 1e-5 and pt <= 1e-5 and pl <= 2e-5 * max(1.0, abs(float(outs["loss"]))) and pg[0] <= 2e-4, (pi, pt, pl, pg)


def test_loss_reference_is_pinned_to_the_cpu_oracle():
    """oracle/gpu_fp32.py::loss_reference (the gathered-feature losses used in 3.) against oracle/clip_oracle.py::clip_loss (global form,
    loss.py:106-107) and ::siglip_loss (rank's rows against every rank's texts, loss.py:406-489) on CPU autograd"""
    from oracle import clip_oracle as O
    from oracle import gpu_fp32
    g = torch.Generator().manual_seed(5)
    W, B, E = 4, 24, 64
    I = torch.nn.functional.normalize(torch.randn(W * B, E, generator=g), dim=-1)
    T = torch.nn.functional.normalize(torch.randn(W * B, E, generator=g), dim=-1)
    # ClipLoss, global
    Ic, Tc, s = I.clone().requires_grad_(True), T.clone().requires_grad_(True), torch.tensor(14.3, requires_grad=True)
    loss = O.clip_loss(Ic[:B], Tc[:B], s, all_image_features=Ic, all_text_features=Tc, local_loss=False)
    loss.backward()
    ref = gpu_fp32.loss_reference(I.cuda(), T.cuda(), torch.tensor(14.3).cuda())
    assert abs(float(ref["loss"]) - float(loss)) <= 1e-5
    assert _rel(ref["dI"], Ic.grad) <= 1e-5 and _rel(ref["dT"], Tc.grad) <= 1e-5 and _rel(ref["dscale"], s.grad) <= 1e-4
    # SigLipLoss, rank 2 of 4
    r = 2
    Ic, Tc = I.clone().requires_grad_(True), T.clone().requires_grad_(True)
    s, b = torch.tensor(10.0, requires_grad=True), torch.tensor(-10.0, requires_grad=True)
    loss = O.siglip_loss(Ic[r * B:(r + 1) * B], [Tc[q * B:(q + 1) * B] for q in range(W)], s, b, rank=r)
    loss.backward()
    ref = gpu_fp32.loss_reference(I.cuda(), T.cuda(), torch.tensor(10.0).cuda(), torch.tensor(-10.0).cuda(), siglip=True, rows=(r * B, (r + 1) * B))
    assert abs(float(ref["loss"]) - float(loss)) <= 1e-5 * max(1.0, float(loss))
    assert _rel(ref["dI"], Ic.grad[r * B:(r + 1) * B]) <= 1e-5 and _rel(ref["dT"], Tc.grad) <= 1e-5
    assert _rel(ref["dscale"], s.grad) <= 1e-4 and _rel(ref["dbias"], b.grad) <= 1e-5


# ---- whole steps ----------------------------------------------------------------------------------------------------------------------------
def stream_bound(tol, eager_rel, stream):
    """tolerance of one gradient by the image tower's residual-stream dtype.  fp32 stream: ``tol``.  bf16 stream = the reference's own amp_bf16
    policy there (layers.py:23-26; tests/test_reference_dropin.py::test_reference_stream_dtypes_under_autocast): a gradient may exceed ``tol`` only
    where the reference's policy itself does (eager autocast against the same fp32 reference, ``eager_rel``) and no further than that policy's own
    error -- never beyond 2 x tol.  Measured: ViT-B-32 at batch 4096 one tensor of 302 (token_embedding.weight 2.23e-2, policy 2.59e-2); ViT-L-14 at
    batch 512 (24 image blocks: 48 roundings of the stream) up to 4.66e-2 against the policy's 4.93e-2, medians 3.0e-2 / 3.4e-2 against 3.5e-2 / 3.7e-2."""
    return tol if stream == "fp32" else max(tol, min(2.0 * tol, eager_rel))


def _whole_step_case(name, B, siglip, chunk, recompute, bound, tag, eager_whole_batch, streams=("fp32",)):
    """native step vs chunked fp32 GPU reference; eager amp_bf16 (same operators under autocast) against the same reference as yardstick;
    ``streams``: the residual-stream dtypes of the image tower to run against that one reference"""
    from oracle import gpu_fp32, torch_eager
    from tests.test_model_gpu import FEAT_TOL, LOSS_TOL, _build, _grad_tol, _step
    cfg = get_model_config(name)
    state = init_state_dict(cfg, seed=0, perturb=True, siglip=siglip)
    batch = synthetic_batch(cfg, B, seed=1234)
    outs, grads = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=chunk, siglip=siglip)
    outs = {k: v.cpu() for k, v in outs.items()}
    grads = {k: v.cpu() for k, v in grads.items()}
    torch.cuda.empty_cache()
    if eager_whole_batch and not siglip:
        a_outs, a_grads = torch_eager.amp_step_grads(cfg, state, batch["image"].cuda(), batch["text"].cuda())
    else:  # in chunks (whole-batch eager activations of the big towers do not fit next to everything else; same identity as the reference's)
        a_outs, a_grads = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=chunk, siglip=siglip, amp=True)
    amp_rel = {k: _rel(a_grads[k], grads[k]) for k in grads}
    amp_feat = max(float((a_outs[k].float().cpu() - outs[k]).abs().max()) for k in ("image_features", "text_features"))
    del a_outs, a_grads
    torch.cuda.empty_cache()
    for stream in streams:
        _whole_step_check(cfg, state, batch, siglip, recompute, bound, f"{tag},stream {stream}" if len(streams) > 1 else tag, outs, grads, amp_rel, amp_feat, stream)
        torch.cuda.empty_cache()


def _whole_step_check(cfg, state, batch, siglip, recompute, bound, tag, outs, grads, amp_rel, amp_feat, stream):
    from tests.test_model_gpu import FEAT_TOL, LOSS_TOL, _build, _grad_tol, _step
    model = _build(cfg, state, siglip=siglip, image_stream=stream)
    if recompute:
        model.set_grad_checkpointing(True)  # every block recomputed, as the reference's --grad-checkpointing (transformer.py:577-585)
    out, loss = _step(model, batch, siglip=siglip)
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"fp32-GPU-reference[{tag}]: feat max_abs {fi:.3e}/{ft:.3e} (eager amp_bf16: {amp_feat:.3e}) loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL * max(1.0, abs(float(outs["loss"])) / 8.0)
    gmax = max(float(v.norm()) for v in grads.values())
    scale_cond = 0.0
    if siglip:
        # d loss / d logit_scale (the parameter: log scale) = sum_ij g_ij * s cos_ij with g = (sigmoid(z) - [i == j]) / B.  At SigLIP's initialisation
        # (scale 10, bias -10, near-orthogonal features) its positive and negative parts are each ~1e-3 and of opposite sign: for ViT-H-14 at batch
        # 512 the sum is 1.2e-3, the size of the shift that bf16-rounded WEIGHTS (the same perturbation for every sample, hence coherent in
        # mean_i cos_ii) put on it -- native 0.88, eager amp_bf16 0.09 relative error there, both meaningless as RELATIVE errors.  The bound for this
        # one scalar is therefore absolute, against the conditioning of the sum: |error| <= 2e-2 * sum_ij |g_ij| * |s cos_ij|.
        sI, T = float(torch.exp(state["logit_scale"].float())) * outs["image_features"].double(), outs["text_features"].double()
        z = sI @ T.t()
        g = torch.sigmoid(z + float(state["logit_bias"])) - torch.eye(z.shape[0], dtype=z.dtype)
        scale_cond = float((g.abs() * z.abs()).sum() / z.shape[0])
        _report(f"fp32-GPU-reference[{tag}]:   d/d logit_scale = {float(grads['logit_scale']):.4e}; conditioning sum |g| |s cos| = {scale_cond:.4e}")
    worst = []
    for k, p in model.named_parameters():
        ref = grads[k]
        assert p.grad is not None, k
        rel = _rel(p.grad, ref)
        tol = _grad_tol(float(ref.norm()), gmax, ref.ndim)
        tol = min(tol, bound) if bound else tol
        if k == "logit_scale" and siglip:
            tol = max(tol, 2e-2 * scale_cond / max(abs(float(ref)), 1e-30))
        tol = stream_bound(tol, amp_rel[k], stream)
        worst.append((rel / tol, rel, k))
    worst.sort(reverse=True)
    for frac, rel, k in worst[:10]:
        _report(f"fp32-GPU-reference[{tag}]:   grad rel_l2={rel:.3e} ({frac:.2f} of its bound; eager amp_bf16 {amp_rel[k]:.3e}) {k}")
    med = lambda v: sorted(v)[len(v) // 2]
    nat_1d, nat_2d = [rel for _, rel, k in worst if grads[k].ndim <= 1], [rel for _, rel, k in worst if grads[k].ndim >= 2]
    amp_1d, amp_2d = [amp_rel[k] for k in grads if grads[k].ndim <= 1], [amp_rel[k] for k in grads if grads[k].ndim >= 2]
    _report(f"fp32-GPU-reference[{tag}]:   {len(worst)} gradients; 1-D: native median rel_l2 {med(nat_1d):.3e} worst {max(nat_1d):.3e}; eager amp_bf16 median "
            f"{med(amp_1d):.3e} worst {max(amp_1d):.3e}; matrices: native median {med(nat_2d):.3e} worst {max(nat_2d):.3e}; eager amp_bf16 median "
            f"{med(amp_2d):.3e} worst {max(amp_2d):.3e}")
    assert len(worst) == len(grads) and worst[0][0] <= 1.0, worst[0]
    assert med(nat_1d) <= 1.1 * med(amp_1d) and med(nat_2d) <= 1.1 * med(amp_2d), "the native step is less accurate than eager PyTorch under the same amp_bf16 policy"


def test_vitb32_siglip_step_at_the_bench_batch_against_fp32_gpu_reference():
    """2. the SigLIP form of round 4's loss fix at a batch that can see it: ViT-B-32 + SigLipLoss (logit_scale ln 10, logit_bias -10: main.py:259-261)
    at batch 4096 -- the positives' (sigmoid - 1) * grad_scale entry rounds to -grad_scale in bf16 exactly as the cross-entropy's label entry did;
    G holds sigmoid * grad_scale and the -1 is applied in fp32 (loss.py::_PairTerm.dX / dY).  Every gradient <= 2e-2."""
    _whole_step_case("ViT-B-32", 4096, True, 512, False, 2e-2, "ViT-B-32 SigLIP,B4096", eager_whole_batch=False)


def test_vitl14_recompute_step_at_batch_512_against_fp32_gpu_reference():
    """4. BASELINE config 4's model and mode (ViT-L-14, every block recomputed, ClipLoss) at batch 512: 131 584 image rows through the 257-token
    attention kernels, patch 14 (K padded 588 -> 640), 24 + 12 blocks"""
    _whole_step_case("ViT-L-14", 512, False, 64, True, None, "ViT-L-14 ckpt,B512", eager_whole_batch=False, streams=("fp32", "bf16"))


def test_vith14_siglip_recompute_step_at_batch_512_against_fp32_gpu_reference():
    """4. BASELINE config 5's model and mode (ViT-H-14 + SigLipLoss, every block recomputed) at batch 512: head_dim 80 streamed attention, width 1280,
    32 + 24 blocks, embed 1024"""
    _whole_step_case("ViT-H-14", 512, True, 64, True, None, "ViT-H-14 SigLIP ckpt,B512", eager_whole_batch=False, streams=("fp32", "bf16"))


# ---- 3. the loss's distributed branches at config-3 / config-5 size ----------------------------------------------------------------------------
from tests.rank_emulation import Collectives as _Collectives, unit_features as _unit_features  # noqa: E402


def test_cliploss_distributed_branches_at_config3_size(monkeypatch):
    """BASELINE config 3: world 8 x local 4096, E = 512 -> N = 32768.  (a') further down: local_loss + gather_with_grad on every rank.  (a) the row-sharded global loss (bench.py's default for N > 1): EVERY rank's
    launch -- [4096 x 32768] logits both ways with label_offset = 4096 * rank -- with the reduce-scatter of the column gradients and the scalar
    all-reduce formed from the ranks' own contributions; (b) the reference's redundant global form on one rank (two [32768 x 32768] fused
    logits + cross-entropy launches, the local slice of the result).  Against fp32 torch on the gathered features: loss, d image_features,
    d text_features (all N rows: every rank's slice), d logit_scale.  The rank-1 completion of the logit gradient (loss.py:133-159) indexes
    ``y16[off:off+R]`` / ``out[off:off+R]``: wrong offsets or a missing completion show as O(1) errors here."""
    from open_clip_amd import loss as L
    from oracle import gpu_fp32
    dev = torch.device("cuda:0")
    W, B, E = 8, 4096, 512
    N = W * B
    I, T = _unit_features(N, E, 11, dev)
    s = torch.tensor(14.285714, device=dev)
    ref = gpu_fp32.loss_reference(I, T, s)
    packed = torch.cat([I, T], dim=1)
    coll = _Collectives(packed)
    monkeypatch.setattr(L, "_all_gather", coll.all_gather)
    monkeypatch.setattr(L, "_reduce_scatter_sum", coll.reduce_scatter)
    monkeypatch.setattr(L, "_all_reduce_sum", coll.all_reduce)
    # (a) row-sharded: every rank
    dI_loc, dT_loc, dscale, losses = [], [], [], []
    for r in range(W):
        Ir, Tr = I[r * B:(r + 1) * B].clone().requires_grad_(True), T[r * B:(r + 1) * B].clone().requires_grad_(True)
        sr = s.clone().requires_grad_(True)
        loss = L.NativeClipLoss(local_loss=False, gather_with_grad=False, rank=r, world_size=W, row_sharded=True)(Ir, Tr, sr)
        loss.backward()
        dI_loc.append(Ir.grad)
        dT_loc.append(Tr.grad)
        dscale.append(sr.grad)
        losses.append(loss.detach())
    assert len(coll.rs_inputs) == W and len(coll.ar_inputs) == W
    through_cols = torch.stack(coll.rs_inputs).sum(0)  # [N, 2E]: what the reduce-scatter delivers, rank q's slice = rows q*B ..
    dI = torch.cat(dI_loc) + through_cols[:, :E]
    dT = torch.cat(dT_loc) + through_cols[:, E:]
    loss_total = float(torch.stack(losses).sum())        # the all-reduce of the ranks' partial row sums
    dscale_total = float(torch.stack(dscale).sum())
    eI, eT = _rel(dI, ref["dI"]), _rel(dT, ref["dT"])
    per_rank = max(max(_rel(dI[r * B:(r + 1) * B], ref["dI"][r * B:(r + 1) * B]), _rel(dT[r * B:(r + 1) * B], ref["dT"][r * B:(r + 1) * B])) for r in range(W))
    _report(f"ClipLoss row-sharded @ W8 x 4096 x E512: loss {loss_total:.6f} vs {float(ref['loss']):.6f}; dI rel_l2 {eI:.3e} dT {eT:.3e} (worst rank slice {per_rank:.3e}); "
            f"dscale {dscale_total:.6e} vs {float(ref['dscale']):.6e}")
    assert abs(loss_total - float(ref["loss"])) <= 2e-3 and eI <= 1e-2 and eT <= 1e-2 and per_rank <= 1.2e-2
    assert abs(dscale_total - float(ref["dscale"])) <= 2e-2 * abs(float(ref["dscale"])) + 1e-6
    # common-mode check (what the round-4 defect was): the batch SUM of the feature gradients, not only their rel-L2
    cm = max(_rel(dI.sum(0), ref["dI"].sum(0)), _rel(dT.sum(0), ref["dT"].sum(0)))
    _report(f"ClipLoss row-sharded @ W8 x 4096 x E512: |sum_b error| / |sum_b gradient| = {cm:.3e}")
    assert cm <= 5e-2  # (the round-4 defect stood at 0.235 here; measured after the fix: 7e-4 / 4e-3 on the bench's own features)
    # (a') local_loss + gather_with_grad -- the mode used at scale (README.md:255-260; loss.py:103-104, :82-83, :23-26): rank r's loss is the mean over ITS rows of
    # both directions (labels offset by 4096 r), the gathered operands carry gradient and the backward of the all-gather is a reduce-scatter (sum).  Summed over
    # the ranks the losses are W x the global loss, so local gradient + reduce-scattered contributions must equal W x the reference's gradient of the global loss
    coll.rs_inputs.clear()
    dI_loc, dT_loc, dscale, losses = [], [], [], []
    for r in range(W):
        Ir, Tr = I[r * B:(r + 1) * B].clone().requires_grad_(True), T[r * B:(r + 1) * B].clone().requires_grad_(True)
        sr = s.clone().requires_grad_(True)
        loss = L.NativeClipLoss(local_loss=True, gather_with_grad=True, rank=r, world_size=W)(Ir, Tr, sr)
        loss.backward()
        dI_loc.append(Ir.grad), dT_loc.append(Tr.grad), dscale.append(sr.grad), losses.append(loss.detach())
    assert len(coll.rs_inputs) == W  # one reduce-scatter per rank, in its backward
    through = torch.stack(coll.rs_inputs).sum(0)  # [N, 2E]: d I_all | d T_all summed over the ranks = what the reduce-scatter hands back, slice by slice
    dI, dT = torch.cat(dI_loc) + through[:, :E], torch.cat(dT_loc) + through[:, E:]
    lI, lT = _rel(dI, W * ref["dI"]), _rel(dT, W * ref["dT"])
    lsum, dsum = float(torch.stack(losses).sum()), float(torch.stack(dscale).sum())
    _report(f"ClipLoss local_loss + gather_with_grad @ W8 x 4096 x E512: sum of the ranks' losses {lsum:.6f} vs W x global {W * float(ref['loss']):.6f}; dI rel_l2 {lI:.3e} dT {lT:.3e} "
            f"(against W x the global gradient); sum of dscale {dsum:.6e} vs {W * float(ref['dscale']):.6e}")
    assert abs(lsum - W * float(ref["loss"])) <= 2e-2 and lI <= 1e-2 and lT <= 1e-2
    assert abs(dsum - W * float(ref["dscale"])) <= 2e-2 * abs(W * float(ref["dscale"])) + 1e-6
    # (b) the reference's redundant global form, rank 5: local slices of the full-batch gradient
    r = 5
    Ir, Tr, sr = I[r * B:(r + 1) * B].clone().requires_grad_(True), T[r * B:(r + 1) * B].clone().requires_grad_(True), s.clone().requires_grad_(True)
    loss = L.NativeClipLoss(local_loss=False, gather_with_grad=False, rank=r, world_size=W, row_sharded=False)(Ir, Tr, sr)
    loss.backward()
    gI, gT = _rel(Ir.grad, ref["dI"][r * B:(r + 1) * B]), _rel(Tr.grad, ref["dT"][r * B:(r + 1) * B])
    _report(f"ClipLoss global (redundant) @ N32768, rank 5: loss {float(loss):.6f} vs {float(ref['loss']):.6f}; dI rel_l2 {gI:.3e} dT {gT:.3e}; "
            f"dscale {float(sr.grad):.6e} vs {float(ref['dscale']):.6e}")
    assert abs(float(loss) - float(ref["loss"])) <= 2e-3 and gI <= 1e-2 and gT <= 1e-2
    assert abs(float(sr.grad) - float(ref["dscale"])) <= 2e-2 * abs(float(ref["dscale"])) + 1e-6


def test_sigliploss_distributed_at_config5_size(monkeypatch):
    """BASELINE config 5: world 8 x local 1024, E = 1024 -> N = 8192; SigLipLoss on rank 3: its 1024 image rows against all 8192 texts, positives
    at column offset 3072 (reference: loss.py:406-489, the local chunk + seven negative-only chunks), d text_features for ALL ranks' texts
    (what the backward's reduce-scatter sums), d logit_scale, d logit_bias -- against fp32 torch.  Also ``chunk_size`` (loss.py:369-404)."""
    from open_clip_amd import loss as L
    from oracle import gpu_fp32
    dev = torch.device("cuda:0")
    W, B, E, r = 8, 1024, 1024, 3
    N = W * B
    I, T = _unit_features(N, E, 13, dev)
    s, b = torch.tensor(10.0, device=dev), torch.tensor(-10.0, device=dev)
    ref = gpu_fp32.loss_reference(I, T, s, b, siglip=True, rows=(r * B, (r + 1) * B))
    for chunk in (0, 256):
        coll = _Collectives(T)
        monkeypatch.setattr(L, "_all_gather", coll.all_gather)
        monkeypatch.setattr(L, "_reduce_scatter_sum", coll.reduce_scatter)
        Ir, Tr = I[r * B:(r + 1) * B].clone().requires_grad_(True), T[r * B:(r + 1) * B].clone().requires_grad_(True)
        sr, br = s.clone().requires_grad_(True), b.clone().requires_grad_(True)
        loss = L.NativeSigLipLoss(rank=r, world_size=W, chunk_size=chunk)(Ir, Tr, sr, br)
        loss.backward()
        dT_all = coll.rs_inputs[0]  # this rank's contribution to every rank's text gradient (the reduce-scatter's input)
        eI, eT = _rel(Ir.grad, ref["dI"]), _rel(dT_all, ref["dT"])
        _report(f"SigLipLoss @ W8 x 1024 x E1024, rank 3, chunk_size {chunk}: loss {float(loss):.6f} vs {float(ref['loss']):.6f}; dI rel_l2 {eI:.3e} dT_all {eT:.3e}; "
                f"dscale {float(sr.grad):.6e} vs {float(ref['dscale']):.6e}; dbias {float(br.grad):.6e} vs {float(ref['dbias']):.6e}")
        assert abs(float(loss) - float(ref["loss"])) <= 2e-3 * max(1.0, float(ref["loss"])) and eI <= 1e-2 and eT <= 1e-2
        assert abs(float(sr.grad) - float(ref["dscale"])) <= 2e-2 * abs(float(ref["dscale"])) + 1e-6
        assert abs(float(br.grad) - float(ref["dbias"])) <= 1e-3 * abs(float(ref["dbias"])) + 1e-7
        cm = max(_rel(Ir.grad.sum(0), ref["dI"].sum(0)), _rel(dT_all.sum(0), ref["dT"].sum(0)))
        assert cm <= 5e-2, cm


# ---- 4. the two native-only approximations of the MLP epilogues, as an ablation of the reference policy ---------------------------------------------------
def test_gelu_polynomial_and_8bit_saved_derivative_cost_no_parity_digit():
    """two approximations ride on every native step and exist nowhere in the reference -- the GELU evaluated through a degree-17 polynomial normal
    CDF (csrc/ocn_common.h::gelu_both_poly4, |Phi error| 1.24e-5) and its derivative saved for the backward in 8-bit fixed point (q = round((gelu' + 0.13) * 200),
    |error| <= 0.0025).  Grafted ONE AT A TIME into the reference's own policy (the eager autocast step of oracle/torch_eager.py, ViT-B-32 at the bench's batch 4096,
    against the fp32 reference of the same step): the medians of the 1-D and of the matrix gradients' errors must not move by more than 3 % -- the bf16 operands of
    the GEMMs dominate both by two orders of magnitude.  (Until round 6 this lived in tools/parity_ablation.py, outside the suite.  Batch 4096 because eager autocast's
    own error grows steeply below it -- medians 1.4e-2 at 4096, 4.3e-2 at 2048, 1.0e-1 at 1024: its bf16 cross-entropy loses the label entry's p like the native loss
    did until round 4 -- and would bury what is measured here; the native step stands at 1.0e-2 ... 1.7e-2 over the same batches.)"""
    import math
    import os
    import re
    import torch.nn.functional as F
    from oracle import gpu_fp32, torch_eager
    cfg = get_model_config("ViT-B-32")
    B = 4096
    state = init_state_dict(cfg, seed=0, perturb=True)
    batch = synthetic_batch(cfg, B, seed=1234)
    _, ref = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=512)
    ref = {k: v.cpu() for k, v in ref.items()}
    torch.cuda.empty_cache()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open_clip_amd", "csrc", "ocn_common.h")).read()
    body = hdr[hdr.index("void gelu_both_poly4"):]
    body = body[:body.index("\n}\n")]
    first = re.search(r"q = u \* ([-0-9.e+]+)f \+ ([-0-9.e+]+)f;", body)
    coeffs = [float(first.group(1)), float(first.group(2))] + [float(c) for c in re.findall(r"q = q \* u \+ ([-0-9.e+]+)f;", body)]
    assert len(coeffs) == 9
    real_gelu = torch_eager.F.gelu  # (torch_eager.F IS torch.nn.functional: the patch below replaces F.gelu itself)

    class Gelu(torch.autograd.Function):
        """forward: exact erf GELU or the kernel's polynomial form; backward: the derivative as the native path keeps it (fp32 from the fp32 pre-activation,
        then exact or in the 8-bit code)"""
        @staticmethod
        def forward(ctx, x, poly, q8):
            xf = x.float()
            if poly:
                xc = xf.clamp(-4.25, 4.25)
                u = xc * xc
                q = u * coeffs[0] + coeffs[1]
                for c in coeffs[2:]:
                    q = q * u + c
                cdf = xc * q + 0.5
                d = (xc * 0.39894228040143268) * torch.exp2(xf * xf * -0.72134752044448170) + cdf
                y = xf * cdf
            else:
                d = 0.5 * (1 + torch.erf(xf / math.sqrt(2.0))) + xf * torch.exp(-0.5 * xf * xf) / math.sqrt(2 * math.pi)
                y = real_gelu(xf)
            if q8:
                d = torch.round((d + 0.13) * 200.0).clamp_(0, 255).to(torch.uint8)
            ctx.q8 = q8
            ctx.save_for_backward(d)
            return y.to(x.dtype)

        @staticmethod
        def backward(ctx, dy):
            (d,) = ctx.saved_tensors
            d = d.float() / 200.0 - 0.13 if ctx.q8 else d
            return (dy.float() * d).to(dy.dtype), None, None

    med = lambda v: sorted(v)[len(v) // 2]
    res = {}
    for tag, poly, q8 in (("reference policy (erf GELU, derivative recomputed)", None, None), ("derivative saved exactly", False, False),
                          ("derivative in the 8-bit code", False, True), ("polynomial CDF + 8-bit derivative (the native epilogues)", True, True)):
        try:
            if poly is not None:
                torch_eager.F.gelu = lambda x, p=poly, q=q8: Gelu.apply(x, p, q)
            _, grads = torch_eager.amp_step_grads(cfg, state, batch["image"].cuda(), batch["text"].cuda())
        finally:
            torch_eager.F.gelu = real_gelu
        rel = {k: _rel(grads[k], ref[k]) for k in ref}
        res[tag] = (med([v for k, v in rel.items() if ref[k].ndim <= 1]), med([v for k, v in rel.items() if ref[k].ndim >= 2]))
        _report(f"gelu ablation [ViT-B-32,B{B}] {tag:58s} median rel_l2 of the 1-D gradients {res[tag][0]:.4e}, of the matrices {res[tag][1]:.4e}")
        del grads
        torch.cuda.empty_cache()
    base = res["reference policy (erf GELU, derivative recomputed)"]
    for tag, (m1, m2) in res.items():
        assert m1 <= 1.03 * base[0] and m2 <= 1.03 * base[1], (tag, m1, m2, base)
