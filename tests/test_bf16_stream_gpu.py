"""The image tower's bf16 residual stream (round 6; ``NativeCLIP(image_stream="bf16" | "bf16-fp32grad")``) on a real MI355X.

What the reference does under ``--precision amp_bf16``: conv1 runs under autocast (transformer.py:794 -> bf16), the class / positional
embeddings are cast to that dtype (:799-801), LayerNorm computes in fp32 and casts back to its INPUT dtype (layers.py:23-26), F.linear returns bf16 and
`q_x + attention(...)` / `x + mlp(...)` (transformer.py:328-329) add two bf16 tensors: the image tower's residual stream and, through autograd, its
gradient are bf16.  The native default keeps that stream in fp32 (stricter); these tests cover the kernels and the model with the reference's dtype:
  * kernels against fp32 torch on the same (bf16-rounded) inputs: LayerNorm forward / backward reading bf16 x (and a bf16 residual gradient),
    the residual epilogue OCN_EPI_BIAS_RESID_BF16 (general kernel, persistent kernel incl. its half-tile tail round and ragged edges, bench shapes),
    the row gather / scatter-add on bf16 streams;
  * the model against the CPU oracle / the reference's golden vectors at small batches (tolerances below: the stream's own rounding is what the
    reference's policy costs, measured next to eager autocast), exactness of the execution variants against each other inside the mode;
  * the whole step at the bench's batch 4096 lives in tests/test_bench_size_gpu.py (same bound as the fp32 stream: every gradient <= 2e-2, medians
    not above eager autocast's).
"""
import numpy as np
import pytest
import torch

from open_clip_amd.configs import get_model_config
from open_clip_amd.synth import init_state_dict, synthetic_batch
from tests.test_kernels_gpu import _report, bf, check, rel_l2

pytestmark = pytest.mark.gpu

STREAMS = ("bf16", "bf16-fp32grad")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from open_clip_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _resid_bf16_ref(ref, bias, resid16):
    """the reference's arithmetic: F.linear's result rounded to bf16, then the bf16 residual add (rounded again)"""
    lin = (ref + bias).to(torch.bfloat16).float()
    return (lin + resid16.float())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (400, 384, 192), (37, 8, 64), (1300, 520, 128), (3000, 264, 256), (70000, 776, 384), (66000, 520, 256),
                                   (25444, 768, 512), (204800, 768, 768), (204800, 768, 3072)])
def test_gemm_nt_bf16_residual_epilogue(dev, M, N, K):
    """shapes: the general kernel (small), the persistent kernel with one / two K-tile pairs, ragged N inside a lane's 16-byte piece, several tiles per
    workgroup (the prefetch crosses tile boundaries), the half-tile tail round with a ragged last tile row, and the two image-tower shapes of the bench"""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M * 5 + N)
    a = bf(torch.randn(M, K, generator=g)).to(dev)
    b = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = bf(torch.randn(M, N, generator=g) * 3).to(dev)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.gemm_nt(ops.EPI_BIAS_RESID_BF16, a, b, out, bias=bias, resid=resid)
    ref = a.float() @ b.float().t()
    want = _resid_bf16_ref(ref, bias, resid)
    # two roundings: the linear's (|lin| * 2^-9) and the sum's
    got = out.float()
    assert torch.isfinite(got).all()
    bound = want.abs() * 2.0 ** -7 + (ref + bias).abs() * 2.0 ** -7 + 1e-6  # (a last-bit difference of the fp32 sums can flip either rounding)
    bad = int(((got - want).abs() > bound).sum())
    _report(f"gemm_nt[{M}x{N}x{K}] bf16 residual epilogue: rel_l2 {rel_l2(got, want):.3e} max_abs {float((got - want).abs().max()):.3e}, {bad} beyond the two-rounding bound")
    assert bad == 0
    # no bias, strided output / residual views (ldc > N)
    if N % 8 == 0 and M <= 70000:
        big_o = torch.full((M, N + 8), float("nan"), dtype=torch.bfloat16, device=dev)
        big_r = bf(torch.randn(M, N + 8, generator=g)).to(dev)
        ops.gemm_nt(ops.EPI_BIAS_RESID_BF16, a, b, big_o[:, :N], resid=big_r[:, :N])
        want2 = ref.to(torch.bfloat16).float() + big_r[:, :N].float()
        assert torch.isnan(big_o[:, N:].float()).all(), "wrote outside the view"
        assert int(((big_o[:, :N].float() - want2).abs() > want2.abs() * 2.0 ** -7 + ref.abs() * 2.0 ** -7 + 1e-6).sum()) == 0


@pytest.mark.parametrize("M,C", [(37, 128), (400, 768), (616, 512), (257, 1024), (100, 1280), (5, 192), (204800, 768)])
def test_layernorm_on_a_bf16_stream(dev, M, C):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x16 = bf(torch.randn(M, C, generator=g) * 2 + 0.5).to(dev)
    x = x16.float()
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    b = (0.1 * torch.randn(C, generator=g)).to(dev)
    y16, y32, mean, rstd = ops.layernorm_fwd(x16, w, b, want_bf16=True, want_f32=True)
    ref = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5)
    check(f"ln_fwd[{M}x{C}] bf16 x -> f32", y32, ref, rel=2e-6)
    check(f"ln_fwd[{M}x{C}] bf16 x -> bf16", y16, ref, bf16_out=True)
    check(f"ln_fwd[{M}x{C}] bf16 x mean", mean, x.mean(-1), rel=1e-5)
    dy = bf(torch.randn(M, C, generator=g)).to(dev)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5).backward(dy.float())
    for dres_dtype in (torch.bfloat16, torch.float32, None):
        dres = None if dres_dtype is None else torch.randn(M, C, generator=g).to(dev).to(dres_dtype)
        want = xr.grad + (0 if dres is None else dres.float())
        for want_f32 in (True, False):
            dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            dx32, dx16 = ops.layernorm_bwd(dy, x16, w, mean, rstd, dw, db, dres=dres, want_f32=want_f32, want_bf16=True)
            tag = f"ln_bwd[{M}x{C}] bf16 x, dres {dres_dtype}, f32 out {want_f32}"
            if want_f32:
                check(tag + " dx", dx32, want, rel=1e-5)
            else:
                assert dx32 is None
            check(tag + " dx16", dx16, want, bf16_out=True)
            check(tag + " dw", dw, wr.grad, rel=1e-4)
            check(tag + " db", db, br.grad, rel=1e-4)
    # the reproducible form reads the bf16 stream too
    dw1, db1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dw2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dres = bf(torch.randn(M, C, generator=g)).to(dev)
    _, a16 = ops.layernorm_bwd(dy, x16, w, mean, rstd, dw1, db1, dres=dres, want_f32=False, want_bf16=True, deterministic=True)
    _, b16 = ops.layernorm_bwd(dy, x16, w, mean, rstd, dw2, db2, dres=dres, want_f32=False, want_bf16=True, deterministic=True)
    assert torch.equal(a16, b16) and torch.equal(dw1, dw2) and torch.equal(db1, db2)
    check(f"ln_bwd[{M}x{C}] bf16 x deterministic dw", dw1, wr.grad, rel=1e-4)
    # combinations without an instantiation fail loudly
    with pytest.raises(RuntimeError, match="unsupported dtype combination"):
        ops.layernorm_bwd(dy.float(), x16, w, mean, rstd, dw1, db1, want_f32=True)
    with pytest.raises(RuntimeError, match="unsupported dtype combination"):
        ops.layernorm_bwd(dy, x, w, mean, rstd, dw1, db1, dres=dres, want_f32=True)


def test_row_gather_and_scatter_add_on_a_bf16_stream(dev):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(3)
    B, L, C = 37, 50, 768
    x16 = bf(torch.randn(B * L, C, generator=g)).to(dev)
    idx = torch.randint(0, L, (B,), generator=g, dtype=torch.int32).to(dev)
    got = ops.gather_rows(x16, idx, B, L)
    rows = torch.arange(B, device=dev) * L + idx.long()
    assert got.dtype == torch.float32 and torch.equal(got, x16[rows].float())
    got0 = ops.gather_rows(x16, None, B, L)
    assert torch.equal(got0, x16[torch.arange(B, device=dev) * L].float())
    absrows = rows.to(torch.int32)
    assert torch.equal(ops.gather_rows(x16, absrows, B, 0), x16[rows].float())
    d = torch.randn(B, C, generator=g).to(dev)
    dx16 = x16.clone()
    ops.scatter_add_rows(d, absrows, None, B, 0, dx16)
    want = x16.clone()
    want[rows] = (x16[rows].float() + d).to(torch.bfloat16)
    assert torch.equal(dx16, want)


# ---------------------------------------------------------------------------------------------------------------------------------------
# model level.  Tolerances of the bf16 image stream against the fp32 CPU oracle at SMALL batches: features / loss as everywhere (4e-3 / 2e-2);
# gradients rel-L2 <= 6e-2 (matrices, embeddings; measured <= 5.6e-2: the positional embedding at batch 8) / 7e-2 (1-D; measured <= 5.6e-2) -- the stream's rounding alone moves the oracle's own gradients by a median of
# 2.4e-2 / at worst 4.2e-2 at batch 8 (HISTORY.md, round 2: the fp32 oracle with its image stream rounded to bf16), on top of the 2.6e-2 / 3.9e-2
# of the fp32-stream path.  At the bench's batch the bound is the same 2e-2 as for the fp32 stream (tests/test_bench_size_gpu.py).
# ---------------------------------------------------------------------------------------------------------------------------------------
GRAD_TOL_MATRIX_BF16, GRAD_TOL_1D_BF16, GRAD_TOL_SMALL_BF16 = 6e-2, 7e-2, 0.15


def _tol(ref_norm, gmax, ndim):
    if ref_norm < 1e-3 * gmax:
        return GRAD_TOL_SMALL_BF16
    return GRAD_TOL_MATRIX_BF16 if ndim >= 2 else GRAD_TOL_1D_BF16


def _compare_to_oracle(tag, model, out, loss, outs, grads):
    from tests.test_model_gpu import FEAT_TOL, LOSS_TOL
    fi = float((out["image_features"].float().cpu() - outs["image_features"]).abs().max())
    ft = float((out["text_features"].float().cpu() - outs["text_features"]).abs().max())
    _report(f"{tag}: feat max_abs {fi:.3e}/{ft:.3e} loss {float(loss):.6f} vs {float(outs['loss']):.6f}")
    assert fi <= FEAT_TOL and ft <= FEAT_TOL, (fi, ft)
    assert abs(float(loss) - float(outs["loss"])) <= LOSS_TOL
    gmax = max(float(v.norm()) for v in grads.values())
    worst = []
    for k, p in model.named_parameters():
        ref = grads[k]
        rel = float((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        worst.append((rel / _tol(float(ref.norm()), gmax, ref.ndim), rel, k))
    worst.sort(reverse=True)
    for frac, rel, k in worst[:6]:
        _report(f"{tag}:   grad rel_l2={rel:.3e} ({frac:.2f} of its tolerance) {k}")
    assert worst[0][0] <= 1.0, worst[0]
    return worst


@pytest.mark.parametrize("stream", STREAMS)
@pytest.mark.parametrize("cfg_name,B,ckpt", [("small-test", 9, False), ("small-test", 16, True), ("ViT-B-32", 8, False)])
def test_bf16_image_stream_against_cpu_oracle(stream, cfg_name, B, ckpt):
    from oracle import clip_oracle as O
    from tests.test_model_gpu import _build, _step
    cfg = get_model_config(cfg_name)
    state = init_state_dict(cfg, seed=21, perturb=True)
    batch = synthetic_batch(cfg, B, seed=77)
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg)
    model = _build(cfg, state, image_stream=stream)
    model.set_grad_checkpointing(ckpt)
    out, loss = _step(model, batch)
    _compare_to_oracle(f"bf16-stream[{stream},{cfg_name},B{B},ckpt{int(ckpt)}]", model, out, loss, outs, grads)
    # the text tower is untouched by the switch: its features are those of the fp32-stream model, bit for bit
    ref_model = _build(cfg, state)
    ref_out, _ = _step(ref_model, batch)
    assert torch.equal(ref_out["text_features"], out["text_features"])


@pytest.mark.parametrize("stream", STREAMS)
def test_bf16_image_stream_execution_variants_agree(stream):
    """inside the mode: block recompute and the one-stream / two-stream towers give the same bits; the pooled last block and the full last block
    agree to the stream's resolution (the full block rounds the last block's rows to bf16, the pooled one keeps the B rows in fp32)"""
    from tests.test_model_gpu import _build, _step
    cfg = get_model_config("ViT-B-32")
    state = init_state_dict(cfg, seed=5, perturb=True)
    batch = synthetic_batch(cfg, 24, seed=9)
    base = _build(cfg, state, image_stream=stream)
    out0, loss0 = _step(base, batch)
    g0 = {k: p.grad.clone() for k, p in base.named_parameters()}
    for variant in ("recompute", "one_stream", "full_last_block", "all_queries"):
        m = _build(cfg, state, image_stream=stream)
        if variant == "recompute":
            m.set_grad_checkpointing(True)
        elif variant == "one_stream":
            m.tower_streams = False
        elif variant == "full_last_block":
            m.pooled_last_block = m.visual.pooled_last_block = False
        else:
            m.pooled_single_query = False
        out, loss = _step(m, batch)
        worst = max((rel_l2(p.grad, g0[k]), k) for k, p in m.named_parameters())
        df = float((out["image_features"].float() - out0["image_features"].float()).abs().max())
        _report(f"bf16-stream[{stream}] {variant} vs shipped: image features max |diff| {df:.2e}, loss {float(loss):.7f} vs {float(loss0):.7f}, worst gradient rel_l2 {worst[0]:.3e} ({worst[1]})")
        if variant in ("recompute", "one_stream"):
            assert df == 0.0 and worst[0] <= 2e-6, (variant, df, worst)
        else:
            assert df <= 2e-3 and worst[0] <= 5e-2, (variant, df, worst)
