"""Per-kernel parity tests on a real MI355X: every C-ABI entry point against a plain fp32 torch statement of the
same op on the same (bf16-rounded) inputs.  Tolerances are written next to each check:
  * fp32 outputs of bf16-operand GEMMs: rel-L2 <= 2e-5 (accumulation order only)
  * bf16 outputs: |err| <= 2^-8 * |ref| + tiny abs (one bf16 rounding)
  * atomically accumulated fp32 reductions: rel-L2 <= 1e-4
A human-readable report with the measured errors is appended to gpurun_out/parity_report.txt.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.txt")


def _report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    print(line)


def rel_l2(got, ref):
    got, ref = got.double().flatten(), ref.double().flatten()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def check(name, got, ref, rel=None, bf16_out=False, abs_tol=0.0):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    r = rel_l2(got, ref)
    mx = float((got - ref).abs().max())
    _report(f"{name:46s} rel_l2={r:.3e} max_abs={mx:.3e} ref_max={float(ref.abs().max()):.3e}")
    if bf16_out:
        bound = ref.abs() * 2.0 ** -7 + abs_tol + 1e-6 * float(ref.abs().max())
        bad = ((got - ref).abs() > bound).sum().item()
        assert bad == 0, f"{name}: {bad} elements beyond the bf16 rounding bound (rel_l2={r:.3e}, max_abs={mx:.3e})"
    if rel is not None:
        assert r <= rel, f"{name}: rel_l2 {r:.3e} > {rel:.1e}"


def check_saved_derivative(name, aux_u8, ref):
    """the 8-bit fixed-point gelu' of OCN_EPI_BIAS_GELU (csrc/ocn_common.h): decoded value within half a step (0.0025) of the exact
    derivative (+ 1e-4: the erf approximation and fp32 evaluation), and no systematic offset"""
    from open_clip_amd import ops
    assert aux_u8.dtype == torch.uint8 and int(aux_u8.max()) <= 252, f"{name}: byte outside the code range"
    err = ops.dgelu_decode(aux_u8) - ref.float()
    mx, mean = float(err.abs().max()), float(err.mean())
    _report(f"{name:46s} max_abs={mx:.3e} mean_err={mean:+.2e} rel_l2={rel_l2(ops.dgelu_decode(aux_u8), ref):.3e}")
    assert mx <= 2.6e-3, f"{name}: decoded derivative off by {mx:.3e} (> half a quantisation step)"
    assert abs(mean) <= 3e-4 or err.numel() < 20000, f"{name}: biased quantisation ({mean:+.2e})"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from open_clip_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16)


# ---------------------------------------------------------------------------------------------------
def test_probe_mfma_layout(dev):
    from open_clip_amd import _lib
    g = torch.Generator().manual_seed(1)
    a = bf(torch.randn(32, 16, generator=g)).to(dev)
    b = bf(torch.randn(32, 16, generator=g)).to(dev)  # [n, k]: asymmetric so a transposed C would be caught
    c = torch.zeros(32, 32, device=dev)
    _lib.call("ocn_probe_mfma32", a.data_ptr(), b.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
    check("probe_mfma32 C=A.B^T", c, a.float() @ b.float().t(), rel=1e-6)


def test_probe_tr16_layout(dev):
    from open_clip_amd import _lib
    src = bf(torch.arange(16 * 32, dtype=torch.float32).reshape(16, 32)).float()  # compare against the bf16-rounded table
    inp = bf(src).to(dev)
    out = torch.zeros(64, 4, dtype=torch.bfloat16, device=dev)
    _lib.call("ocn_probe_tr16", inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    exp = torch.zeros(64, 4)
    for lane in range(64):
        i, g, h = lane & 15, (lane >> 4) & 1, lane >> 5
        for j in range(4):
            exp[lane, j] = src[h * 8 + j, g * 16 + i]
    got = out.float().cpu()
    if not torch.equal(got, exp):
        _report("probe_tr16 MISMATCH; raw lane table (lane: 4 values):")
        for lane in range(64):
            _report(f"  lane {lane:2d}: {got[lane].tolist()}  expected {exp[lane].tolist()}")
    assert torch.equal(got, exp)


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (400, 384, 192), (1000, 2304, 768), (77 * 8, 512, 2048), (37, 6, 64), (4096, 4096, 512),
                                   (1300, 520, 128), (3000, 264, 256), (70000, 776, 384), (25444, 768, 512), (66000, 520, 256)])
def test_gemm_nt_epilogues(dev, M, N, K):
    """every epilogue against fp32 torch.  Shapes 7-9 are for the persistent kernel's epilogue-operand prefetch (round 4): K = 128 is a
    tile of ONE K-tile pair (the prefetch is issued in a tile's first phase), K = 256 two pairs, N % 16 == 8 puts the end of a row inside a lane's
    16-byte piece of the saved gelu', and 70000 x 776 gives every workgroup several tiles (the prefetch crosses tile boundaries; K = 384 is six
    K-tiles: no half-tile tail).  The last two are for the HALF-TILE tail round: 300 tiles = one round + 44 tail tiles split over 88 workgroups, the
    ragged last tile row (100 of 256 rows) among them; 774 tiles = three rounds + 6 tail tiles, K = 256 = exactly one pass of the half tile's
    four-K-tile loop"""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = bf(torch.randn(M, K, generator=g)).to(dev)
    b = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    ref = a.float() @ b.float().t()
    tag = f"gemm_nt[{M}x{N}x{K}]"
    out = ops.gemm_nt(ops.EPI_F32, a, b, torch.empty(M, N, device=dev), bias=bias, alpha=0.5)
    check(tag + " f32", out, 0.5 * ref + bias, rel=2e-5)
    out = ops.gemm_nt(ops.EPI_BF16, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev), bias=bias)
    check(tag + " bf16+bias", out, ref + bias, bf16_out=True)
    out = ops.gemm_nt(ops.EPI_BF16, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    check(tag + " bf16", out, ref, bf16_out=True)
    out = ops.gemm_nt(ops.EPI_BIAS_RESID_F32, a, b, torch.empty(M, N, device=dev), bias=bias, resid=resid)
    check(tag + " resid", out, ref + bias + resid, rel=2e-5)
    aux = torch.full((M, N), 255, dtype=torch.uint8, device=dev)
    out = ops.gemm_nt(ops.EPI_BIAS_GELU, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev), bias=bias, aux=aux)
    pre = (ref + bias).requires_grad_(True)
    act = torch.nn.functional.gelu(pre)
    act.backward(torch.ones_like(act))
    # the forward epilogue saves gelu'(pre-activation) (what the backward multiplies by), not the pre-activation itself, in 8-bit fixed
    # point: |error| <= half a step of 1/200 (+ the 2.5e-5 of the erf approximation)
    check_saved_derivative(tag + " gelu.saved_derivative", aux, pre.grad)
    check(tag + " gelu.out", out, act.detach(), bf16_out=True, abs_tol=1e-3)
    dsaved = torch.randint(0, 253, (M, N), generator=g, dtype=torch.uint8).to(dev)
    out = ops.gemm_nt(ops.EPI_DGELU, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev), aux=dsaved)
    check(tag + " dgelu", out, ref * ops.dgelu_decode(dsaved), bf16_out=True, abs_tol=1e-3)


@pytest.mark.parametrize("M,N,K", [(400, 384, 192), (2048, 1024, 256), (37, 6, 64), (5000, 520, 128)])
def test_gemm_nt_quickgelu_epilogue(dev, M, N, K):
    """OCN_EPI_BIAS_QUICKGELU (layers.py:29-32: x * sigmoid(1.702 x)): output and saved derivative against torch autograd, through the
    persistent and the general kernel; its backward is the shared OCN_EPI_DGELU multiply"""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = bf(torch.randn(M, K, generator=g)).to(dev)
    b = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    aux = torch.full((M, N), 255, dtype=torch.uint8, device=dev)
    out = ops.gemm_nt(ops.EPI_BIAS_QUICKGELU, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev), bias=bias, aux=aux)
    pre = (a.float() @ b.float().t() + bias).requires_grad_(True)
    act = pre * torch.sigmoid(1.702 * pre)
    act.backward(torch.ones_like(act))
    tag = f"gemm_nt[{M}x{N}x{K}] quickgelu"
    check(tag + ".out", out, act.detach(), bf16_out=True, abs_tol=1e-3)
    check_saved_derivative(tag + ".saved_derivative", aux, pre.grad)


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("M,N,K", [(300, 200, 64), (1000, 640, 320), (2500, 768, 1024), (513, 1027, 96), (37, 6, 32), (70000, 512, 256), (66000, 264, 128)])
def test_gemm_nt_every_kernel_variant(dev, variant, M, N, K):
    """both NT kernels forced (include/openclip_hip_debug.h: ocn_set_gemm_variant; 4 = the general ring kernel, 5 = the persistent
    kernel, which hands shapes it does not take to the general one) on ragged shapes, incl. a non-multiple-of-4 N (scalar epilogue
    path), K = 96 (3 ring stages) / K = 64 (2 stages) / K = 32, and many-tiles-per-workgroup shapes"""
    from open_clip_amd import _lib, ops
    g = torch.Generator().manual_seed(variant * 100 + M)
    a = bf(torch.randn(M, K, generator=g)).to(dev)
    b = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias, resid = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    fpre = torch.randint(0, 253, (M, N), generator=g, dtype=torch.uint8).to(dev)
    ref = a.float() @ b.float().t()
    try:
        _lib.call("ocn_set_gemm_variant", variant)
        tag = f"gemm_nt.v{variant}[{M}x{N}x{K}]"
        out = ops.gemm_nt(ops.EPI_F32, a, b, torch.empty(M, N, device=dev), bias=bias, alpha=2.0)
        check(tag + " f32", out, 2.0 * ref + bias, rel=2e-5)
        out = ops.gemm_nt(ops.EPI_BIAS_RESID_F32, a, b, torch.empty(M, N, device=dev), bias=bias, resid=resid)
        check(tag + " resid", out, ref + bias + resid, rel=2e-5)
        out = ops.gemm_nt(ops.EPI_DGELU, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev), aux=fpre)
        check(tag + " dgelu", out, ref * ops.dgelu_decode(fpre), bf16_out=True, abs_tol=1e-3)  # aux = the saved gelu'(pre-activation)
        aux = torch.full((M, N), 255, dtype=torch.uint8, device=dev)
        out = ops.gemm_nt(ops.EPI_BIAS_GELU, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev), bias=bias, aux=aux)
        pre = (ref + bias).requires_grad_(True)
        act = torch.nn.functional.gelu(pre)
        act.backward(torch.ones_like(act))
        check(tag + " gelu.out", out, act.detach(), bf16_out=True, abs_tol=1e-3)
        check_saved_derivative(tag + " gelu.saved_derivative", aux, pre.grad)
    finally:
        _lib.call("ocn_set_gemm_variant", 0)


@pytest.mark.parametrize("tv", [1, 3])  # 1 = general kernel, 3 = hand-scheduled
@pytest.mark.parametrize("M,N,K", [(3000, 640, 328), (100, 264, 520), (40000, 512, 256), (9 * 50, 768, 3072), (20011, 1536, 512)])
def test_gemm_tn_every_kernel_variant(dev, tv, M, N, K):
    from open_clip_amd import _lib, ops
    g = torch.Generator().manual_seed(tv * 10 + M)
    a = bf(torch.randn(M, N, generator=g)).to(dev)
    b = bf(torch.randn(M, K, generator=g)).to(dev)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    try:
        _lib.call("ocn_set_gemm_variant", tv << 4)
        ops.gemm_tn_accum(a, b, dw, db)
    finally:
        _lib.call("ocn_set_gemm_variant", 0)
    check(f"gemm_tn.v{tv}[{M}x{N}x{K}] dW", dw, a.float().t() @ b.float(), rel=1e-4)
    check(f"gemm_tn.v{tv}[{M}x{N}x{K}] dbias", db, a.float().sum(0), rel=1e-4)


@pytest.mark.parametrize("M,N,K", [(3000, 640, 328), (40000, 512, 256), (20011, 1536, 512), (204800, 768, 768), (50000, 2304, 768), (100, 264, 520)])
def test_gemm_tn_deterministic_form(dev, M, N, K):
    """ocn_gemm_tn_accum_det: per-split slabs summed in split order (or one split of the general kernel) -- equal to the atomic form up to
    summation order, accumulates INTO dW / dbias, and the same bits on every run"""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M + K)
    a = bf(torch.randn(M, N, generator=g)).to(dev)
    b = bf(torch.randn(M, K, generator=g)).to(dev)
    pre, preb = torch.randn(N, K, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    runs = []
    for _ in range(3):
        dw, db = pre.clone(), preb.clone()
        ops.gemm_tn_accum(a, b, dw, db, alpha=0.5, deterministic=True)
        runs.append((dw, db))
    ref_w, ref_b = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    for r0 in range(0, M, 32768):  # fp32 reference in row chunks
        ref_w += a[r0:r0 + 32768].float().t() @ b[r0:r0 + 32768].float()
        ref_b += a[r0:r0 + 32768].float().sum(0)
    check(f"gemm_tn.det[{M}x{N}x{K}] dW", runs[0][0], pre + 0.5 * ref_w, rel=1e-4)
    check(f"gemm_tn.det[{M}x{N}x{K}] dbias", runs[0][1], preb + 0.5 * ref_b, rel=1e-4)
    for dw, db in runs[1:]:
        assert torch.equal(dw, runs[0][0]) and torch.equal(db, runs[0][1]), "deterministic wgrad changed between runs"


def test_gemm_nt_strided_views(dev):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(5)
    abig = bf(torch.randn(300, 256, generator=g)).to(dev)
    a = abig[:, 64:192]  # lda 256, K 128
    b = bf(torch.randn(96, 128, generator=g)).to(dev)
    obig = torch.zeros(300, 128, device=dev)
    out = obig[:, :96]
    ops.gemm_nt(ops.EPI_F32, a, b, out)
    check("gemm_nt strided", out, a.float() @ b.float().t(), rel=2e-5)
    assert float(obig[:, 96:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (400, 384, 128), (1000, 768, 2304), (5000, 64, 136), (33, 8, 8), (8 * 77, 1536, 512)])
def test_gemm_tn_accum(dev, M, N, K):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, N, generator=g)).to(dev)
    b = bf(torch.randn(M, K, generator=g)).to(dev)
    base = torch.randn(N, K, generator=g).to(dev)
    dw = base.clone()
    db = torch.zeros(N, device=dev)
    ops.gemm_tn_accum(a, b, dw, db, alpha=0.25)
    check(f"gemm_tn[{M}x{N}x{K}] dW", dw, base + 0.25 * (a.float().t() @ b.float()), rel=1e-4)
    check(f"gemm_tn[{M}x{N}x{K}] dbias", db, 0.25 * a.float().sum(0), rel=1e-4)


def test_cast_and_transpose(dev):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(300, 130, generator=g).to(dev)
    assert torch.equal(ops.cast_bf16(w), w.to(torch.bfloat16))
    assert torch.equal(ops.cast_transpose_bf16(w), w.t().contiguous().to(torch.bfloat16))
    v = torch.randn(1027, generator=g).to(dev)
    assert torch.equal(ops.cast_bf16(v), v.to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,C", [(37, 128), (400, 768), (616, 512), (257, 1024), (100, 1280), (5, 192)])
def test_layernorm(dev, M, C):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M * C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(dev)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    b = (0.1 * torch.randn(C, generator=g)).to(dev)
    y16, y32, mean, rstd = ops.layernorm_fwd(x, w, b, want_bf16=True, want_f32=True)
    ref = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5)
    check(f"ln_fwd[{M}x{C}] f32", y32, ref, rel=2e-6)
    check(f"ln_fwd[{M}x{C}] bf16", y16, ref, bf16_out=True)
    check(f"ln_fwd[{M}x{C}] mean", mean, x.mean(-1), rel=1e-5)
    for dy_dtype in (torch.bfloat16, torch.float32):
        dy = torch.randn(M, C, generator=g).to(dev).to(dy_dtype)
        dres = torch.randn(M, C, generator=g).to(dev)
        xr = x.clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5).backward(dy.float())
        dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx32, dx16 = ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres, want_f32=True, want_bf16=True)
        check(f"ln_bwd[{M}x{C},{dy_dtype}] dx", dx32, xr.grad + dres, rel=1e-5)
        check(f"ln_bwd[{M}x{C},{dy_dtype}] dx16", dx16, xr.grad + dres, bf16_out=True)
        check(f"ln_bwd[{M}x{C},{dy_dtype}] dw", dw, wr.grad, rel=1e-4)
        check(f"ln_bwd[{M}x{C},{dy_dtype}] db", db, br.grad, rel=1e-4)


# ---------------------------------------------------------------------------------------------------
def _attn_ref(qkv, B, L, H, causal, D=64):
    C = H * D
    q, k, v = qkv.float().reshape(B, L, 3, H, D).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * D ** -0.5
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=qkv.device).triu_(1)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * L, C)
    lse = torch.logsumexp(s, dim=-1)  # [B,H,L]
    return o, lse.reshape(-1)


@pytest.mark.parametrize("B,L,H,causal", [(3, 50, 2, False), (2, 77, 3, True), (5, 5, 2, False), (2, 16, 2, True), (1, 257, 2, False), (4, 128, 1, True)])
def test_attention(dev, B, L, H, causal):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(B * L + H)
    C = H * 64
    qkv = bf(torch.randn(B * L, 3 * C, generator=g) * 1.5).to(dev)
    out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125)
    x = qkv.float().requires_grad_(True)
    ref, ref_lse = _attn_ref(x, B, L, H, causal)
    tag = f"attn[B{B} L{L} H{H} c{int(causal)}]"
    check(tag + " out", out, ref.detach(), rel=6e-3)       # P is rounded to bf16 before P.V
    check(tag + " lse", lse, ref_lse.detach(), rel=1e-5)
    dout = bf(torch.randn(B * L, C, generator=g)).to(dev)
    ref.backward(dout.float())
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125)
    check(tag + " dqkv", dqkv, x.grad, rel=1.5e-2)          # P, dS rounded to bf16; delta from bf16 O
    for i, nm in enumerate("qkv"):
        check(tag + f" d{nm}", dqkv[:, i * C:(i + 1) * C], x.grad[:, i * C:(i + 1) * C], rel=2e-2)


@pytest.mark.parametrize("mode", ["image_cls", "text_packed", "text_dense", "long"])
def test_attention_pooled_single_query(dev, mode):
    """ocn_attn_pooled_fwd / ocn_attn_pooled_bwd (csrc/attention_pooled.hip): ONE query per (sequence, head) -- the pooled row of a tower's last
    block (transformer.py:829-831, :941-944) -- against fp32 torch softmax attention restricted to that query: image tower (query = row 0 of every
    sequence, no mask), packed text rows (query = a sequence's last row, ragged lengths), dense causal text (query = the EOT row in the MIDDLE of a
    padded sequence: keys behind it are masked and their dK / dV rows must come back as zeros), and a 257-token sequence (17 key chunks)"""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(len(mode))
    B, H = 6, 3
    C = H * 64
    if mode == "image_cls":
        L, lens, causal = 50, [50] * B, False
        starts = [b * L for b in range(B)]
        rows = [b * L for b in range(B)]
        seq_off = None
    elif mode == "long":
        L, lens, causal = 257, [257] * B, False
        starts = [b * L for b in range(B)]
        rows = [b * L for b in range(B)]
        seq_off = None
    elif mode == "text_packed":
        L, lens, causal = 77, [77, 9, 33, 1, 64, 17], True
        starts = [sum(lens[:b]) for b in range(B)]
        rows = [starts[b] + lens[b] - 1 for b in range(B)]
        seq_off = torch.tensor(starts + [sum(lens)], dtype=torch.int32, device=dev)
    else:
        L, lens, causal = 77, [77] * B, True
        starts = [b * L for b in range(B)]
        rows = [b * L + e for b, e in enumerate([76, 8, 40, 0, 63, 31])]
        seq_off = None
    M = sum(lens)
    kv = bf(torch.randn(M, 2 * C, generator=g) * 1.5).to(dev)
    q = bf(torch.randn(B, C, generator=g) * 1.5).to(dev)
    dout = bf(torch.randn(B, C, generator=g)).to(dev)
    rows_t = torch.tensor(rows, dtype=torch.int32, device=dev)
    out, lse = ops.attn_pooled_fwd(q, kv, rows_t, B, L, H, causal, 0.125, seq_off)
    dkv_nan = None
    dq, dkv = ops.attn_pooled_bwd(q, kv, out, dout, lse, rows_t, B, L, H, causal, 0.125, seq_off)
    qf, kvf = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref_out, ref_lse = [], []
    for b in range(B):
        end = rows[b] + 1 if causal else starts[b] + lens[b]
        k = kvf[starts[b]:end, :C].reshape(-1, H, 64).permute(1, 0, 2)      # [H, n, 64]
        v = kvf[starts[b]:end, C:].reshape(-1, H, 64).permute(1, 0, 2)
        s = (k @ qf[b].reshape(H, 64, 1)).squeeze(-1) * 0.125               # [H, n]
        ref_lse.append(torch.logsumexp(s, dim=-1))
        ref_out.append((torch.softmax(s, dim=-1).unsqueeze(1) @ v).reshape(C))
    ref_out, ref_lse = torch.stack(ref_out), torch.stack(ref_lse).reshape(-1)
    ref_out.backward(dout.float())
    tag = f"attn_pooled[{mode}]"
    check(tag + " out", out, ref_out.detach(), rel=4e-3)
    check(tag + " lse", lse, ref_lse.detach(), rel=1e-5)
    check(tag + " dq", dq, qf.grad, rel=1e-2)
    check(tag + " dkv", dkv, kvf.grad, rel=1e-2)
    assert torch.isfinite(dkv.float()).all()
    if mode == "text_dense":  # rows behind the pooled token: exact zeros, written by the kernel (dkv came from torch.empty)
        for b in range(B):
            assert float(dkv[rows[b] + 1:(b + 1) * L].float().abs().max() if rows[b] + 1 < (b + 1) * L else 0.0) == 0.0


@pytest.mark.parametrize("L,causal,packed", [(50, False, False), (77, True, False), (77, True, True), (100, False, False)])
def test_attention_backward_padded_rows_stay_finite(dev, L, causal, packed):
    """padded rows of the head-resident kernels (the tail of a sequence's last 32-row block) must never reach a result.  Adversarial case:
    the LAST row's scores are all ~ -240 (its saved LSE is ~ -236): P of a padded key / query evaluated against that LSE is exp(+236) = inf,
    and an implementation that annihilates padded entries by a zero factor instead of a predicate would produce inf x 0 = NaN for every
    gradient of the head (the mask-free backward tried in round 3 needed a clamp on P for exactly this case)."""
    from open_clip_amd import ops
    B, H = 3, 2
    C = H * 64
    g = torch.Generator().manual_seed(L)
    lens = [L, L - 7, max(3, L - 40)] if packed else [L] * B
    rows = sum(lens)
    qkv = bf(torch.randn(rows, 3 * C, generator=g) * 1.5)
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    for b in range(B):  # last query of every sequence: q = 30, all keys get a -1 component sum -> S = -1920 * 0.125
        qkv[off[b + 1] - 1, 0:C] = 30.0
        qkv[off[b]:off[b + 1], C:2 * C] = -1.0
    qkv = bf(qkv).to(dev)
    dout = bf(torch.randn(rows, C, generator=g)).to(dev)
    lay = None
    if packed:
        so = torch.tensor(off, dtype=torch.int32, device=dev)
        nb = torch.tensor([(n + 31) // 32 for n in lens])
        lay = ops.SeqLayout(so, torch.sort(nb, stable=True).indices.to(torch.int32).to(dev), torch.bincount(nb - 1, minlength=(L + 31) // 32).tolist())
    out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125, seq_off=lay)
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125, seq_off=lay)
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
    # reference per sequence (fp32)
    for b in range(B):
        x = qkv[off[b]:off[b + 1]].float().requires_grad_(True)
        ref, _ = _attn_ref(x, 1, lens[b], H, causal)
        ref.backward(dout[off[b]:off[b + 1]].float())
        tag = f"attn_padded[L{L} c{int(causal)} p{int(packed)} seq{b}]"
        check(tag + " out", out[off[b]:off[b + 1]], ref.detach(), rel=6e-3)
        check(tag + " dqkv", dqkv[off[b]:off[b + 1]], x.grad, rel=2e-2)


@pytest.mark.parametrize("B,L,H,D,causal,force", [(2, 257, 2, 80, False, False), (3, 50, 3, 80, False, False), (2, 77, 2, 80, True, False),
                                                 (1, 257, 1, 128, False, False), (2, 40, 2, 96, True, False), (1, 400, 2, 64, False, False),
                                                 (2, 400, 1, 64, True, False), (3, 50, 2, 64, False, True), (2, 77, 3, 64, True, True), (4, 7, 2, 80, True, False),
                                                 (2, 257, 2, 64, False, False), (1, 200, 2, 64, True, False), (3, 257, 2, 64, False, True),
                                                 (2, 257, 2, 88, False, False), (2, 77, 3, 88, True, False), (1, 257, 2, 104, False, False),
                                                 (3, 50, 1, 104, True, False), (2, 257, 1, 112, False, False), (2, 40, 2, 112, True, False),
                                                 (1, 730, 2, 80, False, False), (3, 197, 6, 64, False, False), (3, 197, 2, 80, False, False)])
def test_attention_generic_head_dims(dev, B, L, H, D, causal, force):
    """streamed kernels (csrc/attention_generic.hip: K / V resp. Q / dO in 64-row chunks through a two-slot LDS ring): head_dim 80 / 96 / 128
    (ViT-H-14: 80 x 257 tokens; 730 tokens at 378 px), 88 / 104 (ViT-g-14 / ViT-bigG-14: contraction zero-padded to 96 / 112 -- the random
    columns of the neighbouring heads must not leak in) and 112 (ViT-e-14), head_dim 64 beyond 320 tokens, the MIXED default dispatch of head_dim 64 between 129 and 320 tokens
    (streamed forward, head-resident backward sharing its LSE: ViT-L-14's 257) and -- forced through developer knob 7 -- the head_dim-64
    shapes of the specialised kernels, which must agree with them.  Same tolerances as test_attention."""
    from open_clip_amd import _lib, ops
    g = torch.Generator().manual_seed(B * L + H + D)
    C = H * D
    scale = D ** -0.5
    qkv = bf(torch.randn(B * L, 3 * C, generator=g) * 1.5).to(dev)
    dout = bf(torch.randn(B * L, C, generator=g)).to(dev)
    if force:
        out0, lse0 = ops.attn_fwd(qkv, B, L, H, causal, scale, D)
        d0 = ops.attn_bwd(qkv, out0, dout, lse0, B, L, H, causal, scale, D)
        _lib.call("ocn_set_tuning", 7, 1)
    try:
        out, lse = ops.attn_fwd(qkv, B, L, H, causal, scale, D)
        dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, scale, D)
    finally:
        _lib.call("ocn_set_tuning", 7, 0)
    x = qkv.float().requires_grad_(True)
    ref, ref_lse = _attn_ref(x, B, L, H, causal, D)
    tag = f"attn_generic[B{B} L{L} H{H} D{D} c{int(causal)}]"
    check(tag + " out", out, ref.detach(), rel=6e-3)
    check(tag + " lse", lse, ref_lse.detach(), rel=1e-5)
    ref.backward(dout.float())
    check(tag + " dqkv", dqkv, x.grad, rel=1.5e-2)
    for i, nm in enumerate("qkv"):
        check(tag + f" d{nm}", dqkv[:, i * C:(i + 1) * C], x.grad[:, i * C:(i + 1) * C], rel=2e-2)
    if force:
        check(tag + " out vs specialised kernel", out, out0, rel=2e-3)
        check(tag + " dqkv vs specialised kernel", dqkv, d0, rel=6e-3)


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,P,width", [(3, 64, 32, 128), (2, 224, 32, 768), (2, 28, 14, 128)])
def test_vision_embed_kernels(dev, B, H, P, width):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(B + H)
    img = torch.randn(B, 3, H, H, generator=g).to(dev)
    KP = 3 * P * P
    Kpad = (KP + 63) // 64 * 64
    G = (H // P) ** 2
    patches = ops.patchify(img, P, Kpad)
    ref = torch.nn.functional.unfold(img, P, stride=P).transpose(1, 2).reshape(B * G, KP)
    assert torch.equal(patches[:, :KP], ref.to(torch.bfloat16))
    assert float(patches[:, KP:].float().abs().sum()) == 0.0
    po = torch.randn(B * G, width, generator=g).to(dev)
    cls, pos = torch.randn(width, generator=g).to(dev), torch.randn(G + 1, width, generator=g).to(dev)
    emb = ops.embed_assemble_fwd(po, cls, pos, B, G, width)
    ref = torch.cat([cls.expand(B, 1, width), po.reshape(B, G, width)], 1) + pos
    assert torch.equal(emb, ref.reshape(B * (G + 1), width))
    demb = torch.randn(B * (G + 1), width, generator=g).to(dev)
    dpos, dcls = torch.zeros_like(pos), torch.zeros_like(cls)
    dpatch = ops.embed_assemble_bwd(demb, dpos, dcls, B, G, width)
    d3 = demb.reshape(B, G + 1, width)
    check("embed_bwd dpos", dpos, d3.sum(0), rel=1e-5)
    check("embed_bwd dcls", dcls, d3[:, 0].sum(0), rel=1e-5)
    assert torch.equal(dpatch, d3[:, 1:].reshape(B * G, width).to(torch.bfloat16))


@pytest.mark.parametrize("H,P", [(64, 32), (28, 14), (48, 16), (224, 32)])
def test_patchify_uint8_input_fuses_totensor_normalize(dev, H, P):
    """8f-4: uint8 pixels in either layout -> the same bf16 patch matrix as ToTensor + Normalize (constants.py:1-2)
    followed by the float path (one bf16 rounding of a value computed with a fused multiply-add: <= 1 bf16 ulp apart)."""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(H)
    B = 3
    u8 = torch.randint(0, 256, (B, 3, H, H), generator=g, dtype=torch.uint8)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    f = (u8.float() / 255.0 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    KP = 3 * P * P
    Kpad = (KP + 63) // 64 * 64
    ref = ops.patchify(f.to(dev), P, Kpad).float()
    for hwc, src in ((False, u8), (True, u8.permute(0, 2, 3, 1).contiguous())):
        got = ops.patchify_u8(src.to(dev), P, Kpad, mean, std, hwc).float()
        assert float(got[:, KP:].abs().sum()) == 0.0
        err = (got - ref).abs()
        assert float((err / ref.abs().clamp_min(1e-3)).max()) <= 2.0 ** -7, (hwc, float(err.max()))
        assert float((got != ref).float().mean()) < 0.02  # almost always the identical bf16 value


def test_text_embed_and_pool_kernels(dev):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(11)
    B, L, C, V = 37, 77, 192, 1000
    text = torch.randint(0, V, (B, L), generator=g)
    text[:, 5] = V - 1
    text[3, 2] = V - 1  # tie: the first index must win
    text = text.to(dev)
    table, pos = torch.randn(V, C, generator=g).to(dev), torch.randn(L, C, generator=g).to(dev)
    x = ops.token_embed_fwd(text, table, pos)
    assert torch.equal(x, (table[text] + pos).reshape(B * L, C))
    idx = ops.argmax_rows(text)
    assert torch.equal(idx.long(), text.argmax(-1))
    dx = torch.randn(B * L, C, generator=g).to(dev)
    dtable, dpos = torch.zeros_like(table), torch.zeros_like(pos)
    ops.token_embed_bwd(text, dx, dtable, dpos)
    ref = torch.zeros_like(table).index_add_(0, text.reshape(-1), dx)
    check("token_embed_bwd dtable", dtable, ref, rel=1e-5)
    check("token_embed_bwd dpos", dpos, dx.reshape(B, L, C).sum(0), rel=1e-5)
    dtable2, dpos2 = torch.zeros_like(table), torch.zeros_like(pos)
    ops.token_embed_bwd_sorted(text, dx, dtable2, dpos2)
    check("token_embed_bwd_sorted dtable", dtable2, ref, rel=1e-5)
    check("token_embed_bwd_sorted dpos", dpos2, dx.reshape(B, L, C).sum(0), rel=1e-5)
    pooled = ops.gather_rows(x, idx, B, L)
    assert torch.equal(pooled, x.reshape(B, L, C)[torch.arange(B), idx.long()])
    assert torch.equal(ops.gather_rows(x, None, B, L), x.reshape(B, L, C)[:, 0])
    dfull = torch.zeros(B * L, C, device=dev)
    ops.scatter_rows(pooled, idx, dfull, B, L)
    assert torch.equal(dfull.reshape(B, L, C)[torch.arange(B), idx.long()], pooled) and float(dfull.abs().sum()) == float(pooled.abs().sum())
    # the bf16 gather and the absolute-row (L = 0) forms that the pooled last block uses (model.py::_PooledBlockFn)
    rows = (torch.arange(B, device=dev) * L + idx.long()).to(torch.int32)
    x16 = x.bfloat16()
    assert torch.equal(ops.gather_rows_bf16(x16, rows, B, 0), x16[rows.long()])
    assert torch.equal(ops.gather_rows_bf16(x16, idx, B, L), x16[rows.long()])
    assert torch.equal(ops.gather_rows(x, rows, B, 0), pooled)
    d16 = torch.zeros(B * L, C, device=dev, dtype=torch.bfloat16)
    ops.scatter_rows(pooled, rows, None, B, 0, d16)  # bf16 target only
    untouched = torch.ones(B * L, dtype=torch.bool, device=dev)
    untouched[rows.long()] = False
    assert torch.equal(d16[rows.long()], pooled.bfloat16()) and float(d16[untouched].float().abs().max()) == 0.0
    y, y16, inv = ops.l2norm_fwd(pooled)
    check("l2norm_fwd", y, torch.nn.functional.normalize(pooled, dim=-1), rel=1e-6)
    pr = pooled.clone().requires_grad_(True)
    dy = torch.randn(B, C, generator=g).to(dev)
    torch.nn.functional.normalize(pr, dim=-1).backward(dy)
    check("l2norm_bwd", ops.l2norm_bwd(dy, y, inv), pr.grad, rel=1e-5)


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,N,off", [(6, 6, 0), (64, 200, 64), (300, 300, 0)])
def test_loss_row_kernels(dev, R, N, off):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(R + N)
    logits = (torch.randn(R, N, generator=g) * 4).to(dev)
    ldg = (N + 63) // 64 * 64
    G = torch.zeros(R, ldg, dtype=torch.bfloat16, device=dev)
    acc = torch.zeros(2, device=dev)
    s = 14.3
    ops.softmax_ce_rows(logits, G, N, off, 0.5 / R, 0.5 / R, 1.0 / s, acc[0:1], acc[1:2])
    lr = logits.clone().requires_grad_(True)
    labels = torch.arange(R, device=dev) + off
    loss = 0.5 * torch.nn.functional.cross_entropy(lr, labels)
    loss.backward()
    check(f"ce_rows[{R}x{N}] loss", acc[0:1], loss.detach().reshape(1), rel=1e-5)
    # G = softmax * grad_scale: the -onehot * grad_scale part of the gradient is the caller's (applied exactly, loss.py::_PairTerm.dX / dY)
    g_soft = lr.grad.clone()
    g_soft[torch.arange(R), labels] += 0.5 / R
    check(f"ce_rows[{R}x{N}] G", G[:, :N], g_soft, rel=4e-3)
    check(f"ce_rows[{R}x{N}] dscale", acc[1:2], ((lr.grad * logits).sum() / s).reshape(1), rel=1e-4)
    for neg in (0, 1):
        G.zero_()
        acc3 = torch.zeros(3, device=dev)
        bias = -3.0
        ops.siglip_rows(logits, G, N, off, neg, bias, 1.0 / R, 1.0 / R, 1.0 / s, acc3[0:1], acc3[1:2], acc3[2:3])
        lr = logits.clone().requires_grad_(True)
        lab = -torch.ones(R, N, device=dev)
        if not neg:
            lab[torch.arange(R), labels] = 1.0
        loss = -torch.nn.functional.logsigmoid(lab * lr).sum() / R
        loss.backward()
        check(f"siglip_rows[{R}x{N},neg{neg}] loss", acc3[0:1], loss.detach().reshape(1), rel=1e-5)
        g_sig = lr.grad.clone()  # G = sigmoid * grad_scale: the positives' -grad_scale is the caller's (loss.py::_PairTerm.dX / dY)
        if not neg:
            g_sig[torch.arange(R), labels] += 1.0 / R
        check(f"siglip_rows[{R}x{N},neg{neg}] G", G[:, :N], g_sig, rel=4e-3)
        check(f"siglip_rows[{R}x{N},neg{neg}] dscale", acc3[1:2], ((lr.grad * (logits - bias)).sum() / s).reshape(1), rel=1e-4)
        check(f"siglip_rows[{R}x{N},neg{neg}] dbias", acc3[2:3], lr.grad.sum().reshape(1), rel=1e-4)


def test_adamw_and_sumsq(dev):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(2)
    n = 100003
    w0, gr = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.AdamW([p], lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
    w, m, v = w0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    w16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for step in (1, 2, 3):
        p.grad = gr.clone()
        opt.step()
        ops.adamw_step(w, gr, m, v, 5e-4, 0.9, 0.98, 1e-6, 0.2, step, w_bf16=w16)
    check("adamw 3 steps", w, p.detach(), rel=1e-6)
    assert torch.equal(w16, w.to(torch.bfloat16))
    out = torch.zeros(1, device=dev)
    ops.sumsq_accum(gr, out)
    check("sumsq", out, (gr.double() ** 2).sum().float().reshape(1), rel=1e-5)


def test_adamw_multi_fused_clip_and_operand_copies(dev):
    """8f-1: one launch = clip_grad_norm_ + AdamW over every tensor + refresh of the bf16 GEMM-operand copies, against
    torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (fp32; rel-L2 <= 1e-6, copies bit-equal to a cast of the result)."""
    from open_clip_amd.model import _WeightCache
    from open_clip_amd.optim import NativeAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(128, 192), (64, 64), (768,), (), (50, 768), (100003,), (192, 3, 8, 8), (24577,)]
    ws = [torch.randn(sh, generator=g).to(dev) for sh in shapes]
    P = [torch.nn.Parameter(w.clone()) for w in ws]
    R = [torch.nn.Parameter(w.clone()) for w in ws]
    wd_of = lambda p: 0.0 if p.ndim <= 1 else 0.2
    ref = torch.optim.AdamW([{"params": [r], "weight_decay": wd_of(r), "lr": 5e-4 * (1 + i % 2)} for i, r in enumerate(R)],
                            lr=5e-4, betas=(0.9, 0.98), eps=1e-6)
    cache = _WeightCache()
    for p in P:
        if p.ndim in (2, 4):
            cache.get(p, "n"), cache.get(p, "t")
    opt = NativeAdamW([{"params": [p], "weight_decay": wd_of(p), "lr": 5e-4 * (1 + i % 2)} for i, p in enumerate(P)],
                      lr=5e-4, betas=(0.9, 0.98), eps=1e-6, grad_clip_norm=0.7, weight_caches=[cache])
    held = {id(p): (cache.peek(p, "n"), cache.peek(p, "t")) for p in P}
    for step in range(3):
        for i, (p, r) in enumerate(zip(P, R)):
            gr = torch.randn(p.shape, generator=g).to(dev) * (0.1 + step)
            r.grad = gr.clone()
            if i in (0, 5):  # gradient views at a 4-byte offset (DDP bucket views): scalar path / unaligned tile path
                buf = torch.empty(gr.numel() + 1, device=dev)
                buf[1:].copy_(gr.flatten())
                p.grad = buf[1:].view(p.shape)
            else:
                p.grad = gr.clone()
        total = torch.nn.utils.clip_grad_norm_(R, 0.7)
        ref.step()
        opt.step()
        check(f"adamw_multi grad-norm step {step}", opt.last_grad_norm_sq.sqrt(), total.reshape(1), rel=1e-5)
    for i, (p, r) in enumerate(zip(P, R)):
        check(f"adamw_multi param {tuple(p.shape)}", p.detach().reshape(-1), r.detach().reshape(-1), rel=1e-6)
        if p.ndim in (2, 4):
            n16, t16 = cache.get(p, "n"), cache.get(p, "t")
            assert n16 is held[id(p)][0], "the optimizer must refresh the cached copy in place"
            tileable = p.shape[0] % 64 == 0 and (p.numel() // p.shape[0]) % 64 == 0
            assert (t16 is held[id(p)][1]) == tileable, "transposed copies of 64-divisible weights are refreshed in place, others re-cast"
            w2 = p.detach().reshape(p.shape[0], -1)
            assert torch.equal(n16, w2.to(torch.bfloat16)), tuple(p.shape)
            assert torch.equal(t16, w2.t().contiguous().to(torch.bfloat16)), tuple(p.shape)
    # the per-tensor path gives the same numbers
    P2 = [torch.nn.Parameter(w.clone()) for w in ws[:4]]
    o2 = NativeAdamW(P2, lr=5e-4, fused=False)
    o3 = NativeAdamW([torch.nn.Parameter(w.clone()) for w in ws[:4]], lr=5e-4)
    for a, b in zip(o2.param_groups[0]["params"], o3.param_groups[0]["params"]):
        a.grad = torch.ones_like(a) * 0.3
        b.grad = torch.ones_like(b) * 0.3
    o2.step(), o3.step()
    for a, b in zip(o2.param_groups[0]["params"], o3.param_groups[0]["params"]):
        check("adamw fused vs per-tensor", b.detach().reshape(-1), a.detach().reshape(-1), rel=1e-7)


def test_ops_fail_loudly_on_cpu_tensors(dev):
    from open_clip_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.cast_bf16(torch.randn(8))
    with pytest.raises(RuntimeError, match="K=.*multiple of 32"):
        a = torch.zeros(8, 48, dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(ops.EPI_F32, a, a, torch.zeros(8, 8, device=dev))


@pytest.mark.parametrize("R,N,E,off,scale", [(4096, 32768, 512, 3 * 4096, 14.2857), (4096, 4096, 512, 0, 14.2857), (300, 1000, 128, 256, 14.2857), (2048, 16384, 768, 2048, 100.0),
                                             (4096, 4096, 512, 0, 100.0), (512, 2048, 256, 1024, 400.0)])
def test_fused_logits_cross_entropy(dev, R, N, E, off, scale):
    """ocn_fused_logits_ce (no materialised logits, ONE pass of the GEMM since round 6; the row-sharded global loss of config 3 is [4096 x 32768 x 512])
    against fp32 torch on the same bf16 operands: loss within 1e-5 relative; the softmax part of the gradient, G * rowscale[:, None], within one bf16
    rounding of softmax * grad_scale (+2e-3 of the largest value in absolute terms); sum((softmax - onehot) * grad_scale * logits) within 2e-4 relative.
    Scale 100 = the clamp of the reference's recipe (image_text_task.py:91-101); scale 400 with the label pair ANTI-aligned in a quarter of the rows drives
    those rows out of the shifted sums' range and through the exact fix-up kernel."""
    from open_clip_amd import ops
    g = torch.Generator(device=dev).manual_seed(R + N)
    x = torch.nn.functional.normalize(torch.randn(R, E, device=dev, generator=g), dim=-1)
    y = torch.nn.functional.normalize(torch.randn(N, E, device=dev, generator=g), dim=-1)
    if scale > 100:  # rows 0, 4, 8, ...: the label is the opposite of the image feature (logit -scale), every other column near 0: all under the shift's range
        y[off:off + R:4] = -x[::4]
    xs16, y16 = bf(x * scale), bf(y)
    ldg = (N + 63) // 64 * 64
    G = torch.zeros(R, ldg, dtype=torch.bfloat16, device=dev)
    acc = torch.zeros(2, device=dev)
    ls, gs = 0.5 / R, 0.5 / R
    assert ops.fused_logits_ce_supported(R, N, E)
    rowscale = ops.fused_logits_ce(xs16, y16, G, N, off, ls, gs, acc[0:1], acc[1:2])
    torch.cuda.synchronize()
    assert rowscale.shape == (R,) and bool(torch.isfinite(rowscale).all()) and bool((rowscale > 0).all())
    loss_ref, ds_ref, worst, bad, gmax = 0.0, 0.0, 0.0, 0, 0.0
    for r0 in range(0, R, 512):
        sl = slice(r0, min(R, r0 + 512))
        lg = xs16[sl].float() @ y16.float().t()
        lab = torch.arange(sl.start, sl.stop, device=dev) + off
        lse = torch.logsumexp(lg, -1)
        loss_ref += float(((lse - lg[torch.arange(lg.shape[0]), lab]) * ls).sum())
        Gref = torch.softmax(lg, -1)
        Gfull = Gref.clone()
        Gfull[torch.arange(lg.shape[0]), lab] -= 1
        ds_ref += float((Gfull * gs * lg).sum())  # sum((softmax - onehot) * grad_scale * logits): the whole gradient
        Gref *= gs                                 # G * rowscale holds softmax * grad_scale only (the onehot part is the caller's, exact)
        got = G[sl, :N].float() * rowscale[sl, None]
        gmax = max(gmax, float(Gref.abs().max()))
        bad += int(((got - Gref).abs() > Gref.abs() * 2.0 ** -7 + 2e-3 * float(Gref.abs().max())).sum())
        worst = max(worst, rel_l2(got, Gref))
    _report(f"fused_logits_ce[{R}x{N}x{E}, scale {scale:g}] loss {float(acc[0]):.6f} ref {loss_ref:.6f} dscale {float(acc[1]):.6e} ref {ds_ref:.6e} G rel_l2 {worst:.3e}")
    assert abs(float(acc[0]) - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-6
    assert abs(float(acc[1]) - ds_ref) <= 2e-4 * abs(ds_ref) + 1e-6
    assert bad == 0 and worst <= 5e-3
    assert float(G[:, N:].abs().sum()) == 0.0


def test_fused_cross_entropy_gives_the_same_clip_loss_as_the_materialised_path(dev):
    """NativeClipLoss with and without the fused cross-entropy on the same features (B = 1024, E = 512)"""
    import open_clip_amd.loss as L
    g = torch.Generator(device=dev).manual_seed(3)
    res = []
    for fused in (True, False):
        L.USE_FUSED_CE = fused
        img = torch.nn.functional.normalize(torch.randn(1024, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(1)), dim=-1).requires_grad_(True)
        txt = torch.nn.functional.normalize(torch.randn(1024, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(2)), dim=-1).requires_grad_(True)
        s = torch.tensor(14.2857, device=dev, requires_grad=True)
        loss = L.NativeClipLoss()(img, txt, s)
        loss.backward()
        res.append((float(loss.detach()), img.grad.clone(), txt.grad.clone(), float(s.grad)))
    L.USE_FUSED_CE = True
    assert abs(res[0][0] - res[1][0]) < 1e-5
    assert rel_l2(res[0][1], res[1][1]) < 2e-3 and rel_l2(res[0][2], res[1][2]) < 2e-3
    assert abs(res[0][3] - res[1][3]) <= 1e-3 * abs(res[1][3]) + 1e-8


def test_token_embed_backward_sorted_at_bench_size(dev):
    """the segment-reduce embedding backward at the bench's size and token statistics (B 4096 x 77 tokens of a 49408-entry table: SOT
    at position 0, one EOT, zero padding behind it -- runs of thousands of equal ids that cross many chunks) against index_add_"""
    from open_clip_amd import ops
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.synth import synthetic_batch
    cfg = get_model_config("ViT-B-32")
    text = synthetic_batch(cfg, 4096, seed=7)["text"].to(dev)
    B, L, C, V = 4096, 77, 512, 49408
    dx = torch.randn(B * L, C, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    dtable, dpos = torch.zeros(V, C, device=dev), torch.zeros(L, C, device=dev)
    ops.token_embed_bwd_sorted(text, dx, dtable, dpos)
    ref = torch.zeros(V, C, device=dev, dtype=torch.float64).index_add_(0, text.reshape(-1), dx.double())
    check("token_embed_bwd_sorted[4096x77x512] dtable", dtable, ref, rel=2e-6)
    check("token_embed_bwd_sorted[4096x77x512] dpos", dpos, dx.reshape(B, L, C).double().sum(0), rel=2e-6)


# ---- packed ("varlen") text batches: only the first eot+1 tokens of each sequence exist (ocn_seq_pack_plan & friends) ----------------
def _random_text(B, L, vocab, g, full_every=0):
    """SOT, random ids, EOT (= vocab-1, the maximum id) at a random position, zero padding behind it (tokenizer.py:209-226 layout)"""
    text = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        n = L - 2 if (full_every and b % full_every == 0) else int(torch.randint(0, L - 1, (1,), generator=g))
        text[b, 0] = vocab - 2
        text[b, 1:1 + n] = torch.randint(1, vocab - 2, (n,), generator=g)
        text[b, 1 + n] = vocab - 1
    return text


@pytest.mark.parametrize("B,L", [(7, 77), (4096, 77), (5, 16), (1500, 33)])
def test_seq_pack_plan_and_rows(dev, B, L):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(B + L)
    text = _random_text(B, L, 1000, g, full_every=5).to(dev)
    eot, plan, last_row, order = ops.seq_pack_plan(text, vocab=1000, buckets=True)
    seq_off = plan[:B + 1]
    ref_eot = text.argmax(dim=-1)
    ref_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), (ref_eot + 1).cumsum(0)])
    assert torch.equal(eot.long(), ref_eot) and torch.equal(seq_off.long(), ref_off) and torch.equal(last_row.long(), ref_off[1:] - 1)
    assert int(plan[B + 1]) == 0  # no id outside [0, vocab)
    # buckets: counts per ceil(len / 32), order = a permutation of the sequences grouped by that key
    nb = (ref_eot + 1 + 31) // 32
    counts = plan[B + 2:].long()
    assert counts.numel() == (L + 31) // 32 and torch.equal(counts, torch.bincount(nb - 1, minlength=counts.numel()))
    assert torch.equal(torch.sort(order.long()).values, torch.arange(B, device=dev)) and bool((nb[order.long()].diff() >= 0).all())
    M = int(seq_off[-1])
    tokens, posidx = ops.seq_pack_rows(text, seq_off, M)
    keep = torch.arange(L, device=dev)[None, :] <= ref_eot[:, None]
    assert torch.equal(tokens, text[keep]) and torch.equal(posidx.long(), torch.arange(L, device=dev).expand(B, L)[keep])
    _report(f"seq_pack_plan/rows[B{B} L{L}]: M={M} of {B * L} rows kept, bit-exact vs torch argmax/cumsum/masked_select")


@pytest.mark.parametrize("B,L,H,causal", [(6, 77, 3, True), (9, 50, 2, False), (4, 128, 1, True), (5, 20, 2, True), (3, 257, 2, False), (64, 77, 8, True)])
def test_attention_varlen(dev, B, L, H, causal):
    """packed batch (sequence b = rows seq_off[b]..seq_off[b+1]) against the dense fp32 reference run per sequence on its own
    rows; lengths cover 1, exact multiples of 32, L and everything between.  Same tolerances as the dense test."""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(B * L + H + 5)
    C = H * 64
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0], lens[-1] = 1, L
    if B > 2:
        lens[1] = min(L, 32)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
    M = int(off[-1])
    seq_off = off.to(torch.int32).to(dev)
    qkv = bf(torch.randn(M, 3 * C, generator=g) * 1.5).to(dev)
    dout = bf(torch.randn(M, C, generator=g)).to(dev)
    out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125, seq_off=seq_off)
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125, seq_off=seq_off)
    ref_o, ref_d, ref_lse, got_lse = [], [], [], []
    for b in range(B):
        r0, n = int(off[b]), int(lens[b])
        x = qkv[r0:r0 + n].float().requires_grad_(True)
        o, l = _attn_ref(x, 1, n, H, causal)
        o.backward(dout[r0:r0 + n].float())
        ref_o.append(o.detach()); ref_d.append(x.grad); ref_lse.append(l.detach().reshape(H, n))
        got_lse.append(lse.reshape(B, H, L)[b, :, :n])
    tag = f"attn_varlen[B{B} Lmax{L} H{H} c{int(causal)} rows {M}/{B * L}]"
    check(tag + " out", out, torch.cat(ref_o), rel=6e-3)
    check(tag + " lse", torch.cat([t.reshape(-1) for t in got_lse]), torch.cat([t.reshape(-1) for t in ref_lse]), rel=1e-5)
    check(tag + " dqkv", dqkv, torch.cat(ref_d), rel=1.5e-2)
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
    # the same batch launched in buckets of equal block count (workgroups sized for their own sequence): bit-identical results
    nb = (lens + 31) // 32
    order = torch.sort(nb, stable=True).indices.to(torch.int32).to(dev)
    lay = ops.SeqLayout(seq_off, order, torch.bincount(nb - 1, minlength=(L + 31) // 32).tolist())
    out_b, lse_b = ops.attn_fwd(qkv, B, L, H, causal, 0.125, seq_off=lay)
    dqkv_b = ops.attn_bwd(qkv, out_b, dout, lse_b, B, L, H, causal, 0.125, seq_off=lay)
    assert torch.equal(out_b, out) and torch.equal(dqkv_b, dqkv)
    assert all(torch.equal(lse_b.reshape(B, H, L)[b, :, :int(lens[b])], lse.reshape(B, H, L)[b, :, :int(lens[b])]) for b in range(B))


def test_attention_varlen_equals_dense_when_full(dev):
    """seq_off = multiples of L must give the dense kernels' results bit for bit (same code path, same order of operations)"""
    from open_clip_amd import ops
    B, L, H = 12, 77, 8
    g = torch.Generator().manual_seed(3)
    qkv = bf(torch.randn(B * L, 3 * H * 64, generator=g)).to(dev)
    dout = bf(torch.randn(B * L, H * 64, generator=g)).to(dev)
    seq_off = (torch.arange(B + 1, dtype=torch.int32) * L).to(dev)
    o1, l1 = ops.attn_fwd(qkv, B, L, H, True, 0.125)
    o2, l2 = ops.attn_fwd(qkv, B, L, H, True, 0.125, seq_off=seq_off)
    d1 = ops.attn_bwd(qkv, o1, dout, l1, B, L, H, True, 0.125)
    d2 = ops.attn_bwd(qkv, o2, dout, l2, B, L, H, True, 0.125, seq_off=seq_off)
    assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(d1, d2)


@pytest.mark.parametrize("B,dx_dtype", [(64, torch.float32), (4096, torch.float32), (4096, torch.bfloat16)])
def test_token_embed_packed(dev, B, dx_dtype):
    """packed embedding forward (bit-exact: table[token] + pos[l] in fp32) and its segment-reduce backward vs index_add_"""
    from open_clip_amd import ops
    L, C, V = 77, 512, 49408
    g = torch.Generator().manual_seed(B)
    text = _random_text(B, L, V, g).to(dev)
    table = torch.randn(V, C, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    pos = torch.randn(L, C, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    eot, plan, last_row, _ = ops.seq_pack_plan(text)
    seq_off = plan[:B + 1]
    M = int(seq_off[-1])
    tokens, posidx = ops.seq_pack_rows(text, seq_off, M)
    x = ops.token_embed_fwd_rows(tokens, posidx, table, pos)
    dense = ops.token_embed_fwd(text, table, pos).reshape(B, L, C)
    keep = torch.arange(L, device=dev)[None, :] <= eot[:, None]
    assert torch.equal(x, dense[keep])
    dx = torch.randn(M, C, device=dev, generator=torch.Generator(device=dev).manual_seed(4)).to(dx_dtype)
    dtable, dpos = torch.zeros(V, C, device=dev), torch.zeros(L, C, device=dev)
    ops.token_embed_bwd_sorted_varlen(tokens, seq_off, B, L, dx, dtable, dpos)
    ref_t = torch.zeros(V, C, device=dev, dtype=torch.float64).index_add_(0, tokens, dx.double())
    ref_p = torch.zeros(L, C, device=dev, dtype=torch.float64).index_add_(0, posidx.long(), dx.double())
    tag = f"token_embed packed[B{B} rows {M}/{B * L} {str(dx_dtype).split('.')[-1]}]"
    check(tag + " dtable", dtable, ref_t, rel=2e-6)
    check(tag + " dpos", dpos, ref_p, rel=2e-6)


@pytest.mark.parametrize("M", [177243, 1025])
def test_gemm_ragged_rows(dev, M):
    """the persistent GEMMs at a row count that is no multiple of the 256-row tile (what a packed text batch produces), every epilogue
    of the block, output pre-filled with NaN; and the wgrad over the same ragged M"""
    from open_clip_amd import ops
    C = 512
    g = torch.Generator(device=dev).manual_seed(M)
    a = bf(torch.randn(M, C, device=dev, generator=g))
    w = bf(torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5)
    bias = torch.randn(4 * C, device=dev, generator=g)
    out = torch.full((M, 4 * C), float("nan"), device=dev, dtype=torch.bfloat16)
    aux = torch.full((M, 4 * C), 255, device=dev, dtype=torch.uint8)
    ops.gemm_nt(ops.EPI_BIAS_GELU, a, w, out, bias=bias, aux=aux)
    tail = slice(max(0, M - 600), M)
    ref = torch.nn.functional.gelu(a[tail].float() @ w.float().t() + bias)
    assert torch.isfinite(out.float()).all() and int(aux.max()) <= 252
    check(f"gemm_nt ragged M={M} gelu tail rows", out[tail], ref, rel=2.5e-3)
    resid = torch.randn(M, C, device=dev, generator=g)
    out2 = torch.full((M, C), float("nan"), device=dev)
    ops.gemm_nt(ops.EPI_BIAS_RESID_F32, out, bf(w.float().t().contiguous()), out2, bias=bias[:C].contiguous(), resid=resid)
    ref2 = out[tail].float() @ w.float() + bias[:C] + resid[tail]
    assert torch.isfinite(out2).all()
    check(f"gemm_nt ragged M={M} resid tail rows", out2[tail], ref2, rel=2e-5)
    dw, db = torch.zeros(4 * C, C, device=dev), torch.zeros(4 * C, device=dev)
    ops.gemm_tn_accum(out, a, dw, db)
    refw = torch.zeros(4 * C, C, device=dev, dtype=torch.float64)
    for r0 in range(0, M, 16384):
        refw += (out[r0:r0 + 16384].float().t() @ a[r0:r0 + 16384].float()).double()
    check(f"gemm_tn ragged M={M} dW", dw, refw, rel=2e-4)
    check(f"gemm_tn ragged M={M} dbias", db, out.float().sum(0).double(), rel=2e-4)


@pytest.mark.parametrize("M,N,K", [(4096, 512, 32768), (4096, 768, 16384), (1024, 1024, 8192), (2500, 264, 4096)])
def test_gemm_nt_split_k(dev, M, N, K):
    """ocn_gemm_nt_splitk (round 6: the loss's G @ Y at one rank's share of the 8-GPU loss has 32 output tiles and K = 32768) against fp32 torch: the plain
    product, and with the reduction's riders -- per-row scale, bf16 rows subtracted with a factor, a device-resident scale"""
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=g)).to(dev)
    b = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    ks = ops.gemm_nt_splitk_plan(M, N, K)
    assert ks >= 2 and K % (128 * ks) == 0, ks
    ref = a.float() @ b.float().t()
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_nt_splitk(a, b, out, ks)
    check(f"gemm_nt_splitk[{M}x{N}x{K}, {ks} slices]", out, ref, rel=2e-5)
    rs = (torch.rand(M, generator=g) + 0.5).to(dev)
    sub = bf(torch.randn(M, N, generator=g)).to(dev)
    sc = torch.tensor([3.25], device=dev)
    out2 = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_nt_splitk(a, b, out2, ks, rowscale=rs, sub_rows=sub, sub_alpha=0.125, scale=sc)
    check(f"gemm_nt_splitk[{M}x{N}x{K}] with row scale, subtracted rows, device scale", out2, 3.25 * (rs[:, None] * ref - 0.125 * sub.float()), rel=2e-5)
    # a shape that fills the chip is not split
    assert ops.gemm_nt_splitk_plan(32768, 512, 32768) == 1 and ops.gemm_nt_splitk_plan(4096, 512, 1024) == 1


def test_row_scale_helpers(dev):
    from open_clip_amd import ops
    g = torch.Generator().manual_seed(5)
    R, E = 1000, 512
    x16 = bf(torch.randn(R, E, generator=g)).to(dev)
    sc = (torch.rand(R, generator=g) * 1e-3 + 1e-6).to(dev)
    got = ops.scale_rows_bf16(x16, sc)
    assert torch.equal(got, (x16.float() * sc[:, None]).to(torch.bfloat16))
    big = torch.randn(R + 7, E, generator=g).to(dev)
    want = big.clone()
    want[3:3 + R] -= (0.5 / sc)[:, None] * x16.float()
    ops.sub_scaled_rows(big[3:3 + R], x16, sc, 0.5)
    check("sub_scaled_rows", big, want, rel=1e-6)
