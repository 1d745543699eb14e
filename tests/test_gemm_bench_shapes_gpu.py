"""The GEMM kernels at the EXACT shapes of the bench step (ViT-B-32, local batch 4096: M = 4096*50 = 204800 image tokens,
M = 4096*77 = 315392 text tokens), every epilogue the step uses at that shape, against fp32 torch evaluated on the GPU in row
chunks.  These are the launches whose persistent tile walk (37-77 tiles per workgroup, banded order, start stagger) the small
parity shapes never reach.

Tolerances: fp32 outputs rel-L2 <= 2e-5 (accumulation order); bf16 outputs rel-L2 <= 2.5e-3 and every element within one bf16
rounding of the fp32 value (|err| <= 2^-7 |ref| + abs_tol); weight gradients (fp32 atomics over up to 77 M-splits) rel-L2 <= 2e-4.
Outputs are pre-filled with NaN so that a tile the walk never visits cannot pass."""
import pytest
import torch

from tests.test_kernels_gpu import _report, bf, rel_l2

pytestmark = pytest.mark.gpu

MI, MT = 4096 * 50, 4096 * 77
CHUNK = 16384
EPI_NAMES = {0: "bf16", 1: "bias+gelu", 2: "bias+resid_f32", 3: "dgelu", 4: "f32"}

# (name, M, N, K, epilogue, bias?) -- forward and dgrad GEMMs of one residual block of each tower (model.py::_block_forward /
# _BlockFn.backward; reference transformer.py:169,246,295-299,328-329)
NT_CASES = []
for tag, M, C in (("img", MI, 768), ("txt", MT, 512)):
    NT_CASES += [
        (f"{tag} qkv", M, 3 * C, C, 0, True), (f"{tag} out_proj", M, C, C, 2, True), (f"{tag} c_fc", M, 4 * C, C, 1, True),
        (f"{tag} c_proj", M, C, 4 * C, 2, True), (f"{tag} dgelu", M, 4 * C, C, 3, False), (f"{tag} dh2", M, C, 4 * C, 0, False),
        (f"{tag} da", M, C, C, 0, False), (f"{tag} dh1", M, C, 3 * C, 0, False),
    ]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from open_clip_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _gelu_and_derivative(x):
    return torch.nn.functional.gelu(x), 0.5 * (1 + torch.erf(x * 2 ** -0.5)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


@pytest.mark.parametrize("name,M,N,K,epi,with_bias", NT_CASES, ids=[c[0].replace(" ", "_") for c in NT_CASES])
def test_gemm_nt_at_bench_shape(dev, name, M, N, K, epi, with_bias):
    from open_clip_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + 31 * N + K + epi)
    a = bf(torch.randn(M, K, device=dev, generator=g))
    b = bf(torch.randn(N, K, device=dev, generator=g) * K ** -0.5)
    bias = torch.randn(N, device=dev, generator=g) if with_bias else None
    f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    resid = torch.randn(M, N, device=dev, generator=g) if epi == ops.EPI_BIAS_RESID_F32 else None
    aux = None
    if epi == ops.EPI_BIAS_GELU:
        aux = torch.full((M, N), 255, device=dev, dtype=torch.uint8)  # 255 is outside the code range [0, 252]: a row never written shows
    elif epi == ops.EPI_DGELU:
        aux = torch.randint(0, 253, (M, N), device=dev, generator=g, dtype=torch.uint8)  # the 8-bit codes of gelu' (ops.dgelu_decode)
    ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux)
    torch.cuda.synchronize()
    worst, worst_aux, bad = 0.0, 0.0, 0
    for r0 in range(0, M, CHUNK):
        sl = slice(r0, min(M, r0 + CHUNK))
        acc = a[sl].float() @ b.float().t()
        if bias is not None:
            acc += bias
        ref2 = None
        if epi == ops.EPI_BIAS_GELU:
            ref, ref2 = _gelu_and_derivative(acc)
        elif epi == ops.EPI_BIAS_RESID_F32:
            ref = acc + resid[sl]
        elif epi == ops.EPI_DGELU:
            ref = acc * ops.dgelu_decode(aux[sl])
        else:
            ref = acc
        got = out[sl].float()
        assert torch.isfinite(got).all(), f"{name}: rows {r0}.. hold non-finite values (tile never written?)"
        worst = max(worst, rel_l2(got, ref))
        if not f32out:
            bad += int(((got - ref).abs() > ref.abs() * 2.0 ** -7 + 2e-3).sum())
        if ref2 is not None:
            assert int(aux[sl].max()) <= 252, f"{name}: saved gelu' rows {r0}.. hold a byte outside the code range (tile never written?)"
            got2 = ops.dgelu_decode(aux[sl])
            worst_aux = max(worst_aux, rel_l2(got2, ref2))
            bad += int(((got2 - ref2).abs() > 2.6e-3).sum())  # half a step of the 8-bit fixed point (1/400) + the erf approximation
    _report(f"bench-shape gemm_nt {name:12s} [{M}x{N}x{K}] {EPI_NAMES[epi]:15s} rel_l2={worst:.3e}" + (f" aux rel_l2={worst_aux:.3e}" if epi == 1 else ""))
    assert worst <= (2e-5 if f32out else 2.5e-3), (name, worst)
    assert worst_aux <= 4e-3 and bad == 0, (name, worst_aux, bad)  # 8-bit gelu': rms error 0.0014 on values of rms ~0.6


TN_CASES = []
for tag, M, C in (("img", MI, 768), ("txt", MT, 512)):
    TN_CASES += [(f"{tag} wgrad qkv", M, 3 * C, C), (f"{tag} wgrad out_proj", M, C, C), (f"{tag} wgrad c_fc", M, 4 * C, C), (f"{tag} wgrad c_proj", M, C, 4 * C)]


@pytest.mark.parametrize("name,M,N,K", TN_CASES, ids=[c[0].replace(" ", "_") for c in TN_CASES])
def test_gemm_tn_at_bench_shape(dev, name, M, N, K):
    """dW[N,K] += A[M,N]^T B[M,K], dbias[N] += colsum(A): the weight / bias gradients of the block's four Linears"""
    from open_clip_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + 17 * N + K)
    a = bf(torch.randn(M, N, device=dev, generator=g))
    b = bf(torch.randn(M, K, device=dev, generator=g))
    pre = torch.randn(N, K, device=dev, generator=g)  # the kernel accumulates INTO dW: start from a known non-zero value
    dw, db = pre.clone(), torch.zeros(N, device=dev)
    ops.gemm_tn_accum(a, b, dw, db)
    torch.cuda.synchronize()
    ref = torch.zeros(N, K, device=dev, dtype=torch.float64)
    refb = torch.zeros(N, device=dev, dtype=torch.float64)
    for r0 in range(0, M, CHUNK):
        sl = slice(r0, min(M, r0 + CHUNK))
        ref += (a[sl].float().t() @ b[sl].float()).double()
        refb += a[sl].float().sum(0).double()
    e1, e2 = rel_l2(dw.double() - pre.double(), ref), rel_l2(db, refb)
    _report(f"bench-shape gemm_tn {name:18s} dW[{N}x{K}] over M={M}: rel_l2={e1:.3e} dbias rel_l2={e2:.3e}")
    assert torch.isfinite(dw).all() and e1 <= 2e-4 and e2 <= 2e-4, (name, e1, e2)


@pytest.mark.parametrize("tag,M,C,bias", [("img", MI, 768, True), ("txt", MT, 512, True), ("txt packed", 177803, 512, True), ("img no bias", MI, 768, False),
                                          ("small (two launches)", 3000, 128, True)])
def test_gemm_tn_pair_at_bench_shape(dev, tag, M, C, bias):
    """ocn_gemm_tn_accum2: the out-proj (dW[C,C]) and QKV (dW[3C,C]) weight gradients of a block in ONE launch, against fp32 torch in row
    chunks and against the two single launches; both accumulate INTO pre-filled dW"""
    from open_clip_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + C)
    a1, b1 = bf(torch.randn(M, C, device=dev, generator=g)), bf(torch.randn(M, C, device=dev, generator=g))
    a2, b2 = bf(torch.randn(M, 3 * C, device=dev, generator=g)), bf(torch.randn(M, C, device=dev, generator=g))
    pre1, pre2 = torch.randn(C, C, device=dev, generator=g), torch.randn(3 * C, C, device=dev, generator=g)
    dw1, dw2 = pre1.clone(), pre2.clone()
    db1, db2 = (torch.zeros(C, device=dev), torch.zeros(3 * C, device=dev)) if bias else (None, None)
    ops.gemm_tn_accum2(a1, b1, dw1, db1, a2, b2, dw2, db2)
    s1, s2 = pre1.clone(), pre2.clone()
    sb1, sb2 = (torch.zeros(C, device=dev), torch.zeros(3 * C, device=dev)) if bias else (None, None)
    ops.gemm_tn_accum(a1, b1, s1, sb1)
    ops.gemm_tn_accum(a2, b2, s2, sb2)
    torch.cuda.synchronize()
    worst = 0.0
    for a, b, dw, pre, db, single, sb in ((a1, b1, dw1, pre1, db1, s1, sb1), (a2, b2, dw2, pre2, db2, s2, sb2)):
        ref = torch.zeros(dw.shape, device=dev, dtype=torch.float64)
        refb = torch.zeros(dw.shape[0], device=dev, dtype=torch.float64)
        for r0 in range(0, M, CHUNK):
            sl = slice(r0, min(M, r0 + CHUNK))
            ref += (a[sl].float().t() @ b[sl].float()).double()
            refb += a[sl].float().sum(0).double()
        e = [rel_l2(dw.double() - pre.double(), ref), rel_l2(dw.double() - pre.double(), single.double() - pre.double())]
        if bias:
            e += [rel_l2(db, refb), rel_l2(db, sb)]
        assert torch.isfinite(dw).all() and max(e) <= 2e-4, (tag, dw.shape, e)
        worst = max(worst, max(e))
    _report(f"bench-shape paired wgrad {tag:12s} dW[{C}x{C}] + dW[{3 * C}x{C}] over M={M}: worst rel_l2 (vs fp32 torch, vs two single launches, dW and dbias) = {worst:.3e}")
