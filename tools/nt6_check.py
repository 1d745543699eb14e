"""Developer check of the trickled-epilogue NT GEMM (gemm_nt6.hip) on the MI355X (run through gpurun): every epilogue against fp32
torch on a ragged-M shape and on one bench shape (row-chunked), then timing against the 256x256 kernel (variant 5), each with and
without the epilogue's memory traffic / VALU work (developer knobs 32 / 128 / 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
Mi, Mt = 4096 * 50, 4096 * 77


def gelu_ref(x):
    return torch.nn.functional.gelu(x), 0.5 * (1 + torch.erf(x * 2 ** -0.5)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


def reference(epi, a, b, bias, resid, aux):
    acc = a.float() @ b.float().t() + (bias if bias is not None else 0.0)
    if epi == ops.EPI_BF16:
        return acc, None
    if epi == ops.EPI_BIAS_GELU:
        return gelu_ref(acc)
    if epi == ops.EPI_BIAS_RESID_F32:
        return acc + resid, None
    if epi == ops.EPI_DGELU:
        return acc * aux.float(), None
    return acc, None


def run(epi, a, b, bias, resid, aux_in):
    M, N = a.shape[0], b.shape[0]
    f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    aux = None
    if epi == ops.EPI_BIAS_GELU:
        aux = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    elif epi == ops.EPI_DGELU:
        aux = aux_in
    ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux)
    return out, aux if epi == ops.EPI_BIAS_GELU else None


def check(M, N, K, chunk=16384):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16().to(dev)
    b = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    auxin = torch.randn(M, N, generator=g).bfloat16().to(dev)
    for epi in (0, 1, 2, 3, 4):
        res = {}
        for variant in (6, 5):
            _lib.call("ocn_set_gemm_variant", variant)
            use_bias = None if epi == ops.EPI_DGELU else bias
            out, aux = run(epi, a, b, use_bias, resid if epi == 2 else None, auxin)
            torch.cuda.synchronize()
            worst = 0.0
            worst_aux = 0.0
            for r0 in range(0, M, chunk):
                sl = slice(r0, min(M, r0 + chunk))
                ref, ref2 = reference(epi, a[sl], b, use_bias, resid[sl], auxin[sl])
                err = (out[sl].float() - ref).norm() / ref.norm()
                worst = max(worst, float(err))
                if ref2 is not None:
                    worst_aux = max(worst_aux, float((aux[sl].float() - ref2).norm() / ref2.norm()))
            res[variant] = (worst, worst_aux, bool(torch.isnan(out.float()).any()))
        print(f"check [{M}x{N}x{K}] epi {epi}: " + " | ".join(f"v{v}: rel_l2 {r[0]:.2e} aux {r[1]:.2e} nan={r[2]}" for v, r in res.items()), flush=True)
    _lib.call("ocn_set_gemm_variant", 0)


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def bench():
    cases = [("img qkv", Mi, 2304, 768, [0]), ("img da", Mi, 768, 768, [0]), ("img dh2", Mi, 768, 3072, [0]), ("img fc", Mi, 3072, 768, [0, 1, 3]),
             ("img out", Mi, 768, 768, [2]), ("img proj", Mi, 768, 3072, [2]), ("txt qkv", Mt, 1536, 512, [0]), ("txt fc", Mt, 2048, 512, [0, 1, 3]),
             ("txt out", Mt, 512, 512, [2]), ("txt proj", Mt, 512, 2048, [2])]
    tot = {}
    for name, M, N, K, epis in cases:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        for epi in epis:
            f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
            resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
            aux = torch.randn(M, N, device=dev).bfloat16() if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
            row = []
            for variant, mask, tag in ((5, 0, "nt5"), (6, 0, "nt6"), (5, 32 | 128 | 1, "nt5 bare"), (6, 32 | 128 | 1, "nt6 bare"), (16, 0, "nt6sync"), (5, 0, "nt5"), (6, 0, "nt6")):
                _lib.call("ocn_set_tuning", 13, 1 if variant == 16 else 0)
                variant = 6 if variant == 16 else variant
                _lib.call("ocn_set_gemm_variant", variant | (mask << 8))
                ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
                row.append(f"{tag} {ms:.3f} ({2.0 * M * N * K / ms / 1e9:5.0f})")
                if mask == 0 and tag != "nt6sync":
                    tot[variant] = tot.get(variant, 0.0) + ms / 2
            print(f"{name:9s} epi {epi}: " + " | ".join(row), flush=True)
            del out, resid, aux
        del a, b
    print("sum ms: " + ", ".join(f"v{v} {t:.3f}" for v, t in tot.items()))
    _lib.call("ocn_set_gemm_variant", 0)


def one(variant, mask, sync, epi, M, N, K, flavour=0):
    _lib.call("ocn_set_tuning", 13, sync)
    _lib.call("ocn_set_tuning", 14, flavour)
    _lib.call("ocn_set_gemm_variant", variant | (mask << 8))
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
    aux = torch.randn(M, N, device=dev).bfloat16() if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
    ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
    print(f"one v{variant} mask {mask} sync {sync} epi {epi} [{M}x{N}x{K}] store-flavour {flavour}: {ms:.3f} ms", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "one":
        one(*[int(v) for v in sys.argv[2:10]])
        sys.exit(0)
    if what in ("all", "check"):
        check(1000, 256, 512)
        check(3000, 384, 576 + 64)
        check(Mi // 8, 768, 768)
        check(Mi // 4, 3072, 768, chunk=8192)
    if what in ("all", "bench"):
        bench()
