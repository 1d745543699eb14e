"""GEMM micro-benchmark on the MI355X (developer tool; run through gpurun).  For the ViT-B-32 local-batch-4096
shapes it times every NT tile geometry x epilogue and the TN kernel with HIP events and prints TFLOP/s, after
checking each geometry against torch on a ragged shape."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
Mi, Mt = 4096 * 50, 4096 * 77
SHAPES_NT = [  # (name, M, N, K)
    ("img qkv", Mi, 2304, 768), ("img out", Mi, 768, 768), ("img fc", Mi, 3072, 768), ("img proj", Mi, 768, 3072),
    ("txt qkv", Mt, 1536, 512), ("txt out", Mt, 512, 512), ("txt fc", Mt, 2048, 512), ("txt proj", Mt, 512, 2048),
]
# variant ids may carry a developer ablation mask in bits 8+: 4 + 256*mask
variants = [] if (len(sys.argv) > 1 and sys.argv[1] == "-") else [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4,5".split(","))]  # "-" = skip the NT sweep
epis = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,3".split(","))]


def timeit(fn, iters=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def check(variant):
    _lib.call("ocn_set_gemm_variant", variant)
    g = torch.Generator().manual_seed(0)
    if variant >> 8:
        return float("nan")
    M, N, K = 1000, 640, 320
    a = torch.randn(M, K, generator=g).bfloat16().to(dev)
    b = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(dev)
    out = ops.gemm_nt(ops.EPI_F32, a, b, torch.empty(M, N, device=dev))
    ref = a.float() @ b.float().t()
    return float((out - ref).norm() / ref.norm())


for v in variants:
    print(f"variant {v}: rel_l2 vs torch = {check(v):.2e}")

print(f"{'shape':10s} {'epi':>3s} " + " ".join(f"{'v' + str(v):>9s}" for v in variants) + "   (TFLOP/s)")
tot = {v: 0.0 for v in variants}
for name, M, N, K in (SHAPES_NT if variants else []):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    for epi in epis:
        f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
        resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
        aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
        best = {v: 1e9 for v in variants}
        for rnd in range(3):  # interleaved rounds, best-of (DVFS / cache state drifts between back-to-back variants)
            for v in variants:
                _lib.call("ocn_set_gemm_variant", v)
                best[v] = min(best[v], timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux), iters=4))
        row = []
        for v in variants:
            row.append(2.0 * M * N * K / best[v] / 1e9)
            tot[v] += best[v]
        print(f"{name:10s} {epi:3d} " + " ".join(f"{t:9.0f}" for t in row))
        del out, resid, aux
    del a, b
print("sum ms per variant:", {v: round(t, 2) for v, t in tot.items()})
_lib.call("ocn_set_gemm_variant", 0)

print("TN (wgrad): variant 1 = general 128x128, 3 = 256x256 hand-scheduled (tn5)")
for tv in (1, 3):
    _lib.call("ocn_set_gemm_variant", tv << 4)
    g = torch.Generator().manual_seed(1)
    M, N, K = 3000, 640, 328
    a = torch.randn(M, N, generator=g).bfloat16().to(dev)
    b = torch.randn(M, K, generator=g).bfloat16().to(dev)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.gemm_tn_accum(a, b, dw, db)
    ref = a.float().t() @ b.float()
    print(f"  TN variant {tv}: dW rel_l2 = {float((dw - ref).norm() / ref.norm()):.2e}  dbias rel_l2 = {float((db - a.float().sum(0)).norm() / a.float().sum(0).norm()):.2e}")
for name, M, N, K in SHAPES_NT:
    a = torch.randn(M, N, device=dev).bfloat16()
    b = torch.randn(M, K, device=dev).bfloat16()
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    row = []
    for tv in (1, 3):
        _lib.call("ocn_set_gemm_variant", tv << 4)
        ms = timeit(lambda: ops.gemm_tn_accum(a, b, dw, db))
        ms0 = timeit(lambda: ops.gemm_tn_accum(a, b, dw, None))
        row.append(f"v{tv}: {2.0 * M * N * K / ms / 1e9:6.0f} TF/s {ms:.3f} ms (no dbias {2.0 * M * N * K / ms0 / 1e9:6.0f})")
    print(f"{name:10s} dW[{N},{K}] over M={M}:  " + "   ".join(row))
    del a, b
_lib.call("ocn_set_gemm_variant", 0)
