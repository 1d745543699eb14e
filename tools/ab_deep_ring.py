"""A/B of the deep epilogue-operand ring of the persistent NT GEMM (gemm_nt5.hip::epilogue5, AUX bit 5) on the MI355X: the dGELU and
fp32-residual epilogues at the bench's shapes with the ring deep (shipped) and shallow (developer knob 0x100000), outputs compared
BITWISE (same arithmetic, only the order of the epilogue's loads and stores differs) and timed in interleaved rounds.
Run through gpurun:  python tools/ab_deep_ring.py > gpurun_out/deep_ring.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
Mi, Mt = 4096 * 50, 177803  # image rows; packed text rows of the bench's synthetic captions
FLIP = 256 * 0x100000
SHAPES = [("img proj+resid", ops.EPI_BIAS_RESID_F32, Mi, 768, 3072), ("img out+resid", ops.EPI_BIAS_RESID_F32, Mi, 768, 768),
          ("img dgelu", ops.EPI_DGELU, Mi, 3072, 768), ("txt proj+resid", ops.EPI_BIAS_RESID_F32, Mt, 512, 2048),
          ("txt out+resid", ops.EPI_BIAS_RESID_F32, Mt, 512, 512), ("txt dgelu", ops.EPI_DGELU, Mt, 2048, 512)]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print(f"{'shape':16s} {'deep ms':>9s} {'shallow ms':>11s} {'deep TF/s':>10s} {'shallow TF/s':>13s}  bitwise-equal")
tot = [0.0, 0.0]
for name, epi, M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    b = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    f32out = epi == ops.EPI_BIAS_RESID_F32
    resid = torch.randn(M, N, device=dev, generator=g) if f32out else None
    aux = torch.randint(0, 253, (M, N), device=dev, generator=g, dtype=torch.uint8) if epi == ops.EPI_DGELU else None
    outs = []
    for v in (0, FLIP):
        _lib.call("ocn_set_gemm_variant", v)
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
        ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux)
        outs.append(out)
    same = torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0].float()).all())
    best = [1e9, 1e9]
    for rnd in range(4):
        for i, v in enumerate((0, FLIP)):
            _lib.call("ocn_set_gemm_variant", v)
            best[i] = min(best[i], timeit(lambda: ops.gemm_nt(epi, a, b, outs[i], bias=bias, resid=resid, aux=aux)))
    fl = 2.0 * M * N * K / 1e9
    tot[0] += best[0]
    tot[1] += best[1]
    print(f"{name:16s} {best[0]:9.4f} {best[1]:11.4f} {fl / best[0]:10.0f} {fl / best[1]:13.0f}  {same}", flush=True)
    del a, b, resid, aux, outs
_lib.call("ocn_set_gemm_variant", 0)
print(f"sum: deep {tot[0]:.3f} ms, shallow {tot[1]:.3f} ms  (per step: 12 blocks x these six launches)")
