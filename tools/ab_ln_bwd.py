"""A/B of the software-pipelined LayerNorm backward (layernorm.hip::ln_bwd_pf_kernel, two rows per wave in flight) against the one-row kernel
(developer knob 8 = 2) at the bench's shapes; dx compared between the two.  Run through gpurun: python tools/ab_ln_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print(f"{'shape':28s} {'pipelined ms':>13s} {'one row ms':>11s} {'TB/s':>6s} {'TB/s':>6s}  max |dx diff|")
for name, M, C in (("image  [204800 x 768]", 204800, 768), ("text packed [177803 x 512]", 177803, 512), ("text dense [315392 x 512]", 315392, 512)):
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(M, C, device=dev, generator=g) * 2 + 0.3
    dy = torch.randn(M, C, device=dev, generator=g).bfloat16()
    dres = torch.randn(M, C, device=dev, generator=g)
    w = torch.randn(C, device=dev, generator=g)
    _, _, mean, rstd = ops.layernorm_fwd(x, w, torch.zeros_like(w))
    outs, best = [], [1e9, 1e9]
    for knob in (0, 2):
        _lib.call("ocn_set_tuning", 8, knob)
        dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx, dx16 = ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres, want_f32=True, want_bf16=True)
        outs.append((dx.clone(), dx16.clone(), dw, db))
    for rnd in range(4):
        for i, knob in enumerate((0, 2)):
            _lib.call("ocn_set_tuning", 8, knob)
            dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            best[i] = min(best[i], timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres, want_f32=True, want_bf16=True)))
    _lib.call("ocn_set_tuning", 8, 0)
    gb = M * C * 16 / 1e9
    diff = float((outs[0][0] - outs[1][0]).abs().max())
    dwrel = float((outs[0][2] - outs[1][2]).norm() / outs[1][2].norm())
    print(f"{name:28s} {best[0]:13.4f} {best[1]:11.4f} {gb / best[0]:6.2f} {gb / best[1]:6.2f}  {diff:.2e} (dgamma rel {dwrel:.1e}, dx16 equal: {torch.equal(outs[0][1], outs[1][1])})", flush=True)
