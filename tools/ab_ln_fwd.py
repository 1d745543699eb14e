"""LayerNorm forward at the bench's shapes: non-temporal (default) against default cache policy for x (developer knob 12).  Run through gpurun: python tools/ab_ln_fwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=12):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


SHAPES = (("image  [204800 x 768]", 204800, 768), ("text packed [177803 x 512]", 177803, 512), ("text dense [315392 x 512]", 315392, 512),
          ("ViT-L-14 [526336 x 1024]", 526336, 1024), ("ViT-H-14 [263168 x 1280]", 263168, 1280))
NAMES = {1: "default policy for x", 0: "non-temporal x (shipped)"}
ref = {}
for rnd in range(2):
    for mode in (1, 0):
        _lib.call("ocn_set_tuning", 12, mode)
        line = [f"round {rnd} {NAMES[mode]:36s}"]
        for name, M, C in SHAPES:
            g = torch.Generator(device=dev).manual_seed(3)
            x = torch.randn(M, C, device=dev, generator=g) * 2 + 0.3
            w = torch.randn(C, device=dev, generator=g)
            b = torch.randn(C, device=dev, generator=g)
            y, _, mean, rstd = ops.layernorm_fwd(x, w, b)
            if name not in ref:
                ref[name] = (y.clone(), mean.clone(), rstd.clone())
            same = torch.equal(y, ref[name][0]) and torch.equal(mean, ref[name][1]) and torch.equal(rstd, ref[name][2])
            t = min(timeit(lambda: ops.layernorm_fwd(x, w, b)) for _ in range(4))
            line.append(f"{name.split('[')[0].strip()} {t * 1e3:.1f} us {M * C * 6 / 1e9 / t:.2f} TB/s{'' if same else ' DIFFERS'}")
            del x, y
        print(" | ".join(line), flush=True)
_lib.call("ocn_set_tuning", 12, 0)
