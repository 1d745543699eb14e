"""Attention backward at the bench's image / packed-text shapes with parts switched off (developer knob 1: 1 = no input staging, 2 = no
arithmetic, 4 = no stores; results are wrong then): how much of a launch is memory, how much arithmetic, how well the two overlap.
usage: python tools/attn_bwd_ablation.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def shape(which):
    g = torch.Generator(device=dev).manual_seed(1)
    if which == "image":
        B, L, H, causal, lay = 4096, 50, 12, False, None
        M = B * L
    else:
        B, L, H, causal = 4096, 77, 8, True
        lens = torch.randint(8, 77, (B,), generator=torch.Generator().manual_seed(1234)) + 1
        off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
        M = int(off[-1])
        nb = (lens + 31) // 32
        lay = ops.SeqLayout(off.to(torch.int32).to(dev), torch.sort(nb, stable=True).indices.to(torch.int32).to(dev),
                            torch.bincount(nb - 1, minlength=(L + 31) // 32).tolist())
    C = H * 64
    qkv = (torch.randn(M, 3 * C, device=dev, generator=g) * 1.5).bfloat16()
    dout = torch.randn(M, C, device=dev, generator=g).bfloat16()
    out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125, seq_off=lay)
    return lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125, seq_off=lay), M * C * 2 * 8


NAMES = {0: "full", 2: "no arithmetic (loads + stores)", 5: "no loads, no stores (arithmetic)", 1: "no loads", 4: "no stores", 3: "stores only", 6: "loads only", 7: "nothing"}
for which in ("image", "text packed"):
    fn, nbytes = shape("image" if which == "image" else "text")
    line = []
    for mask in (0, 2, 5, 1, 4, 6, 3, 7):
        _lib.call("ocn_set_tuning", 1, mask)
        t = timeit(fn)
        line.append(f"{NAMES[mask]} {t:.3f} ms")
    _lib.call("ocn_set_tuning", 1, 0)
    print(f"{which}: " + " | ".join(line) + f"   (algorithmic {nbytes / 1e9:.2f} GB)", flush=True)
