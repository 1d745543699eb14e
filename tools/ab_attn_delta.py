"""A/B of the attention backward's row term (round 6): delta from P and dP (shipped for sequences of up to two key blocks: O is not read) against
delta = sum_d dO * O (developer knob 2 = 6), at the bench's image shape and the packed text tower's short buckets.  usage: python tools/ab_attn_delta.py"""
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


for name, B, L, H, causal in (("image tower [4096 x 50 x 12 heads]", 4096, 50, 12, False), ("[4096 x 30 x 8 heads, causal]", 4096, 30, 8, True),
                              ("[4096 x 64 x 8 heads]", 4096, 64, 8, False)):
    C = H * 64
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = (torch.randn(B * L, 3 * C, device=dev, generator=g) * 1.5).bfloat16()
    dout = torch.randn(B * L, C, device=dev, generator=g).bfloat16()
    out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125)
    res = {}
    for tag, k in (("delta from O (round 5)", 6), ("delta from P, dP (shipped)", 0)):
        _lib.call("ocn_set_tuning", 2, k)
        got = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125)
        ms = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125))
        nbytes = B * L * C * 2 * (3 + (2 if k == 6 else 1) + 3)
        res[tag] = got
        print(f"{name}: {tag:28s} {ms * 1e3:7.1f} us  {nbytes / ms / 1e9:6.2f} TB/s of its own algorithmic bytes ({nbytes / 1e9:.2f} GB)", flush=True)
    a, b = res["delta from O (round 5)"].float(), res["delta from P, dP (shipped)"].float()
    print(f"{name}: rel_l2 between the two forms {float((a - b).norm() / a.norm()):.3e}", flush=True)
_lib.call("ocn_set_tuning", 2, 0)
