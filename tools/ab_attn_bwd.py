"""A/B of the attention backward builds at the bench's shapes (developer knobs of include/openclip_hip_debug.h):
  one-pass (knob 2 = 5) | two-pass dK / dV, 168 registers (knob 2 = 4) | the same + the generic kernel for the causal buckets too (knob 6 = 1)
Results of every variant are compared with the one-pass build's.  usage: python tools/ab_attn_bwd.py"""
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


def knobs(k2, k6):
    _lib.call("ocn_set_tuning", 2, k2)
    _lib.call("ocn_set_tuning", 6, k6)


VARIANTS = [("one-pass", 5, 0), ("two-pass", 4, 0), ("two-pass, generic kernel for causal buckets", 4, 1), ("one-pass, generic kernel for causal buckets", 5, 1)]


def case(name, B, L, H, causal, lens=None):
    C = H * 64
    g = torch.Generator(device=dev).manual_seed(1)
    if lens is None:
        M, lay = B * L, None
    else:
        off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
        M = int(off[-1])
        nb = (lens + 31) // 32
        order = torch.sort(nb, stable=True).indices.to(torch.int32).to(dev)
        lay = ops.SeqLayout(off.to(torch.int32).to(dev), order, torch.bincount(nb - 1, minlength=(L + 31) // 32).tolist())
    qkv = (torch.randn(M, 3 * C, device=dev, generator=g) * 1.5).bfloat16()
    dout = torch.randn(M, C, device=dev, generator=g).bfloat16()
    knobs(5, 0)
    out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125, seq_off=lay)
    ref = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125, seq_off=lay)
    nbytes = M * C * 2 * (3 + 2 + 3)  # qkv, out + dout read, dqkv written
    for vname, k2, k6 in VARIANTS:
        if not causal and k6:
            continue
        knobs(k2, k6)
        got = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125, seq_off=lay)
        err = float((got.float() - ref.float()).norm() / ref.float().norm())
        ms = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125, seq_off=lay))
        print(f"{name:34s} {vname:46s} {ms:.4f} ms  {nbytes / ms / 1e9:5.2f} TB/s  rel diff vs one-pass {err:.1e}", flush=True)
    knobs(0, 0)


g = torch.Generator().manual_seed(1234)
case("image B4096 L50 H12", 4096, 50, 12, False)
case("text packed B4096 L77 H8 (3 buckets)", 4096, 77, 8, True, torch.randint(8, 77, (4096,), generator=g) + 1)
case("text dense B4096 L77 H8", 4096, 77, 8, True)
case("ViT-L-14 image B1024 L257 H16", 1024, 257, 16, False)
