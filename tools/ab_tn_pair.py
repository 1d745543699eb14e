"""A/B of the paired wgrad launch (ocn_gemm_tn_accum2: out-proj + QKV weight gradients of a block in one launch) against two single
launches (developer knob 13 = 1) at the bench's shapes.  Run through gpurun:  python tools/ab_tn_pair.py > gpurun_out/tn_pair.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


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


print(f"{'shape':34s} {'paired ms':>10s} {'two launches ms':>16s} {'paired TF/s':>12s} {'two TF/s':>9s}")
for name, M, C in (("img  dW[768,768] + dW[2304,768]", 4096 * 50, 768), ("txt  dW[512,512] + dW[1536,512] (packed)", 177803, 512),
                   ("txt  dW[512,512] + dW[1536,512] (dense)", 4096 * 77, 512)):
    g = torch.Generator(device=dev).manual_seed(1)
    a1, b1 = torch.randn(M, C, device=dev, generator=g).bfloat16(), torch.randn(M, C, device=dev, generator=g).bfloat16()
    a2, b2 = torch.randn(M, 3 * C, device=dev, generator=g).bfloat16(), torch.randn(M, C, device=dev, generator=g).bfloat16()
    dw1, dw2 = torch.zeros(C, C, device=dev), torch.zeros(3 * C, C, device=dev)
    db1, db2 = torch.zeros(C, device=dev), torch.zeros(3 * C, device=dev)
    best = [1e9, 1e9]
    for rnd in range(4):
        for i, knob in enumerate((0, 1)):
            _lib.call("ocn_set_tuning", 13, knob)
            best[i] = min(best[i], timeit(lambda: ops.gemm_tn_accum2(a1, b1, dw1, db1, a2, b2, dw2, db2)))
    _lib.call("ocn_set_tuning", 13, 0)
    fl = 2.0 * M * 4 * C * C / 1e9
    print(f"{name:34s} {best[0]:10.4f} {best[1]:16.4f} {fl / best[0]:12.0f} {fl / best[1]:9.0f}", flush=True)
