"""What the atomic epilogue of the weight-gradient GEMM costs (developer tool; gpurun): the wgrad shapes of the ViT-B-32 step at local batch 4096, each
timed as shipped and with the epilogue's fp32 atomics skipped (knob 4 = 1: results wrong).  Round 5 used it for an experiment that is NOT in the tree:
every M-split of a tile starting its atomics at another 32 x 32 block (eight copies of the unrolled epilogue behind a scalar branch on split & 7) -- no
gain (text shapes -1 %, image c_fc +5 %): the epilogue is bound by the atomic THROUGHPUT of the L2s (64 MB of fp32 atomics per launch at ~1.5 TB/s),
not by same-address serialisation (profiles/r05_tn5_epilogue_rotation.txt).  Also prints TF/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


MI, MT = 4096 * 50, 177803
print(f"{'shape (dW[N,K] += A[M,N]^T B[M,K])':44s} {'shipped ms':>10s} {'TF/s':>6s} {'again ms':>14s} {'no atomics ms':>14s} {'epilogue share':>15s}")
for name, M, N, K in (("img c_fc   [204800 x 3072 x 768]", MI, 3072, 768), ("img c_proj [204800 x 768 x 3072]", MI, 768, 3072), ("img qkv    [204800 x 2304 x 768]", MI, 2304, 768),
                      ("txt c_fc   [177803 x 2048 x 512]", MT, 2048, 512), ("txt c_proj [177803 x 512 x 2048]", MT, 512, 2048), ("txt qkv    [177803 x 1536 x 512]", MT, 1536, 512)):
    g = torch.Generator(device=dev).manual_seed(1)
    a, b = torch.randn(M, N, device=dev, generator=g).bfloat16(), torch.randn(M, K, device=dev, generator=g).bfloat16()
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    t = []
    for knob in (0, 0, 1):  # shipped, twice | no atomics  (knob 2 selected the un-rotated order while the rotation experiment was in the tree)
        _lib.call("ocn_set_tuning", 4, knob)
        t.append(timeit(lambda: ops.gemm_tn_accum(a, b, dw, db)))
    _lib.call("ocn_set_tuning", 4, 0)
    print(f"{name:44s} {t[0]:10.4f} {2.0 * M * N * K / t[0] / 1e9:6.0f} {t[1]:14.4f} {t[2]:14.4f} {100 * (t[0] - t[2]) / t[0]:14.1f}%", flush=True)
    del a, b, dw, db
