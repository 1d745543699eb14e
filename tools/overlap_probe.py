"""Developer probe: does an HBM-bound kernel that fits NEXT TO a wgrad workgroup on a CU (<= 64 VGPRs, <= 32 KiB LDS) really run
under the MFMA-bound wgrad GEMM when the two are enqueued on different HIP streams?  LayerNorm forward (44 VGPRs, no LDS) is
the partner; the result decides whether a register-lean LayerNorm backward is worth writing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K, C = 4096 * 50, 3072, 768, 768
a = torch.randn(M, N, device=dev).bfloat16()
b = torch.randn(M, K, device=dev).bfloat16()
dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
x = torch.randn(M, C, device=dev)
w, bb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def tn():
    ops.gemm_tn_accum(a, b, dw, db)


def ln(n):
    for _ in range(n):
        ops.layernorm_fwd(x, w, bb)


def timed(fn, iters=5):
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


def both(n, ln_first):
    def run():
        side.wait_stream(main)
        if ln_first:
            ln(n)
            with torch.cuda.stream(side):
                tn()
        else:
            with torch.cuda.stream(side):
                tn()
            ln(n)
        main.wait_stream(side)
    return run


t_tn = timed(tn)
for n in (1, 2, 4, 6):
    t_ln = timed(lambda: ln(n))
    t_a = timed(both(n, False))
    t_b = timed(both(n, True))
    print(f"wgrad {t_tn:.3f} ms | {n} x LN fwd {t_ln:.3f} ms | sum {t_tn + t_ln:.3f} | concurrent: wgrad enqueued first {t_a:.3f}, LN enqueued first {t_b:.3f}", flush=True)
