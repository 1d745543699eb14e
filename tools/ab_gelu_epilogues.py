"""A/B of the GELU / dGELU epilogue GEMMs between two builds of the library (this tree: gelu' saved in 8 bits; --root another tree, e.g.
the round-2 one with a bf16 gelu').  Calls the C ABI directly (raw pointers), so the same script drives either build.
usage: python tools/ab_gelu_epilogues.py [--root PATH]"""
import argparse
import os
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--root", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = ap.parse_args()
sys.path.insert(0, args.root)
from open_clip_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
MI, MT = 4096 * 50, 177803


def timeit(fn, iters=8):
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


st = torch.cuda.current_stream().cuda_stream
print(f"# library: {_lib.LIB_PATH}")
for name, epi, M, N, K in [("img c_fc + GELU", 1, MI, 3072, 768), ("img dGELU", 3, MI, 3072, 768), ("txt c_fc + GELU", 1, MT, 2048, 512), ("txt dGELU", 3, MT, 2048, 512),
                           ("img c_proj + resid", 2, MI, 768, 3072), ("img plain dgrad", 0, MI, 768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.bfloat16)
    bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == 2 else None
    aux = torch.randint(0, 120, (M, N * 2), device=dev, dtype=torch.uint8)  # big enough for a bf16 or a u8 image; any bit pattern is a finite value in both
    fn = lambda: _lib.call("ocn_gemm_nt", epi, a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), N, M, N, K, bias.data_ptr() if epi != 3 else 0,
                           0 if resid is None else resid.data_ptr(), aux.data_ptr() if epi in (1, 3) else 0, 1.0, st)
    ms = timeit(fn)
    print(f"{name:20s} [{M}x{N}x{K}] {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:6.0f} TF/s", flush=True)
    del a, b, out, aux, resid
