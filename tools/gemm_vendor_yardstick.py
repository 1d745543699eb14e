"""Per-shape yardstick (tools only, never the product): the hand-written GEMM kernels against torch.matmul (hipBLASLt / rocBLAS on ROCm)
on the 16 NT shapes (8 forward + 8 dgrad) and the 8 wgrad shapes of the ViT-B-32 step at local batch 4096, on the same GPU, plain bf16
output on both sides (what this silicon gives on these shapes with the vendor's kernels: VERDICT r2, weak #7).
usage: python tools/gemm_vendor_yardstick.py > gpurun_out/gemm_vs_vendor.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
MI, MT = 4096 * 50, 177803


def timeit(fn, iters=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def main():
    print(f"# {torch.cuda.get_device_name(0)}; torch {torch.__version__}; TFLOP/s = 2 M N K / time; 'native' = ocn_gemm_nt(EPI_BF16) / ocn_gemm_tn_accum")
    print(f"{'shape':34s} {'native ms':>10s} {'TF/s':>7s} {'vendor ms':>10s} {'TF/s':>7s} {'native/vendor':>14s}")
    tot_n = tot_v = 0.0
    for tag, M, C in (("img", MI, 768), ("txt", MT, 512)):
        nt = [("qkv", 3 * C, C), ("out_proj", C, C), ("c_fc", 4 * C, C), ("c_proj", C, 4 * C),
              ("d c_proj (dGELU input)", 4 * C, C), ("d c_fc", C, 4 * C), ("d out_proj", C, C), ("d qkv", C, 3 * C)]
        for name, N, K in nt:
            a = torch.randn(M, K, device=dev).bfloat16()
            b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            t_n = timeit(lambda: ops.gemm_nt(ops.EPI_BF16, a, b, out))
            bt = b.t()
            t_v = timeit(lambda: torch.matmul(a, bt, out=out))
            fl = 2.0 * M * N * K / 1e9
            tot_n += t_n
            tot_v += t_v
            print(f"NT {tag} {name:24s} [{M}x{N}x{K}] {t_n:8.3f} {fl / t_n:7.0f} {t_v:10.3f} {fl / t_v:7.0f} {t_v / t_n:14.2f}")
            del a, b, out
        for name, N, K in [("wgrad qkv", 3 * C, C), ("wgrad out_proj", C, C), ("wgrad c_fc", 4 * C, C), ("wgrad c_proj", C, 4 * C)]:
            a = torch.randn(M, N, device=dev).bfloat16()
            b = torch.randn(M, K, device=dev).bfloat16()
            dw = torch.zeros(N, K, device=dev)
            t_n = timeit(lambda: ops.gemm_tn_accum(a, b, dw))
            at = a.t()
            o16 = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
            t_v = timeit(lambda: torch.matmul(at, b, out=o16))
            fl = 2.0 * M * N * K / 1e9
            tot_n += t_n
            tot_v += t_v
            print(f"TN {tag} {name:24s} [{M}x{N}x{K}] {t_n:8.3f} {fl / t_n:7.0f} {t_v:10.3f} {fl / t_v:7.0f} {t_v / t_n:14.2f}")
            del a, b, dw, o16
    print(f"# sum over the 24 shapes: native {tot_n:.2f} ms, vendor {tot_v:.2f} ms (vendor / native = {tot_v / tot_n:.2f}); the native kernels additionally "
          "fuse bias / GELU / residual / fp32 accumulation into dW, which the vendor calls would need extra passes for")


if __name__ == "__main__":
    main()
