"""Developer tool: per-tile timeline of the persistent NT kernel (ablation bit 64 logs 100 MHz wall-clock stamps of
workgroups 0 and 133).  usage: gemm_trace.py M N K [epi]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).bfloat16()
b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dbg = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)  # aux doubles as the debug buffer (EPI 0 ignores it otherwise)
for _ in range(3):
    _lib.call("ocn_set_gemm_variant", 5)
    ops.gemm_nt(0, a, b, out)
_lib.call("ocn_set_gemm_variant", 5 + 256 * 64)
_lib.call("ocn_gemm_nt", 0, a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), N, M, N, K, 0, 0, dbg.data_ptr(), 1.0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
_lib.call("ocn_set_gemm_variant", 0)
raw = dbg.view(torch.int64).flatten()[:1024].cpu()
for blk, base in ((0, 0), (133, 512)):
    t = raw[base:base + 64].view(8, 8).double() / 100.0  # us
    t0 = t[0, 0]
    print(f"block {blk}: per tile [start, after Ktile0, after Ktile1, mainloop end, epilogue end] (us since first tile start)")
    for i in range(8):
        print("  tile %d: " % i + " ".join(f"{float(x - t0):8.2f}" for x in t[i, :5]) + f"   main {float(t[i,3]-t[i,0]):6.2f}  epi {float(t[i,4]-t[i,3]):6.2f}  kt0 {float(t[i,1]-t[i,0]):5.2f} kt1 {float(t[i,2]-t[i,1]):5.2f}")
