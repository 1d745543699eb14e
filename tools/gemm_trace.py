"""Developer tool: per-tile timeline of the persistent NT kernel (variant bit 64 selects the DBG build, which logs
100 MHz wall-clock stamps of workgroups 0 and 133).  usage: gemm_trace.py M N K [epi]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).bfloat16()
b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
f32out = epi in (2, 4)
out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
resid = torch.randn(M, N, device=dev) if epi == 2 else None
aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (1, 3) else None
bias = torch.randn(N, device=dev)
for variant in (5, 5, 5 + 256 * 64):
    _lib.call("ocn_set_gemm_variant", variant)
    ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux)
torch.cuda.synchronize()
_lib.call("ocn_set_gemm_variant", 0)
lib = _lib.load()
buf = (ctypes.c_longlong * 1024)()
lib.ocn_debug_nt5_trace.argtypes = [ctypes.c_void_p]
assert lib.ocn_debug_nt5_trace(buf) == 0
raw = torch.tensor(list(buf), dtype=torch.int64)
print(f"M={M} N={N} K={K} epi={epi}")
for blk, base in ((0, 0), (133, 512)):
    t = raw[base:base + 64].view(8, 8).double() / 100.0  # us
    t0 = t[0, 0]
    print(f"block {blk}: per tile [start, mainloop end, epilogue end] (us since first tile start)")
    for i in range(8):
        print(f"  tile {i}: {float(t[i,0]-t0):8.2f} {float(t[i,3]-t0):8.2f} {float(t[i,4]-t0):8.2f}   main {float(t[i,3]-t[i,0]):6.2f}  epi {float(t[i,4]-t[i,3]):6.2f}"
              f"  [bias + DMA drain {float(t[i,5]-t[i,3]):5.2f} | first half {float(t[i,6]-t[i,5]):5.2f} | second half {float(t[i,4]-t[i,6]):5.2f}]")
