"""Runs ONE GEMM configuration a few times (for rocprofv3 --pmc / --kernel-trace).  usage: gemm_one.py kind variant epi M N K [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

kind, variant, epi, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 4
dev = torch.device("cuda:0")
_lib.call("ocn_set_gemm_variant", variant)
if kind == "nt":
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    f32out = epi in (2, 4)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    resid = torch.randn(M, N, device=dev) if epi == 2 else None
    aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (1, 3) else None
    bias = torch.randn(N, device=dev)
    for _ in range(iters):
        ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux)
else:
    a = torch.randn(M, N, device=dev).bfloat16()
    b = torch.randn(M, K, device=dev).bfloat16()
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    for _ in range(iters):
        ops.gemm_tn_accum(a, b, dw, db)
torch.cuda.synchronize()
