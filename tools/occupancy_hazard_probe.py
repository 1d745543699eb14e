"""Developer probe: what does a persistent GEMM cost when a few CUs are held by another stream's kernels at its launch (the
situation of a multi-GPU step: collective kernels of the gradient all-reduce sit on CUs while the backward's GEMMs start)?
`ocn_debug_occupy` parks n workgroups (140 KiB of LDS each, so nothing of the GEMM fits beside them) for a few ms on a side
stream; the GEMM is then timed on the main stream with one (default) and two (developer knob 10) workgroups per CU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
Mi = 4096 * 50
side = torch.cuda.Stream()
sink = torch.zeros(1, dtype=torch.int32, device=dev)


def run(fn, occupy, iters=3):
    best = 1e9
    for _ in range(iters):
        torch.cuda.synchronize()
        if occupy:
            _lib.call("ocn_debug_occupy", occupy, 4000, sink.data_ptr(), side.cuda_stream)
            torch.cuda._sleep(200000)  # let the occupier land first
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for name, M, N, K, epi in [("img fc", Mi, 3072, 768, 0), ("img fc gelu", Mi, 3072, 768, 1), ("img dh2", Mi, 768, 3072, 0)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    aux = torch.empty(M, N, device=dev, dtype=torch.uint8) if epi == 1 else None
    bias = torch.randn(N, device=dev)
    fn = lambda: ops.gemm_nt(epi, a, b, out, bias=bias, aux=aux)
    fn()
    row = []
    for per_cu in (1, 2, 3):
        _lib.call("ocn_set_tuning", 10, per_cu)
        for occ in (0, 1, 8, 32):
            row.append(f"{per_cu}/CU, {occ:2d} CUs held: {run(fn, occ):.3f} ms")
    _lib.call("ocn_set_tuning", 10, 0)
    print(f"NT {name}: " + " | ".join(row), flush=True)
    ops.set_tile_rescue(True)  # the product's multi-GPU form: static shares, finishers hand out the shares of workgroups that have not started
    print(f"NT {name}, tile rescue: " + " | ".join(f"{occ:2d} CUs held: {run(fn, occ):.3f} ms" for occ in (0, 1, 8, 32, 64)), flush=True)
    ops.set_tile_rescue(False)
    del a, b, out, aux

M, N, K = Mi, 3072, 768
a = torch.randn(M, N, device=dev).bfloat16()
b = torch.randn(M, K, device=dev).bfloat16()
dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
fn = lambda: ops.gemm_tn_accum(a, b, dw, db)
fn()
for k in (1, 2, 3):
    _lib.call("ocn_set_tuning", 11, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"wgrad img fc, {k} per CU: steady {e0.elapsed_time(e1) / 5:.3f} ms | " + " | ".join(f"{occ:2d} CUs held: {run(fn, occ):.3f} ms" for occ in (0, 1, 8, 32)), flush=True)
_lib.call("ocn_set_tuning", 11, 0)
ops.set_tile_rescue(True)
fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"wgrad img fc, tile rescue: steady {e0.elapsed_time(e1) / 5:.3f} ms | " + " | ".join(f"{occ:2d} CUs held: {run(fn, occ):.3f} ms" for occ in (0, 1, 8, 32, 64)), flush=True)
ops.set_tile_rescue(False)
