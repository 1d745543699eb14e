"""Per-shape timing of the NT GEMMs of the ViT-B-32 step (local batch 4096, packed text rows), each with the epilogue the step runs it with.
One library per process (OCN_LIB_PATH selects it: the product build, the developer build, or an older build kept under _ab/), so an A/B is
two or more runs of this script, alternated by the calling shell script.  Calls the C ABI with raw pointers: works with any build.
usage: [OCN_LIB_PATH=...] python tools/ab_nt.py [--knob MASK] [--only SUBSTR] [--json PATH]
  --knob  developer ablation mask of the persistent NT kernel (ocn_set_gemm_variant bits 8+; developer build only), e.g. 2 / 8 flip the
          epilogue's store / load cache policy"""
import argparse
import json
import os
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--knob", type=int, default=0)
ap.add_argument("--only", default="")
ap.add_argument("--json", default="")
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE", help="ocn_set_tuning(KEY, VALUE) before the run (developer knobs)")
args = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
MI, MT = 4096 * 50, 177803
# (name, epilogue, M, N, K): 0 plain bf16, 1 bias + GELU (+ saved gelu'), 2 bias + fp32 residual, 3 x saved gelu'
SHAPES = [
    ("img qkv", 0, MI, 2304, 768), ("img out_proj+res", 2, MI, 768, 768), ("img c_fc+gelu", 1, MI, 3072, 768), ("img c_proj+res", 2, MI, 768, 3072),
    ("img d c_proj dgelu", 3, MI, 3072, 768), ("img d c_fc", 0, MI, 768, 3072), ("img d out_proj", 0, MI, 768, 768), ("img d qkv", 0, MI, 768, 2304),
    ("txt qkv", 0, MT, 1536, 512), ("txt out_proj+res", 2, MT, 512, 512), ("txt c_fc+gelu", 1, MT, 2048, 512), ("txt c_proj+res", 2, MT, 512, 2048),
    ("txt d c_proj dgelu", 3, MT, 2048, 512), ("txt d c_fc", 0, MT, 512, 2048), ("txt d out_proj", 0, MT, 512, 512), ("txt d qkv", 0, MT, 512, 1536),
]


def timeit(fn, iters):
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


if args.knob:
    _lib.call("ocn_set_gemm_variant", args.knob << 8)
for kv in args.tuning:
    k, v = kv.split("=")
    _lib.call("ocn_set_tuning", int(k), int(v))
st = torch.cuda.current_stream().cuda_stream
print(f"# library: {_lib.LIB_PATH}  knob {args.knob}  tuning {args.tuning}")
res = {}
tot = 0.0
for name, epi, M, N, K in SHAPES:
    if args.only and args.only not in name:
        continue
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.bfloat16)
    bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == 2 else None
    aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (1, 3) else None
    fn = lambda: _lib.call("ocn_gemm_nt", epi, a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), N, M, N, K, bias.data_ptr() if epi in (1, 2) else 0,
                           0 if resid is None else resid.data_ptr(), 0 if aux is None else aux.data_ptr(), 1.0, st)
    ms = timeit(fn, args.iters)
    tot += ms
    res[name] = ms
    print(f"{name:20s} epi {epi} [{M}x{N}x{K}] {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:6.0f} TF/s", flush=True)
    del a, b, out, aux, resid
print(f"# sum {tot:.3f} ms")
if args.json:
    with open(args.json, "a") as f:
        f.write(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "knob": args.knob, "tuning": args.tuning, "ms": res, "sum": tot}) + "\n")
