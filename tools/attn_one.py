"""One attention shape of the bench, forward + backward, N times (a workload for rocprofv3 counter passes).
usage: python tools/attn_one.py [image|text] [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "image"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator(device=dev).manual_seed(1)
if which == "image":
    B, L, H, causal, lay = 4096, 50, 12, False, None
    M = B * L
else:
    B, L, H, causal = 4096, 77, 8, True
    gl = torch.Generator().manual_seed(1234)
    lens = torch.randint(8, 77, (B,), generator=gl) + 1
    off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
    M = int(off[-1])
    nb = (lens + 31) // 32
    lay = ops.SeqLayout(off.to(torch.int32).to(dev), torch.sort(nb, stable=True).indices.to(torch.int32).to(dev),
                        torch.bincount(nb - 1, minlength=(L + 31) // 32).tolist())
C = H * 64
qkv = (torch.randn(M, 3 * C, device=dev, generator=g) * 1.5).bfloat16()
dout = torch.randn(M, C, device=dev, generator=g).bfloat16()
for _ in range(n):
    out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125, seq_off=lay)
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, causal, 0.125, seq_off=lay)
torch.cuda.synchronize()
print("done", which, M)
