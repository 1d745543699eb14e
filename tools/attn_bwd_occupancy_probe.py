"""What more resident workgroups would buy the image tower's attention backward (developer tool; gpurun; OCN_LIB_PATH = the developer library).
NOTE: the two probe instantiations this script selects (knob 2 = 6 / 7: `attn_bwd_kernel<256, 4, true, ALIAS>` in csrc/attention.hip, ALIAS = the dO image
aliases the V image) existed in the tree only for the measurement of profiles/r05_attention_backward_occupancy_probe.txt (commit 'attention backward: the
8-workgroups-per-CU route probed'); they were removed again so that the kernel sources -- and with them the hash the PMC traffic record is tied to -- stay those
of the measured library.  With today's library knobs 6 / 7 select the shipped kernel.
knob 2 = 0: shipped (two-pass, 158 registers, three LDS images: six 2-wave workgroups per CU); 6: the same kernel held to 128 registers (20 spilled; still six
per CU: LDS-bound); 7: 128 registers AND two LDS images (dO aliases V: results WRONG, timing only): eight workgroups per CU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
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


B, L, H = 4096, 50, 12
C = H * 64
g = torch.Generator(device=dev).manual_seed(1)
qkv = (torch.randn(B * L, 3 * C, device=dev, generator=g) * 1.5).bfloat16()
dout = torch.randn(B * L, C, device=dev, generator=g).bfloat16()
out, lse = ops.attn_fwd(qkv, B, L, H, False, 0.125)
nbytes = B * L * C * 2 * 8
for knob, what in ((0, "shipped: 158 registers, 3 LDS images (6 workgroups / CU)"), (6, "128 registers (20 spilled), 3 images (6 / CU)"),
                   (7, "128 registers, 2 images: 8 / CU (results wrong)"), (0, "shipped again")):
    _lib.call("ocn_set_tuning", 2, knob)
    ms = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, False, 0.125))
    print(f"image attention backward B4096 L50 H12  {what:62s} {ms:.4f} ms  {nbytes / ms / 1e9:5.2f} TB/s", flush=True)
_lib.call("ocn_set_tuning", 2, 0)
