"""Where the accumulation path's extra time goes (developer tool; gpurun): the reference's --accum-freq semantics (train.py:236-311) evaluate the loss on the
CONCATENATION of all micro-batches' features for every micro-batch.  Times NativeClipLoss forward + backward at [4096 x 4096] (a plain step) and at
[32768 x 32768] (every micro-batch of --accum-freq 8 at local batch 4096), E = 512, and the gradient `+=` of a second backward into existing .grad."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd.loss import NativeClipLoss  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=5):
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


loss_fn = NativeClipLoss()
for N in (4096, 32768):
    g = torch.Generator(device=dev).manual_seed(N)
    I = torch.nn.functional.normalize(torch.randn(N, 512, device=dev, generator=g), dim=-1).requires_grad_(True)
    T = torch.nn.functional.normalize(torch.randn(N, 512, device=dev, generator=g), dim=-1).requires_grad_(True)
    s = torch.tensor(14.28, device=dev, requires_grad=True)

    def step():
        I.grad = T.grad = s.grad = None
        loss_fn(I, T, s).backward()

    ms = timeit(step)
    print(f"NativeClipLoss forward + backward at [{N} x {N}] x E 512: {ms:.3f} ms  ({12.0 * N * N * 512 / ms / 1e9:.0f} TFLOP/s over 6 GEMM-equivalents of 2 N N E)", flush=True)
# the `+=` of a second micro-batch's gradients: 302 tensors, 151 M elements (what autograd's AccumulateGrad does when .grad exists)
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.synth import init_state_dict  # noqa: E402
shapes = [v.shape for v in init_state_dict(get_model_config("ViT-B-32"), seed=0).values()]
a = [torch.randn(sh, device=dev) for sh in shapes]
b = [torch.randn(sh, device=dev) for sh in shapes]


def accumulate():
    for x, y in zip(a, b):
        x.add_(y)


print(f"gradient accumulation `+=` over the {len(shapes)} parameter tensors of ViT-B-32 ({sum(x.numel() for x in a) / 1e6:.0f} M elements): {timeit(accumulate):.3f} ms", flush=True)
