"""Config-3 logits: NativeClipLoss at the sizes one rank of the 8-GPU node sees -- the row-sharded global loss ([4096 x 32768] rows both
ways, one-rank communicator: the distributed code path with identity collectives), the redundant [32768 x 32768] form (--naive-global-loss) and the
plain [4096 x 4096] step -- forward + backward, wall time by HIP events; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
python tools/logits_probe.py [forms...]   forms: sharded naive local"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd.comm import NativeComm  # noqa: E402
from open_clip_amd.loss import NativeClipLoss  # noqa: E402

dev = torch.device("cuda:0")
forms = sys.argv[1:] or ["sharded", "naive", "local"]
B, W, E = 4096, 8, 512


def timeit(fn, iters=4):
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


g = torch.Generator(device=dev).manual_seed(7)
feats = lambda n: torch.nn.functional.normalize(torch.randn(n, E, device=dev, generator=g), dim=-1).requires_grad_(True)
s = torch.tensor(14.28, device=dev, requires_grad=True)
if "sharded" in forms:
    # one rank's share of the 8-rank global loss: emulate the gather by handing the loss a world of 1 with N = 32768 "remote" rows is not possible through the
    # public module -- so time the two PairTerms the branch evaluates, with the same calls (loss.py::_ClipLossFn, row_sharded branch)
    from open_clip_amd.loss import _term
    I, T = feats(B).detach(), feats(B).detach()
    I_all, T_all = feats(W * B).detach(), feats(W * B).detach()
    sd = s.detach().reshape(1)
    acc = torch.zeros(2, device=dev)

    def sharded():
        N = W * B
        ti = _term(I, T_all, sd, False).compute_logits()
        tt = _term(T, I_all, sd, False).compute_logits()
        for term in (ti, tt):
            term.softmax_ce(B * 3, 0.5 / N, 0.5 / N, acc)
        dI, dT = ti.dX(), tt.dX()
        through_cols = torch.zeros(N, 2 * E, device=dev)
        tt.dY(into=through_cols[:, :E])
        ti.dY(into=through_cols[:, E:])
        return through_cols, dI, dT

    ms = timeit(sharded)
    fl = 2 * 3 * 2 * B * (W * B) * E  # per direction: one logits pass + dX + dY (round 6; two logits passes until round 5)
    print(f"row-sharded global ClipLoss, one rank's work at R {B} x N {W * B} x E {E}: {ms:.3f} ms; 6 GEMMs of 2 R N E = {fl / 1e12:.2f} TFLOP (one logits pass + dX + dY "
          f"per direction: executed = algorithmic) = {fl / ms / 1e9:.0f} TFLOP/s = {fl / ms / 1e9 / 2500:.3f} of the MFMA peak", flush=True)
if "naive" in forms:
    I, T = feats(W * B), feats(W * B)
    loss_fn = NativeClipLoss()

    def naive():
        I.grad = T.grad = s.grad = None
        loss_fn(I, T, s).backward()

    ms = timeit(naive, iters=2)
    N = W * B
    print(f"redundant global ClipLoss at N {N} x N {N} x E {E} (every rank of --naive-global-loss; every micro-batch of --accum-freq 8): {ms:.3f} ms; "
          f"algorithmic 6 GEMMs of 2 N N E: {12 * N * N * E / ms / 1e9:.0f} TFLOP/s = {12 * N * N * E / ms / 1e9 / 2500:.3f} of the MFMA peak", flush=True)
if "local" in forms:
    I, T = feats(B), feats(B)
    loss_fn = NativeClipLoss()

    def local():
        I.grad = T.grad = s.grad = None
        loss_fn(I, T, s).backward()

    ms = timeit(local)
    print(f"ClipLoss of the plain step at {B} x {B} x E {E}: {ms:.3f} ms; algorithmic 6 GEMMs: {12 * B * B * E / ms / 1e9:.0f} TFLOP/s", flush=True)
