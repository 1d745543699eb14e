"""Attention at the 257-token shapes of BASELINE configs 4 / 5 (ViT-L-14: head_dim 64, 16 heads, local batch 2048; ViT-H-14: head_dim 80,
16 heads, local batch 1024): the streamed generic kernels (csrc/attention_generic.hip) and, for head_dim 64, the head-resident kernels
of csrc/attention.hip (developer knob 7 = 1 forces the generic path; knob 2 = 4 / 5 selects the two-pass / one-pass resident backward).
Every variant is checked against fp32 torch on the first sequences.  usage: python tools/ab_attn_long.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

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


def ref(qkv, dout, B, L, H, D):
    C = H * D
    x = qkv.float().requires_grad_(True)
    q, k, v = x.reshape(B, L, 3, H, D).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-1, -2)) * D ** -0.5, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * L, C)
    o.backward(dout.float())
    return o.detach(), x.grad


def case(name, B, L, H, D, variants):
    C = H * D
    g = torch.Generator(device=dev).manual_seed(3)
    qkv = (torch.randn(B * L, 3 * C, device=dev, generator=g) * 1.2).bfloat16()
    dout = torch.randn(B * L, C, device=dev, generator=g).bfloat16()
    nref = 8
    o_ref, d_ref = ref(qkv[:nref * L], dout[:nref * L], nref, L, H, D)
    flops_f, flops_b = 4.0 * L * L * D * H * B, 10.0 * L * L * D * H * B
    by_f, by_b = B * L * C * 2 * 4, B * L * C * 2 * 8
    for vname, k7, k2 in variants:
        _lib.call("ocn_set_tuning", 7, k7)
        _lib.call("ocn_set_tuning", 2, k2)
        out, lse = ops.attn_fwd(qkv, B, L, H, False, D ** -0.5, D)
        dq = ops.attn_bwd(qkv, out, dout, lse, B, L, H, False, D ** -0.5, D)
        eo = float((out[:nref * L].float() - o_ref).norm() / o_ref.norm())
        ed = float((dq[:nref * L].float() - d_ref).norm() / d_ref.norm())
        tf = timeit(lambda: ops.attn_fwd(qkv, B, L, H, False, D ** -0.5, D))
        tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, False, D ** -0.5, D))
        print(f"{name:28s} {vname:30s} fwd {tf:7.3f} ms ({flops_f / tf / 1e9:5.0f} TF/s, {by_f / tf / 1e9:4.2f} TB/s)  bwd {tb:7.3f} ms ({flops_b / tb / 1e9:5.0f} TF/s, "
              f"{by_b / tb / 1e9:4.2f} TB/s)  rel err out {eo:.1e} dqkv {ed:.1e}", flush=True)
    _lib.call("ocn_set_tuning", 7, 0)
    _lib.call("ocn_set_tuning", 2, 0)


case("ViT-H-14 image B1024 L257 hd80", 1024, 257, 16, 80, [("streamed generic", 0, 0)])
case("ViT-L-14 image B2048 L257 hd64", 2048, 257, 16, 64, [("resident, one-pass bwd", 2, 5), ("resident, two-pass bwd", 2, 4), ("streamed generic", 1, 0)])
case("hd128 B256 L257 H8", 256, 257, 8, 128, [("streamed generic", 0, 0)])
case("hd96 B256 L100 H8", 256, 100, 8, 96, [("streamed generic", 0, 0)])
case("ViT-B-32 image B4096 L50 hd64", 4096, 50, 12, 64, [("resident, one-pass bwd", 2, 5), ("resident, two-pass bwd", 2, 4), ("streamed generic", 1, 0)])
