"""Developer probe (run through gpurun): can the HBM-bound kernels of a block backward (LayerNorm backward, attention backward) and the
MFMA-bound weight-gradient GEMM share the chip by CU PARTITION (two streams created with hipExtStreamCreateWithCUMask) better than by
co-residency (profiles/r01_wgrad_ln_overlap_probe.txt: tails only -- a persistent GEMM workgroup owns its CU's registers and LDS)?
Prints, per partition, each kernel alone on its CU subset and both together, against back-to-back on the whole chip."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
Mi, Mt = 4096 * 50, 4096 * 77
NCU = 256


def masked_stream(cus):
    """stream restricted to the CUs in `cus` (iterable of CU indices)"""
    words = [0] * (NCU // 32)
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    out = ctypes.c_void_p()
    _lib.call("ocn_debug_stream_with_cu_mask", ctypes.cast(arr, ctypes.c_void_p), len(words), ctypes.cast(ctypes.byref(out), ctypes.c_void_p))
    return torch.cuda.ExternalStream(out.value, device=dev)


def timed(fn, stream, iters=6):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def both(fa, sa, fb, sb, iters=6):
    """fa on stream sa and fb on stream sb, launched together `iters` times with a join after each pair"""
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        sa.wait_stream(main)
        sb.wait_stream(main)
        with torch.cuda.stream(sa):
            fa()
        with torch.cuda.stream(sb):
            fb()
        main.wait_stream(sa)
        main.wait_stream(sb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    # image tower block: LN backward [Mi x 768], attention backward (B 4096, L 50, H 12), wgrad c_fc dW[3072,768] over Mi
    M, C, H, L, B = Mi, 768, 12, 50, 4096
    x, dres = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
    dy = torch.randn(M, C, device=dev).bfloat16()
    w = torch.ones(C, device=dev)
    mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    qkv = torch.randn(M, 3 * C, device=dev).bfloat16()
    do = torch.randn(M, C, device=dev).bfloat16()
    out, lse = ops.attn_fwd(qkv, B, L, H, False, 0.125)
    a = torch.randn(M, 4 * C, device=dev).bfloat16()
    b = torch.randn(M, C, device=dev).bfloat16()
    dW, dB = torch.zeros(4 * C, C, device=dev), torch.zeros(4 * C, device=dev)

    def ln():
        ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres, want_f32=True, want_bf16=True)

    def attn():
        ops.attn_bwd(qkv, out, do, lse, B, L, H, False, 0.125)

    def wgrad():
        ops.gemm_tn_accum(a, b, dW, dB)

    full = torch.cuda.current_stream()
    t_ln, t_at, t_wg = timed(ln, full), timed(attn, full), timed(wgrad, full)
    print(f"whole chip, alone: ln_bwd {t_ln:.3f} ms | attn_bwd {t_at:.3f} ms | wgrad c_fc {t_wg:.3f} ms | back to back: ln+wgrad {t_ln + t_wg:.3f}, attn+wgrad {t_at + t_wg:.3f}", flush=True)
    side = torch.cuda.Stream()
    print(f"two unrestricted streams (co-residency): ln||wgrad {both(ln, full, wgrad, side):.3f} ms | attn||wgrad {both(attn, full, wgrad, side):.3f} ms", flush=True)
    for n_hbm in (32, 64, 96, 128):
        # spread both subsets over all 8 XCDs: CU index c belongs to the HBM-kernel subset when (c % 8) < n_hbm / 32 ... in units of 8
        k = n_hbm // 32  # of every 8 consecutive CU indices, k go to the HBM-bound stream
        hbm_cus = [c for c in range(NCU) if (c % 8) < k]
        mf_cus = [c for c in range(NCU) if (c % 8) >= k]
        s_h, s_m = masked_stream(hbm_cus), masked_stream(mf_cus)
        _lib.call("ocn_set_tuning", 15, len(mf_cus))
        a_ln, a_at, a_wg = timed(ln, s_h), timed(attn, s_h), timed(wgrad, s_m)
        p1, p2 = both(ln, s_h, wgrad, s_m), both(attn, s_h, wgrad, s_m)
        _lib.call("ocn_set_tuning", 15, 0)
        print(f"{n_hbm:3d} CUs HBM-bound / {NCU - n_hbm} CUs wgrad: alone ln {a_ln:.3f} attn {a_at:.3f} wgrad {a_wg:.3f} | together ln||wgrad {p1:.3f} "
              f"(serial whole-chip {t_ln + t_wg:.3f}) attn||wgrad {p2:.3f} (serial {t_at + t_wg:.3f})", flush=True)


if __name__ == "__main__":
    main()
