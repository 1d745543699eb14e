"""Developer sweeps on the MI355X (run through gpurun): tile-walk band width of the persistent NT GEMM, epilogue VALU
cost, and attention-backward ablations (what the loads / the arithmetic / the stores cost on their own)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
Mi, Mt = 4096 * 50, 4096 * 77


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def nt_sweep():
    cases = [("img qkv", Mi, 2304, 768, [0], [9, 5, 3]), ("img fc", Mi, 3072, 768, [0, 1, 3], [12, 6, 4, 3, 2]),
             ("img proj", Mi, 768, 3072, [0, 2], [3, 1]), ("img dh1", Mi, 768, 2304, [0], [3, 1]),
             ("txt qkv", Mt, 1536, 512, [0], [6, 3]), ("txt fc", Mt, 2048, 512, [0, 1, 3], [8, 4, 2]),
             ("glogits", 32768, 32768, 512, [4], [128, 10, 8, 4])]
    for name, M, N, K, epis, bands in cases:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        for epi in epis:
            f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
            resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
            aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
            row = []
            for band in bands + [0]:
                _lib.call("ocn_set_gemm_variant", 5 | (band << 16))
                ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
                row.append(f"band {band:3d}: {2.0 * M * N * K / ms / 1e9:6.0f} TF/s {ms:.3f} ms")
            if epi in (1, 3):  # no-VALU epilogue at the automatic band
                _lib.call("ocn_set_gemm_variant", 5 | (1 << 8))
                ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
                row.append(f"no-gelu-math: {2.0 * M * N * K / ms / 1e9:6.0f} TF/s")
            print(f"{name:9s} epi {epi}: " + " | ".join(row), flush=True)
            del out, resid, aux
        del a, b
    _lib.call("ocn_set_gemm_variant", 0)


def attn_sweep():
    for name, B, L, H, causal in [("img", 4096, 50, 12, False), ("txt", 4096, 77, 8, True)]:
        C = H * 64
        qkv = torch.randn(B * L, 3 * C, device=dev).bfloat16()
        do = torch.randn(B * L, C, device=dev).bfloat16()
        out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125)
        gb_f = B * L * C * 2 * 4 / 1e9
        row = []
        for knob in (0, 2, 0, 2):  # developer knob 9 = 2: non-temporal policy for the forward's LDS-DMA loads
            _lib.call("ocn_set_tuning", 9, knob)
            ms_f = timeit(lambda: ops.attn_fwd(qkv, B, L, H, causal, 0.125))
            row.append(f"fwd[{'nt' if knob else 'default'}] {ms_f:.3f} ms ({gb_f / ms_f:.2f} TB/s)")
        _lib.call("ocn_set_tuning", 9, 0)
        gb_b = B * L * C * 2 * 8 / 1e9
        for mask in (0, 1, 2, 4, 3, 5, 6, 7):
            _lib.call("ocn_set_tuning", 1, mask)
            ms = timeit(lambda: ops.attn_bwd(qkv, out, do, lse, B, L, H, causal, 0.125))
            row.append(f"bwd[-{'L' if mask & 1 else ''}{'C' if mask & 2 else ''}{'S' if mask & 4 else ''}] {ms:.3f} ms")
        _lib.call("ocn_set_tuning", 1, 0)
        print(f"attn {name} L={L}: " + " | ".join(row) + f"   (bwd algorithmic {gb_b:.2f} GB)", flush=True)


def attn_wpe_sweep():
    """attention backward: one-pass (knob 2 = 5) / two-pass dK, dV (4) / default (0) builds"""
    for name, B, L, H, causal in [("img", 4096, 50, 12, False), ("txt", 4096, 77, 8, True)]:
        C = H * 64
        qkv = torch.randn(B * L, 3 * C, device=dev).bfloat16()
        do = torch.randn(B * L, C, device=dev).bfloat16()
        out, lse = ops.attn_fwd(qkv, B, L, H, causal, 0.125)
        row, ref = [], None
        for wpe in (5, 4, 0):
            _lib.call("ocn_set_tuning", 2, wpe)
            got = ops.attn_bwd(qkv, out, do, lse, B, L, H, causal, 0.125)
            ref = got if ref is None else ref
            ms = timeit(lambda: ops.attn_bwd(qkv, out, do, lse, B, L, H, causal, 0.125))
            row.append(f"knob2={wpe}: {ms:.3f} ms{'' if torch.equal(got, ref) else ' (MISMATCH)'}")
        _lib.call("ocn_set_tuning", 2, 0)
        if causal:  # barrier-free 5-image causal kernel (default) against the generic one
            _lib.call("ocn_set_tuning", 6, 1)
            got = ops.attn_bwd(qkv, out, do, lse, B, L, H, causal, 0.125)
            ms = timeit(lambda: ops.attn_bwd(qkv, out, do, lse, B, L, H, causal, 0.125))
            _lib.call("ocn_set_tuning", 6, 0)
            row.append(f"generic kernel: {ms:.3f} ms (max |diff| vs causal kernel {float((got.float() - ref.float()).abs().max()):.2e})")
        print(f"attn bwd {name} L={L}: " + " | ".join(row), flush=True)
        row = []
        for extra_kb in (0, 8, 20, 45):  # fewer resident workgroups per CU: does the arithmetic-only time scale with occupancy?
            _lib.call("ocn_set_tuning", 5, extra_kb)
            for mask in (0, 5):
                _lib.call("ocn_set_tuning", 1, mask)
                ms = timeit(lambda: ops.attn_bwd(qkv, out, do, lse, B, L, H, causal, 0.125))
                row.append(f"+{extra_kb}KB LDS {'compute-only' if mask else 'full'} {ms:.3f}")
        _lib.call("ocn_set_tuning", 5, 0)
        _lib.call("ocn_set_tuning", 1, 0)
        print("     occupancy probe: " + " | ".join(row), flush=True)


def tn_sweep():
    """weight-gradient kernel with and without its atomic epilogue"""
    for name, M, N, K in [("img qkv", Mi, 2304, 768), ("img out", Mi, 768, 768), ("img fc", Mi, 3072, 768), ("img proj", Mi, 768, 3072),
                          ("txt qkv", Mt, 1536, 512), ("txt out", Mt, 512, 512), ("txt fc", Mt, 2048, 512), ("txt proj", Mt, 512, 2048)]:
        a = torch.randn(M, N, device=dev).bfloat16()
        b = torch.randn(M, K, device=dev).bfloat16()
        dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        row = []
        for ab in (0, 1):
            _lib.call("ocn_set_tuning", 4, ab)
            ms = timeit(lambda: ops.gemm_tn_accum(a, b, dw, db))
            row.append(f"{'no-epilogue' if ab else 'full'} {ms:.3f} ms ({2.0 * M * N * K / ms / 1e9:5.0f} TF/s)")
        _lib.call("ocn_set_tuning", 4, 0)
        for det in (False, True, False, True):  # the reproducible form: per-split slabs + an ordered reduce pass instead of atomics
            ms = timeit(lambda: ops.gemm_tn_accum(a, b, dw, db, deterministic=det))
            row.append(f"{'deterministic' if det else 'atomics'} {ms:.3f} ms")
        print(f"tn {name:9s} dW[{N},{K}]: " + " | ".join(row), flush=True)
        del a, b


def ntstore_sweep():
    """cache policy of the epilogue: knob bit 2 flips non-temporal stores (default: on for the GELU epilogue only), bit 8 makes
    the loads of the residual / saved operand non-temporal"""
    cases = [("img qkv", Mi, 2304, 768, [0]), ("img dh2", Mi, 768, 3072, [0]), ("img da", Mi, 768, 768, [0]), ("img dh1", Mi, 768, 2304, [0]),
             ("img fc", Mi, 3072, 768, [1, 3]), ("img out", Mi, 768, 768, [2]), ("img proj", Mi, 768, 3072, [2]),
             ("txt qkv", Mt, 1536, 512, [0]), ("txt dh2", Mt, 512, 2048, [0]), ("txt fc", Mt, 2048, 512, [1, 3]), ("txt out", Mt, 512, 512, [2])]
    for name, M, N, K, epis in cases:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        for epi in epis:
            f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
            resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
            aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
            row = []
            for mask in (0, 16, 0, 16, 0, 16):
                _lib.call("ocn_set_gemm_variant", 5 | (mask << 8))
                ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
                row.append(f"{ {0: 'default', 16: 'A-nt'}[mask]} {2.0 * M * N * K / ms / 1e9:5.0f}")
            print(f"{name:9s} epi {epi} (TF/s): " + " | ".join(row), flush=True)
            del out, resid, aux
        del a, b
    _lib.call("ocn_set_gemm_variant", 0)


def ln_sweep():
    """LayerNorm backward with / without the non-temporal policy on its read-once operands (developer knob 8)"""
    for M, C in [(Mi, 768), (Mt, 512)]:
        x, dres = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
        dy = torch.randn(M, C, device=dev).bfloat16()
        w = torch.ones(C, device=dev)
        mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
        dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        row = []
        gb = M * C * 16 / 1e9
        for knob in (0, 1, 0, 1):
            _lib.call("ocn_set_tuning", 8, knob)
            ms = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres, want_f32=True, want_bf16=True))
            row.append(f"{'default' if knob else 'non-temporal'} {ms:.3f} ms ({gb / ms:.2f} TB/s)")
        _lib.call("ocn_set_tuning", 8, 0)
        print(f"ln_bwd [{M}x{C}]: " + " | ".join(row), flush=True)


def percu_sweep():
    """persistent NT GEMM launched with k workgroups per CU (developer knob 10), steady-state timing on the step's shapes"""
    cases = [("img qkv", Mi, 2304, 768, 0), ("img dh2", Mi, 768, 3072, 0), ("img da", Mi, 768, 768, 0), ("img dh1", Mi, 768, 2304, 0),
             ("img fc", Mi, 3072, 768, 1), ("img dgelu", Mi, 3072, 768, 3), ("img out", Mi, 768, 768, 2), ("img proj", Mi, 768, 3072, 2),
             ("txt qkv", Mt, 1536, 512, 0), ("txt dh2", Mt, 512, 2048, 0), ("txt da", Mt, 512, 512, 0), ("txt dh1", Mt, 512, 1536, 0),
             ("txt fc", Mt, 2048, 512, 1), ("txt dgelu", Mt, 2048, 512, 3), ("txt out", Mt, 512, 512, 2), ("txt proj", Mt, 512, 2048, 2)]
    tot = {}
    for name, M, N, K, epi in cases:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
        resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
        aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
        row = []
        for k in (1, 2, 3, 4, 1, 2):
            _lib.call("ocn_set_tuning", 10, k)
            ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
            row.append(f"{k}/CU {ms:.3f} ms")
            tot[k] = tot.get(k, 0.0) + ms
        print(f"{name:10s} epi {epi}: " + " | ".join(row), flush=True)
        del a, b, out, resid, aux
    _lib.call("ocn_set_tuning", 10, 0)
    print("sum over shapes (1/CU and 2/CU measured twice): " + ", ".join(f"{k}/CU {v:.3f} ms" for k, v in tot.items()), flush=True)


def stagger_sweep():
    """start-phase stagger of the persistent NT kernel (developer build; us per phase class; 0 = 63 = off -- the product's setting since round 5 --,
    62 = the old automatic rule: csrc/gemm_nt5.hip::nt5_stagger)"""
    cases = [("img fc", Mi, 3072, 768, [0, 1, 3]), ("img out", Mi, 768, 768, [2]), ("img proj", Mi, 768, 3072, [0, 2]),
             ("txt fc", Mt, 2048, 512, [0, 1, 3]), ("txt out", Mt, 512, 512, [2]), ("txt proj", Mt, 512, 2048, [2]), ("img qkv", Mi, 2304, 768, [0])]
    for name, M, N, K, epis in cases:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        for epi in epis:
            f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
            resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
            aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
            row = []
            for st in (63, 2, 4, 6, 8, 12, 0):
                _lib.call("ocn_set_gemm_variant", 5 | (st << 21))
                ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
                row.append(f"st {st:2d}: {2.0 * M * N * K / ms / 1e9:5.0f}")
            print(f"{name:9s} epi {epi} (TF/s): " + " | ".join(row), flush=True)
            del out, resid, aux
        del a, b
    _lib.call("ocn_set_gemm_variant", 0)


def epi_ablation_sweep():
    """what the epilogue of the persistent NT kernel costs on top of the main loop: developer knobs 32 (stores dropped by a
    zero-sized descriptor), 128 (epilogue operand loads dropped), 1 (GELU arithmetic skipped) -- the upper bound of what hiding
    the epilogue under the next tile's main loop can buy (results of the ablated runs are wrong by construction)"""
    cases = [("img qkv", Mi, 2304, 768, [0]), ("img da", Mi, 768, 768, [0]), ("img fc", Mi, 3072, 768, [0, 1, 3]), ("img out", Mi, 768, 768, [2]),
             ("img proj", Mi, 768, 3072, [2]), ("txt fc", Mt, 2048, 512, [0, 1, 3]), ("txt out", Mt, 512, 512, [2]), ("txt proj", Mt, 512, 2048, [2])]
    for name, M, N, K, epis in cases:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        for epi in epis:
            f32out = epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_F32)
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
            resid = torch.randn(M, N, device=dev) if epi == ops.EPI_BIAS_RESID_F32 else None
            aux = torch.randint(0, 253, (M, N), device=dev, dtype=torch.uint8) if epi in (ops.EPI_BIAS_GELU, ops.EPI_DGELU) else None
            row = []
            for mask, tag in ((0, "full"), (0x80000, "stores->L2 window"), (32, "-stores"), (32 | 128, "-stores-loads"), (32 | 128 | 1, "-stores-loads-valu"), (0, "full")):
                _lib.call("ocn_set_gemm_variant", 5 | (mask << 8))
                ms = timeit(lambda: ops.gemm_nt(epi, a, b, out, bias=bias, resid=resid, aux=aux))
                row.append(f"{tag} {ms:.3f} ms ({2.0 * M * N * K / ms / 1e9:5.0f})")
            print(f"{name:9s} epi {epi}: " + " | ".join(row), flush=True)
            del out, resid, aux
        del a, b
    _lib.call("ocn_set_gemm_variant", 0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "attn"):
        attn_sweep()
    if what in ("all", "nt"):
        nt_sweep()
    if what in ("all", "percu"):
        percu_sweep()
    if what in ("all", "ln"):
        ln_sweep()
    if what in ("all", "ntstore"):
        ntstore_sweep()
    if what in ("all", "stagger"):
        stagger_sweep()
    if what in ("all", "attnw"):
        attn_wpe_sweep()
    if what in ("all", "tn"):
        tn_sweep()
    if what in ("all", "epiabl"):
        epi_ablation_sweep()
