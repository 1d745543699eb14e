"""packed vs dense text-tower attention alone on the stream (B 4096, H 8, L 77, synthetic caption lengths)"""
import sys, torch
sys.path.insert(0, ".")
from open_clip_amd import ops, _lib
from open_clip_amd.configs import get_model_config
from open_clip_amd.synth import synthetic_batch
_lib.load()
dev = torch.device("cuda:0")
B, L, H = 4096, 77, 8
C = H * 64
text = synthetic_batch(get_model_config("ViT-B-32"), B, seed=1234)["text"].to(dev)
eot, plan, last, _ = ops.seq_pack_plan(text)
seq_off = plan[:text.shape[0] + 1]
M = int(seq_off[-1])


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, rows, so in (("dense", B * L, None), ("packed", M, seq_off)):
    qkv = torch.randn(rows, 3 * C, device=dev).bfloat16()
    dout = torch.randn(rows, C, device=dev).bfloat16()
    out, lse = ops.attn_fwd(qkv, B, L, H, True, 0.125, seq_off=so)
    tf = timeit(lambda: ops.attn_fwd(qkv, B, L, H, True, 0.125, seq_off=so))
    tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, True, 0.125, seq_off=so))
    by_f = rows * 4 * C * 2
    by_b = rows * (3 * C + C + C + 3 * C) * 2
    print(f"{name:7s} rows {rows:7d}: fwd {tf:7.1f} us ({by_f / tf / 1e6:6.2f} TB/s)  bwd {tb:7.1f} us ({by_b / tb / 1e6:6.2f} TB/s)")
# sorted by length (longest first): does the order of workgroups matter?
lens = (eot + 1).long()
order = torch.argsort(lens, descending=True)
so2 = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), lens[order].cumsum(0)]).to(torch.int32)
qkv = torch.randn(M, 3 * C, device=dev).bfloat16()
dout = torch.randn(M, C, device=dev).bfloat16()
out, lse = ops.attn_fwd(qkv, B, L, H, True, 0.125, seq_off=so2)
print("packed, longest first: fwd %.1f us bwd %.1f us" % (timeit(lambda: ops.attn_fwd(qkv, B, L, H, True, 0.125, seq_off=so2)),
                                                         timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, True, 0.125, seq_off=so2))))
for knob6 in (1,):
    _lib.call("ocn_set_tuning", 6, knob6)
    out, lse = ops.attn_fwd(qkv, B, L, H, True, 0.125, seq_off=seq_off)
    print("packed, generic bwd kernel (knob 6=1): bwd %.1f us" % timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, True, 0.125, seq_off=seq_off)))
    _lib.call("ocn_set_tuning", 6, 0)
