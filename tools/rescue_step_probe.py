"""Developer probe: gradients of one ViT-B-32 step (batch 256, bf16 image stream) under {static, rescue} x {no CU held, 24 CUs held}, against a static
run: which parameters differ by more than the run-to-run noise of the wgrads' atomics, and under which condition."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd import _lib, ops  # noqa: E402
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.loss import NativeClipLoss  # noqa: E402
from open_clip_amd.model import NativeCLIP  # noqa: E402
from open_clip_amd.synth import init_state_dict, synthetic_batch  # noqa: E402

cfg = get_model_config("ViT-B-32")
state = init_state_dict(cfg, seed=3)
batch = synthetic_batch(cfg, 256, seed=5)
side, sink = torch.cuda.Stream(), torch.zeros(1, dtype=torch.int32, device="cuda:0")
kw = dict(image_stream=os.environ.get("STREAM", "bf16"))
if os.environ.get("SERIAL"):
    kw["overlap_towers"] = False


def step(rescue, held):
    ops.set_tile_rescue(rescue)
    m = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=True, **kw)
    m.load_state_dict(state, strict=True)
    m = m.cuda().train()
    torch.cuda.synchronize()
    if held:
        _lib.call("ocn_debug_occupy", held, 40000, sink.data_ptr(), side.cuda_stream)
        torch.cuda._sleep(60000)
    out = m(image=batch["image"].cuda(), text=batch["text"].cuda())
    loss = NativeClipLoss()(**out)
    loss.backward()
    torch.cuda.synchronize()
    return out["image_features"].float(), out["text_features"].float(), float(loss), {k: p.grad.float() for k, p in m.named_parameters()}


a = step(False, 0)
for rep in range(int(os.environ.get("REPS", "4"))):
    for rescue, held in ((False, 0), (False, 24), (True, 0), (True, 24)):
        b = step(rescue, held)
        bad = []
        for k in a[3]:
            d = (a[3][k] - b[3][k]).norm().item() / max(a[3][k].norm().item(), 1e-30)
            if d > 1e-5:
                bad.append((d, k))
        bad.sort(reverse=True)
        print(f"rep {rep} rescue {int(rescue)} held {held:2d}: features equal {torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])}, loss diff {abs(a[2] - b[2]):.2e}, "
              f"{len(bad)} of {len(a[3])} gradients differ by > 1e-5 rel" + ("; worst: " + ", ".join(f"{k} {d:.1e}" for d, k in bad[:6]) if bad else ""), flush=True)
