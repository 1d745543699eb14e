"""which call sites issue the step's zero-fill kernels?  (developer tool; gpurun)  One ViT-B-32 training step at batch 4096 under torch.profiler with
stacks; prints aten::zero_ / aten::fill_ / aten::copy_ calls grouped by the innermost open_clip_amd frame."""
import collections
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.loss import NativeClipLoss  # noqa: E402
from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of  # noqa: E402
from open_clip_amd.synth import init_state_dict, synthetic_batch  # noqa: E402
from tests.test_model_gpu import _build  # noqa: E402

cfg = get_model_config("ViT-B-32")
model = _build(cfg, init_state_dict(cfg, seed=0))
batch = synthetic_batch(cfg, 4096, seed=1, device="cuda")
opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=1e-5, betas=(0.9, 0.98), eps=1e-6, weight_caches=weight_caches_of(model))
loss_fn = NativeClipLoss()


def step():
    opt.zero_grad(set_to_none=True)
    loss_fn(**model(image=batch["image"], text=batch["text"])).backward()
    opt.step()
    with torch.no_grad():
        model.logit_scale.clamp_(0, math.log(100))


step(); step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous", "aten::cat", "aten::mul_", "aten::add_", "aten::sub_"):
        site = next((f for f in ev.stack if "open_clip_amd" in f or "bench.py" in f), ev.stack[0] if ev.stack else "?")
        cnt[(ev.name, site.strip()[-110:])] += 1
for (name, site), n in cnt.most_common(60):
    print(f"{n:5d} {name:18s} {site}")
