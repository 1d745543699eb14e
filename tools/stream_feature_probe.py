"""Developer probe (round 6): what the image tower's bf16 residual stream does to the FEATURES at the bench batch, against the fp32 GPU reference --
rel-L2, max-abs and the common-mode part (error of the batch mean relative to the error's rms) -- for the native fp32 / bf16 streams and for eager
autocast, and what that does to the gradient the text tower receives (dL/dT).  python tools/stream_feature_probe.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.synth import init_state_dict, synthetic_batch  # noqa: E402
from oracle import gpu_fp32  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = get_model_config("ViT-B-32")
state = init_state_dict(cfg, seed=0, perturb=True)
batch = synthetic_batch(cfg, B, seed=1234)
ref, rgr = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=512)
amp, agr = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=512, amp=True)


def stats(tag, f, key="image_features"):
    r = ref[key].float()
    e = f.float() - r
    rel = float(e.norm() / r.norm())
    cm = float(e.mean(0).norm() / (e.norm() / e.shape[0] ** 0.5))
    print(f"{tag:28s} {key:16s} rel_l2 {rel:.3e} max_abs {float(e.abs().max()):.3e} common-mode |mean_b e| / rms_b|e| = {cm:.3f}  (1/sqrt(B) = {B ** -0.5:.3f} if independent)")


stats("eager amp_bf16 (chunked)", amp["image_features"])
stats("eager amp_bf16 (chunked)", amp["text_features"], "text_features")
stats("eager amp_bf16 (chunked)", amp["d_text_features"], "d_text_features")
from tests.test_model_gpu import _build, _step  # noqa: E402
for stream in ("fp32", "bf16", "bf16-fp32grad"):
    m = _build(cfg, state, image_stream=stream)
    out, loss = _step(m, batch)
    stats(f"native {stream}", out["image_features"].detach())
    stats(f"native {stream}", out["text_features"].detach(), "text_features")
    for k in ("token_embedding.weight", "transformer.resblocks.0.ln_1.weight", "visual.conv1.weight", "visual.class_embedding"):
        g = dict(m.named_parameters())[k].grad.float()
        print(f"   {k}: rel_l2 {float((g - rgr[k]).norm() / rgr[k].norm()):.3e} (eager {float((agr[k] - rgr[k]).norm() / rgr[k].norm()):.3e})")
    del m, out, loss
    torch.cuda.empty_cache()
