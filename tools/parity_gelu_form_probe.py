"""Which GELU arithmetic does the whole-step parity at size prefer?  (developer tool; gpurun; OCN_LIB_PATH = the developer library)
Runs tests/test_parity_at_size_gpu.py::_whole_step_case for ViT-H-14 + SigLIP at batch 512 (where d/d logit_scale is a cancelling sum at the noise
floor of the bf16 policy) and ViT-B-32 ClipLoss at batch 4096 twice in separate processes: polynomial-CDF form (product) / Abramowitz-Stegun form
(knob 0x400000).  usage: OCN_LIB_PATH=open_clip_amd/libopenclip_hip_dev.so python tools/parity_gelu_form_probe.py {poly|as} {h14|b32}"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_clip_amd import _lib  # noqa: E402

form, which = sys.argv[1], sys.argv[2]
if form == "as":
    _lib.call("ocn_set_gemm_variant", 0x400000 << 8)
from tests import test_parity_at_size_gpu as T  # noqa: E402

print(f"##### GELU form: {form}  case: {which}", flush=True)
if which == "h14":
    T._whole_step_case("ViT-H-14", 512, True, 64, True, None, f"ViT-H-14 SigLIP ckpt,B512,gelu={form}", eager_whole_batch=False)
else:
    T._whole_step_case("ViT-B-32", 4096, False, 512, False, 2e-2, f"ViT-B-32,B4096,gelu={form}", eager_whole_batch=False)
