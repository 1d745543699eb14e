"""Where does the native step's gradient error at the bench's batch come from?  (developer tool; run through gpurun)

One fp32 GPU reference (oracle/gpu_fp32.py) of the ViT-B-32 step at batch B, then, against it:
  * the native step in its execution variants (single-query / all-query pooled last block, full last block, dense text tower, one stream);
  * the same step as eager PyTorch operators under autocast(bf16) (oracle/torch_eager.py) -- the reference's own amp_bf16 policy -- as is, and with
    ONE native rounding grafted in at a time: the GELU derivative saved in 8-bit fixed point / in bf16 (instead of recomputed from the bf16
    pre-activation), attention probabilities rounded to bf16 in front of P.V, logits kept in fp32.
Prints median / worst rel-L2 of the 1-D and of the matrix gradients per variant.  usage: python tools/parity_ablation.py [B]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.synth import init_state_dict, synthetic_batch  # noqa: E402
from oracle import gpu_fp32, torch_eager  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = get_model_config("ViT-B-32")
state = init_state_dict(cfg, seed=0, perturb=True)
batch = synthetic_batch(cfg, B, seed=1234)
outs, ref = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=512)
ref = {k: v.cpu() for k, v in ref.items()}
torch.cuda.empty_cache()
WATCH = ["transformer.resblocks.11.ln_1.bias", "ln_final.bias", "transformer.resblocks.11.mlp.c_proj.bias", "transformer.resblocks.5.mlp.c_proj.bias",
         "visual.transformer.resblocks.11.mlp.c_proj.bias", "visual.transformer.resblocks.5.ln_1.bias", "text_projection", "transformer.resblocks.11.mlp.c_fc.weight"]


def report(tag, grads):
    rel = {k: float((grads[k].float().cpu() - ref[k]).norm() / ref[k].norm().clamp_min(1e-30)) for k in ref if k in grads}
    one = sorted(v for k, v in rel.items() if ref[k].ndim <= 1)
    two = sorted(v for k, v in rel.items() if ref[k].ndim >= 2)
    txt = sorted(v for k, v in rel.items() if ref[k].ndim <= 1 and not k.startswith("visual."))
    img = sorted(v for k, v in rel.items() if ref[k].ndim <= 1 and k.startswith("visual."))
    print(f"{tag:44s} 1-D med {one[len(one) // 2]:.2e} worst {one[-1]:.2e} (text med {txt[len(txt) // 2]:.2e}, image med {img[len(img) // 2]:.2e}) | "
          f"matrices med {two[len(two) // 2]:.2e} worst {two[-1]:.2e} | " + " ".join(f"{rel[k]:.2e}" for k in WATCH), flush=True)


print("watch columns: " + " | ".join(WATCH))
# ---- native variants ----
from tests.test_model_gpu import _build, _step  # noqa: E402
for tag, opts in [("native (shipped)", {}), ("native pooled_single_query=False", {"psq": False}), ("native pooled_last_block=False", {"plb": False}),
                  ("native pack_text=False", {"pack": False}), ("native tower_streams=False", {"ts": False})]:
    model = _build(cfg, state)
    if "psq" in opts:
        model.pooled_single_query = False
    if "plb" in opts:
        model.pooled_last_block = model.visual.pooled_last_block = False
    if "pack" in opts:
        model.pack_text = False
    if "ts" in opts:
        model.tower_streams = False
    _step(model, batch)
    report(tag, {k: p.grad for k, p in model.named_parameters()})
    del model
    torch.cuda.empty_cache()


# ---- eager autocast variants ----
_GELU, _SDPA = F.gelu, F.scaled_dot_product_attention


class GeluSavedDerivative(torch.autograd.Function):
    """forward exact erf-GELU; backward multiplies by the derivative as the NATIVE path saves it: evaluated in fp32 from the fp32 pre-activation in the
    forward epilogue, then stored in `mode` precision"""
    @staticmethod
    def forward(ctx, x, mode):
        xf = x.float()
        d = 0.5 * (1 + torch.erf(xf / math.sqrt(2.0))) + xf * torch.exp(-0.5 * xf * xf) / math.sqrt(2 * math.pi)
        if mode == "q8":
            d = (torch.round((d + 0.13) * 200.0).to(torch.uint8))
        elif mode == "bf16":
            d = d.bfloat16()
        ctx.mode = mode
        ctx.save_for_backward(d)
        return _GELU(xf).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (d,) = ctx.saved_tensors
        d = d.float() / 200.0 - 0.13 if ctx.mode == "q8" else d.float()
        return (dy.float() * d).to(dy.dtype), None


def eager(tag, gelu_mode=None, p_bf16=False, fp32_logits=False, chunk=1024):
    dev = torch.device("cuda:0")
    try:
        if gelu_mode:
            torch_eager.F.gelu = lambda x: GeluSavedDerivative.apply(x, gelu_mode)
        if p_bf16:
            def sdpa(q, k, v, attn_mask=None, scale=None):
                s = (q.float() @ k.float().transpose(-1, -2)) * scale
                if attn_mask is not None:
                    s = s + attn_mask.float()
                p = torch.softmax(s, dim=-1).to(torch.bfloat16)
                return (p @ v.to(torch.bfloat16)).to(q.dtype)
            torch_eager.F.scaled_dot_product_attention = sdpa
        model = torch_eager.EagerCLIP(cfg, state).to(dev).train()
        img, txt = batch["image"].to(dev), batch["text"].to(dev)
        with torch.amp.autocast("cuda", dtype=torch.bfloat16):
            i, t, s = model(img, txt)
            if not fp32_logits:
                loss = torch_eager.clip_loss(i, t, s)
        if fp32_logits:
            loss = torch_eager.clip_loss(i.float(), t.float(), s.float())
        loss.backward()
        torch.cuda.synchronize()
        report(tag, {k.replace("/", "."): p.grad for k, p in model.p.items() if p.grad is not None})
        del model
    finally:
        torch_eager.F.gelu, torch_eager.F.scaled_dot_product_attention = _GELU, _SDPA
        torch.cuda.empty_cache()


eager("eager autocast(bf16)")
eager("eager + gelu' saved in 8-bit fixed point", gelu_mode="q8")
eager("eager + gelu' saved in bf16", gelu_mode="bf16")
eager("eager + P rounded to bf16 (explicit attention)", p_bf16=True)
eager("eager + fp32 logits / cross-entropy", fp32_logits=True)
eager("eager + all three", gelu_mode="q8", p_bf16=True, fp32_logits=True)
