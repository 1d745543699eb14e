"""Is the loss backward where the native step's gradient error enters?  (developer tool; gpurun)  ViT-B-32 at batch B against the fp32 GPU reference:
(1) gradient of the loss with respect to the features -- native step, eager autocast step; (2) the native loss ALONE on the reference's exact fp32
features (isolates the fused logits + cross-entropy kernels from the towers' forward error); (3) the same for eager's loss expression."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.loss import NativeClipLoss  # noqa: E402
from open_clip_amd.synth import init_state_dict, synthetic_batch  # noqa: E402
from oracle import gpu_fp32, torch_eager  # noqa: E402
from tests.test_model_gpu import _build  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = get_model_config("ViT-B-32")
state = init_state_dict(cfg, seed=0, perturb=True)
batch = synthetic_batch(cfg, B, seed=1234)
outs, ref = gpu_fp32.step_reference(cfg, state, batch["image"], batch["text"], chunk=512)
dI, dT = outs["d_image_features"], outs["d_text_features"]
rel = lambda a, b: float((a.float() - b).norm() / b.norm())
print(f"reference: |dI| {float(dI.norm()):.3e} |dT| {float(dT.norm()):.3e}; row norms: mean {float(dI.norm(dim=1).mean()):.3e}; feature spread: "
      f"|I - mean| {float((outs['image_features'] - outs['image_features'].mean(0)).norm(dim=1).mean()):.3f} |T - mean| {float((outs['text_features'] - outs['text_features'].mean(0)).norm(dim=1).mean()):.3f}")
s = torch.tensor(float(state["logit_scale"]), device="cuda").exp()
# (2) the native loss alone on exact features
I = outs["image_features"].clone().requires_grad_(True)
T = outs["text_features"].clone().requires_grad_(True)
loss = NativeClipLoss()(I, T, s)
loss.backward()
print(f"native loss on the reference's fp32 features: loss {float(loss):.6f} (ref {float(outs['loss']):.6f}); dI rel {rel(I.grad, dI):.3e}  dT rel {rel(T.grad, dT):.3e}")
# (3) eager's loss expression under autocast on exact features
I2 = outs["image_features"].clone().requires_grad_(True)
T2 = outs["text_features"].clone().requires_grad_(True)
with torch.amp.autocast("cuda", dtype=torch.bfloat16):
    l2 = torch_eager.clip_loss(I2, T2, s)
l2.backward()
print(f"eager autocast loss on the same features:     loss {float(l2):.6f}; dI rel {rel(I2.grad, dI):.3e}  dT rel {rel(T2.grad, dT):.3e}")
I3 = outs["image_features"].clone().requires_grad_(True)
T3 = outs["text_features"].clone().requires_grad_(True)
l3 = torch_eager.clip_loss(I3, T3, s)
l3.backward()
print(f"eager fp32 loss on the same features:         loss {float(l3):.6f}; dI rel {rel(I3.grad, dI):.3e}  dT rel {rel(T3.grad, dT):.3e}")
# (1) whole steps
model = _build(cfg, state)
out = model(image=batch["image"].cuda(), text=batch["text"].cuda())
out["image_features"].retain_grad()
out["text_features"].retain_grad()
NativeClipLoss()(**out).backward()
print(f"native step: features rel {rel(out['image_features'], outs['image_features']):.3e} / {rel(out['text_features'], outs['text_features']):.3e}; "
      f"dI rel {rel(out['image_features'].grad, dI):.3e}  dT rel {rel(out['text_features'].grad, dT):.3e}")
del model, out
torch.cuda.empty_cache()
em = torch_eager.EagerCLIP(cfg, state).cuda().train()
with torch.amp.autocast("cuda", dtype=torch.bfloat16):
    i, t, sc = em(batch["image"].cuda(), batch["text"].cuda())
    i.retain_grad()
    t.retain_grad()
    le = torch_eager.clip_loss(i, t, sc)
le.backward()
print(f"eager autocast step: features rel {rel(i, outs['image_features']):.3e} / {rel(t, outs['text_features']):.3e}; dI rel {rel(i.grad, dI):.3e}  dT rel {rel(t.grad, dT):.3e}")


# ---- structure of the feature-gradient error: what part of it is COMMON to all rows (adds coherently in every sum over the batch)? ----
def structure(tag, g, gref, feat):
    e = (g.float() - gref)
    cm = e.mean(0)
    frac_cm = float(cm.norm() * (e.shape[0] ** 0.5) / e.norm())
    radial = (e * feat).sum(1)  # component along the row's own (unit) feature
    frac_rad = float(radial.norm() / e.norm())
    ref_cm = float(gref.mean(0).norm() * (gref.shape[0] ** 0.5) / gref.norm())
    print(f"{tag:28s} error rel {float(e.norm() / gref.norm()):.3e}; common-mode share of the error {frac_cm:.3f} (of the reference gradient itself {ref_cm:.3f}); "
          f"radial share {frac_rad:.3f}; |sum_b e_b| / |sum_b g_b| = {float(e.sum(0).norm() / gref.sum(0).norm()):.3e}")


model = _build(cfg, state)
out = model(image=batch["image"].cuda(), text=batch["text"].cuda())
out["image_features"].retain_grad()
out["text_features"].retain_grad()
NativeClipLoss()(**out).backward()
structure("native step dI", out["image_features"].grad, dI, outs["image_features"])
structure("native step dT", out["text_features"].grad, dT, outs["text_features"])
structure("native loss-only dI", I.grad, dI, outs["image_features"])
structure("native loss-only dT", T.grad, dT, outs["text_features"])
structure("eager loss-only dI", I2.grad, dI, outs["image_features"])
structure("eager loss-only dT", T2.grad, dT, outs["text_features"])
structure("eager step dI", i.grad, dI, outs["image_features"])
structure("eager step dT", t.grad, dT, outs["text_features"])


# ---- emulation of the native loss arithmetic in torch, one rounding at a time (exact fp32 features in) ----
def emu(tag, round_g, fix_diag=False, round_ops=True, g_two_sums=False):
    X, Y = outs["image_features"], outs["text_features"]
    bfr = (lambda z: z.bfloat16().float()) if round_ops else (lambda z: z)
    n = X.shape[0]
    gs = 1.0 / (2 * n)
    dX = torch.zeros_like(X)
    dY = torch.zeros_like(Y)
    for (A, Bm, dA, dB) in ((X, Y, dX, dY), (Y, X, dY, dX)):  # li = s X Y^T (rows: images), lt = s Y X^T
        As, B16 = bfr(A * s), bfr(Bm)
        logits = As @ B16.t()
        P = torch.softmax(logits, dim=-1)
        G = (P - torch.eye(n, device=P.device)) * gs
        Gr = G.bfloat16().float() if round_g else G
        if fix_diag:  # what the bf16 rounding of the diagonal entry dropped, carried separately in fp32
            dfix = (G.diagonal() - Gr.diagonal())
        dA += s * (Gr @ B16)
        dB += Gr.t() @ As
        if fix_diag:
            dA += s * dfix[:, None] * B16
            dB += dfix[:, None] * As
    structure(tag + " dI", dX, dI, outs["image_features"])
    structure(tag + " dT", dY, dT, outs["text_features"])


emu("emu fp32 G, bf16 operands", round_g=False)
emu("emu bf16 G", round_g=True)
emu("emu bf16 G + fp32 diagonal fix", round_g=True, fix_diag=True)
emu("emu bf16 G, fp32 operands", round_g=True, round_ops=False)
