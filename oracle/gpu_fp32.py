"""Whole-step fp32 reference ON THE GPU for batches the CPU oracle cannot reach -- TEST INFRASTRUCTURE ONLY.

``oracle/clip_oracle.py`` (pinned against the reference's own outputs, tests/golden/) takes minutes of host time at batch 512 and would take
the better part of an hour at the bench's batch of 4096.  This file evaluates the SAME training step -- both towers, ClipLoss or SigLipLoss, backward -- in
pure fp32 (no autocast, TF32 off) with the library operators of ``oracle/torch_eager.py::EagerCLIP`` on the MI355X, in batch CHUNKS so that
fp32 activations of 4096 pairs never have to exist at once:

    1. features of every chunk under no_grad                                  -> I [B,E], T [B,E]
    2. the loss on the full batch from those features, with gradient          -> loss, dL/dI, dL/dT, dL/dlogit_scale   (loss.py:91-141)
    3. every chunk again with gradient, backward of <I_c, dL/dI_c> + <T_c, dL/dT_c>   -> parameter gradients accumulate

which is exact: the towers are per-sample maps, so dL/dtheta = sum_c J_c^T dL/dfeatures_c (the same identity the reference's own
--accum-freq path rests on, train.py:236-311).  It is pinned, not trusted: tests/test_bench_size_gpu.py first checks it against the CPU oracle
at batch 512 (features, loss and all 302 gradients to <= 1e-4), then uses it as the reference of the native step at batch 4096; the SigLIP form and
the ViT-L-14 / ViT-H-14 configurations are pinned the same way at batches the host can do by tests/test_parity_at_size_gpu.py before they are used.
Never imported by the product (``open_clip_amd``)."""
import torch

from oracle.torch_eager import EagerCLIP, clip_loss, siglip_loss


def step_reference(cfg, state, image, text, chunk=512, siglip=False, amp=False):
    """-> (outs, grads): outs = image_features / text_features [B,E] fp32, loss; grads keyed by the reference's state-dict names.
    ``image`` / ``text`` may live on the host: chunks are moved to the GPU one at a time.
    ``siglip``: SigLipLoss (loss.py:356-367, world_size 1: -sum logsigmoid(labels * (s I T^T + b)) / B) instead of ClipLoss; the state then
    carries ``logit_bias``.  ``amp``: NOT a reference -- the same chunked evaluation under ``torch.amp.autocast(bf16)`` (towers and loss), i.e.
    the reference's own --precision amp_bf16 policy as eager operators, for models whose whole-batch eager activations do not fit: the
    yardstick the native gradients' error is compared with (tests/test_parity_at_size_gpu.py), never a checker."""
    dev = torch.device("cuda:0")
    prev = (torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32)
    torch.backends.cuda.matmul.allow_tf32 = torch.backends.cudnn.allow_tf32 = False
    cast = (lambda: torch.amp.autocast("cuda", dtype=torch.bfloat16)) if amp else (lambda: torch.amp.autocast("cuda", enabled=False))
    try:
        model = EagerCLIP(cfg, state).to(dev).train()
        B = image.shape[0]
        spans = [(r, min(B, r + chunk)) for r in range(0, B, chunk)]

        def towers(r0, r1):
            with cast():
                i, t = model.encode_image(image[r0:r1].to(dev).float()), model.encode_text(text[r0:r1].to(dev))
            return i.float(), t.float()

        with torch.no_grad():
            feats = [towers(r0, r1) for r0, r1 in spans]
        I = torch.cat([f[0] for f in feats]).requires_grad_(True)
        T = torch.cat([f[1] for f in feats]).requires_grad_(True)
        del feats
        with cast():
            if siglip:
                loss = siglip_loss(I, T, model.w("logit_scale").exp(), model.w("logit_bias"))
            else:
                loss = clip_loss(I, T, model.w("logit_scale").exp())
        loss.float().backward()  # -> I.grad, T.grad, logit_scale.grad (logit_bias.grad)
        for r0, r1 in spans:
            i, t = towers(r0, r1)
            ((i * I.grad[r0:r1]).sum() + (t * T.grad[r0:r1]).sum()).backward()
        torch.cuda.synchronize()
        grads = {k.replace("/", "."): (p.grad.detach().float().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model.p.items()}
        outs = {"image_features": I.detach().clone(), "text_features": T.detach().clone(), "loss": loss.detach().float().clone(),
                "d_image_features": I.grad.detach().clone(), "d_text_features": T.grad.detach().clone()}
        del model
        torch.cuda.empty_cache()
        return outs, grads
    finally:
        torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32 = prev


def loss_reference(I_all, T_all, logit_scale, logit_bias=None, siglip=False, rows=None):
    """fp32 statement of the two distributed loss forms on GATHERED features, for the at-size checks of the native loss's branches:
      ClipLoss (``siglip=False``): loss.py:106-107 -- the global loss over all N rows, li = s I_all T_all^T, lt = li^T, labels arange(N);
      SigLipLoss (``siglip=True``, ``rows`` = (lo, hi) of the calling rank): loss.py:406-489 -- the rank's B image rows against ALL texts,
        positives where the text index equals the row's global index, / B.
    -> dict(loss, dI [N,E] (SigLIP: the rank's rows only), dT [N,E], dscale (d loss / d logit_scale.exp()), dbias) in fp32 on the GPU.
    Plain torch autograd on fp32 tensors (TF32 off); pinned against oracle/clip_oracle.py at a small size by the test that uses it."""
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        I = I_all.detach().float().clone().requires_grad_(True)
        T = T_all.detach().float().clone().requires_grad_(True)
        s = logit_scale.detach().float().clone().requires_grad_(True)
        b = None if logit_bias is None else logit_bias.detach().float().clone().requires_grad_(True)
        if siglip:
            lo, hi = rows
            logits = (s * I[lo:hi]) @ T.t() + b
            labels = -torch.ones_like(logits)
            idx = torch.arange(hi - lo, device=I.device)
            labels[idx, lo + idx] = 1.0
            loss = -torch.nn.functional.logsigmoid(labels * logits).sum() / (hi - lo)
        else:
            loss = clip_loss(I, T, s)
        loss.backward()
        out = {"loss": loss.detach(), "dI": I.grad[rows[0]:rows[1]] if siglip else I.grad, "dT": T.grad, "dscale": s.grad,
               "dbias": None if b is None else b.grad}
        return out
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
