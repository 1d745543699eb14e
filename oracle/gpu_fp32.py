"""Whole-step fp32 reference ON THE GPU for batches the CPU oracle cannot reach -- TEST INFRASTRUCTURE ONLY.

``oracle/clip_oracle.py`` (pinned against the reference's own outputs, tests/golden/) takes minutes of host time at batch 512 and would take
the better part of an hour at the bench's batch of 4096.  This file evaluates the SAME training step -- both towers, ClipLoss, backward -- in
pure fp32 (no autocast, TF32 off) with the library operators of ``oracle/torch_eager.py::EagerCLIP`` on the MI355X, in batch CHUNKS so that
fp32 activations of 4096 pairs never have to exist at once:

    1. features of every chunk under no_grad                                  -> I [B,E], T [B,E]
    2. the loss on the full batch from those features, with gradient          -> loss, dL/dI, dL/dT, dL/dlogit_scale   (loss.py:91-141)
    3. every chunk again with gradient, backward of <I_c, dL/dI_c> + <T_c, dL/dT_c>   -> parameter gradients accumulate

which is exact: the towers are per-sample maps, so dL/dtheta = sum_c J_c^T dL/dfeatures_c (the same identity the reference's own
--accum-freq path rests on, train.py:236-311).  It is pinned, not trusted: tests/test_bench_size_gpu.py first checks it against the CPU oracle
at batch 512 (features, loss and all 302 gradients to <= 1e-4), then uses it as the reference of the native step at batch 4096.
Never imported by the product (``open_clip_amd``)."""
import torch

from oracle.torch_eager import EagerCLIP, clip_loss


def step_reference(cfg, state, image, text, chunk=512):
    """-> (outs, grads): outs = image_features / text_features [B,E] fp32, loss; grads keyed by the reference's state-dict names.
    ``image`` / ``text`` may live on the host: chunks are moved to the GPU one at a time."""
    dev = torch.device("cuda:0")
    prev = (torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32)
    torch.backends.cuda.matmul.allow_tf32 = torch.backends.cudnn.allow_tf32 = False
    try:
        model = EagerCLIP(cfg, state).to(dev).train()
        B = image.shape[0]
        spans = [(r, min(B, r + chunk)) for r in range(0, B, chunk)]

        def towers(r0, r1):
            return model.encode_image(image[r0:r1].to(dev).float()), model.encode_text(text[r0:r1].to(dev))

        with torch.no_grad():
            feats = [towers(r0, r1) for r0, r1 in spans]
        I = torch.cat([f[0] for f in feats]).requires_grad_(True)
        T = torch.cat([f[1] for f in feats]).requires_grad_(True)
        del feats
        loss = clip_loss(I, T, model.w("logit_scale").exp())
        loss.backward()  # -> I.grad, T.grad, logit_scale.grad
        for r0, r1 in spans:
            i, t = towers(r0, r1)
            ((i * I.grad[r0:r1]).sum() + (t * T.grad[r0:r1]).sum()).backward()
        torch.cuda.synchronize()
        grads = {k.replace("/", "."): (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model.p.items()}
        outs = {"image_features": I.detach().clone(), "text_features": T.detach().clone(), "loss": loss.detach().clone(),
                "d_image_features": I.grad.detach().clone(), "d_text_features": T.grad.detach().clone()}
        del model
        torch.cuda.empty_cache()
        return outs, grads
    finally:
        torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32 = prev
