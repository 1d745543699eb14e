"""Same-GPU baseline: the reference's hot path as PLAIN PyTorch-ROCm eager ops under bf16 autocast -- BENCH INFRASTRUCTURE ONLY.

``bench.py``'s ``torch_eager_baseline`` leg times this on the MI355X next to the native step: it is "what a hipify of open_clip's
PyTorch stack gives" (last row).  /root/reference does not exist on the GPU box, so the reference's modules cannot be
imported there; this file calls the SAME library operators at the same places the reference does (each cited below), on the
reference's parameter names, so rocBLAS/hipBLASLt GEMMs, the SDPA kernel, ATen LayerNorm / GELU / softmax-CE and torch.optim.AdamW
do all the work.  It is never imported by the product (``open_clip_amd``) and never used as a checker.

    F.conv2d                               transformer.py:794        (conv1, kernel = stride = patch, no bias)
    F.layer_norm                           layers.py:20-26
    F.linear (in_proj / out_proj / mlp)    transformer.py:169,246,295-299
    F.scaled_dot_product_attention         transformer.py:223-228    (additive attn_mask for the text tower, :1716-1722)
    F.gelu (erf)                           transformer.py:295-299    (nn.GELU)
    F.normalize, F.cross_entropy           model.py:391,411; loss.py:136-139
    torch.amp.autocast(bf16), AdamW        precision.py:6-17; train.py:163-185
"""
import math

import torch
import torch.nn.functional as F


class EagerCLIP(torch.nn.Module):
    def __init__(self, cfg, state):
        super().__init__()
        self.cfg = cfg
        self.p = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v.clone().float()) for k, v in state.items()})
        L = cfg["text_cfg"]["context_length"]
        self.register_buffer("attn_mask", torch.full((L, L), float("-inf")).triu_(1), persistent=False)

    def w(self, k):
        return self.p[k.replace(".", "/")]

    @staticmethod
    def ln(x, w, b):
        """layers.py:20-26: the reference's LayerNorm hands its result back in the dtype of its INPUT.  Under autocast F.layer_norm returns fp32; with
        the cast the image tower's residual stream stays bf16 from conv1 on (transformer.py:794), as in the reference -- until round 6 this file
        left the cast out and its "amp_bf16" image stream was fp32 from ln_pre on: a yardstick more accurate (and slower) than the reference's policy"""
        return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5).to(x.dtype)

    def block(self, x, pre, heads, mask):
        B, L, C = x.shape
        h = self.ln(x, self.w(pre + "ln_1.weight"), self.w(pre + "ln_1.bias"))
        q, k, v = F.linear(h, self.w(pre + "attn.in_proj_weight"), self.w(pre + "attn.in_proj_bias")).chunk(3, dim=-1)
        q, k, v = (t.reshape(B, L, heads, C // heads).transpose(1, 2) for t in (q, k, v))
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=None if mask is None else mask.to(q.dtype), scale=(C // heads) ** -0.5)
        a = a.transpose(1, 2).reshape(B, L, C)
        x = x + F.linear(a, self.w(pre + "attn.out_proj.weight"), self.w(pre + "attn.out_proj.bias"))
        h = self.ln(x, self.w(pre + "ln_2.weight"), self.w(pre + "ln_2.bias"))
        h = F.gelu(F.linear(h, self.w(pre + "mlp.c_fc.weight"), self.w(pre + "mlp.c_fc.bias")))
        return x + F.linear(h, self.w(pre + "mlp.c_proj.weight"), self.w(pre + "mlp.c_proj.bias"))

    def encode_image(self, image):
        v = self.cfg["vision_cfg"]
        width, ps = v["width"], v["patch_size"]
        x = F.conv2d(image, self.w("visual.conv1.weight"), stride=ps)
        x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)
        cls = self.w("visual.class_embedding").to(x.dtype).reshape(1, 1, width).expand(x.shape[0], -1, -1)
        x = torch.cat([cls, x], dim=1) + self.w("visual.positional_embedding").to(x.dtype)
        x = self.ln(x, self.w("visual.ln_pre.weight"), self.w("visual.ln_pre.bias"))
        heads = width // v.get("head_width", 64)
        for i in range(v["layers"]):
            x = self.block(x, f"visual.transformer.resblocks.{i}.", heads, None)
        x = self.ln(x, self.w("visual.ln_post.weight"), self.w("visual.ln_post.bias"))
        return F.normalize(x[:, 0] @ self.w("visual.proj"), dim=-1)

    def encode_text(self, text):
        t = self.cfg["text_cfg"]
        x = F.embedding(text, self.w("token_embedding.weight")) + self.w("positional_embedding")
        for i in range(t["layers"]):
            x = self.block(x, f"transformer.resblocks.{i}.", t["heads"], self.attn_mask)
        x = self.ln(x, self.w("ln_final.weight"), self.w("ln_final.bias"))
        x = x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)] @ self.w("text_projection")
        return F.normalize(x, dim=-1)

    def forward(self, image, text):
        return self.encode_image(image), self.encode_text(text), self.w("logit_scale").exp()


def clip_loss(i, t, s):
    li = s * i @ t.T
    lt = s * t @ i.T
    labels = torch.arange(li.shape[0], device=li.device)
    return (F.cross_entropy(li, labels) + F.cross_entropy(lt, labels)) / 2


def siglip_loss(i, t, s, b):
    """loss.py:356-367 (SigLipLoss._loss, world_size 1): -sum(logsigmoid(labels * (s I T^T + b))) / B, labels = 2 eye - 1"""
    logits = s * i @ t.T + b
    labels = 2 * torch.eye(logits.shape[0], device=logits.device, dtype=logits.dtype) - 1
    return -F.logsigmoid(labels * logits).sum() / logits.shape[0]


def amp_step_grads(cfg, state, image, text):
    """features, loss and every parameter gradient of ONE step of the same eager operators under ``torch.amp.autocast(bf16)`` -- the reference's own
    ``--precision amp_bf16`` policy (precision.py:6-17) on this GPU.  tests/test_bench_size_gpu.py measures it against the fp32 reference at the
    bench's batch to show what that POLICY costs the 1-D gradients there (column sums over thousands of rows that cancel), next to the native
    path's own error: a yardstick for the tolerance, never a checker."""
    dev = image.device
    model = EagerCLIP(cfg, state).to(dev).train()
    with torch.amp.autocast("cuda", dtype=torch.bfloat16):
        i, t, s = model(image, text)
        loss = clip_loss(i, t, s)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k.replace("/", "."): p.grad.detach().float().clone() for k, p in model.p.items() if p.grad is not None}
    outs = {"image_features": i.detach().float(), "text_features": t.detach().float(), "loss": loss.detach().float()}
    del model
    torch.cuda.empty_cache()
    return outs, grads


def time_step(cfg, state, batch, steps=5, warmup=2, lr=5e-4):
    """seconds per training step (forward under bf16 autocast, backward, AdamW, clamp) on ``batch`` (already on the device)"""
    dev = batch["image"].device
    model = EagerCLIP(cfg, state).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.amp.autocast("cuda", dtype=torch.bfloat16):
            i, t, s = model(batch["image"], batch["text"])
            loss = clip_loss(i, t, s)
        loss.backward()
        opt.step()
        with torch.no_grad():
            model.w("logit_scale").clamp_(0, math.log(100))
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # every step timed on its own, the MEDIAN reported: one step that pays for an allocator growth or a library's first-use tuning must not set the
    # baseline (round 6: one run of four averaged 1.3 s per step on a box whose other runs gave 0.5 s)
    evs = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loss = step()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    times = sorted(a.elapsed_time(b) / 1e3 for a, b in evs)
    time_step.last_all = [round(t, 4) for t in times]
    sec = times[len(times) // 2]
    peak = torch.cuda.max_memory_allocated()
    del model, opt
    torch.cuda.empty_cache()
    return sec, float(loss.detach()), peak
