"""Generates ``tests/golden/*.npz`` by running the REFERENCE (``/root/reference/src/open_clip``)
on seeded inputs.  TEST INFRASTRUCTURE ONLY; run in the build container (the reference is not
on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Fixtures (all fp32 CPU, ``torch.use_deterministic_algorithms(True)``):
  tiny_clip.npz      'tiny-test' config; weights (fp16-representable, stored as fp16), inputs and the
                     reference's features / logits / loss / parameter grads (``CLIP`` + ``ClipLoss``)
  tiny_siglip.npz    same model with logit_bias; ``SigLipLoss`` (world_size 1)
  tiny_quickgelu.npz same shape with ``quick_gelu=True`` (QuickGELU in both towers, layers.py:29-32)
  tiny_hd88.npz      'hd88-test' config: image head_width 88 and mlp_ratio 4.3637 (ViT-g-14's shape class)
  vitb32_b8.npz      ViT-B-32, B=8; weights regenerated from ``init_state_dict(seed=0, perturb=True)``
                     (checksums stored), reference outputs + grad samples
  dist_loss_w2.npz   ``ClipLoss`` (3 gather modes) and ``SigLipLoss`` ('bidir') under gloo, world_size 2/3:
                     per-rank loss and feature grads on random unit features
Large grads are stored as (L2 norm, sum, 4096-element strided sample); small ones in full.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_shim import import_reference  # noqa: E402
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.synth import init_state_dict, synthetic_batch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FULL_LIMIT = 20000
SAMPLE = 4096


def sample_idx(n):
    stride = max(1, n // SAMPLE)
    return np.arange(0, n, stride)[:SAMPLE]


def pack_grads(out, grads):
    for k, g in grads.items():
        a = g.detach().float().reshape(-1).numpy()
        out["gnorm/" + k] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
        out["gsum/" + k] = np.float64(a.astype(np.float64).sum())
        if a.size <= FULL_LIMIT:
            out["grad/" + k] = a.reshape(tuple(g.shape))
        else:
            out["gsample/" + k] = a[sample_idx(a.size)]


def ref_model(cfg, state, siglip=False):
    import_reference()
    from open_clip.model import CLIP

    kw = dict(init_logit_scale=float(np.log(10)), init_logit_bias=-10.0) if siglip else {}
    v = {k: cfg["vision_cfg"][k] for k in ("image_size", "layers", "width", "patch_size", "head_width", "mlp_ratio") if k in cfg["vision_cfg"]}
    t = {k: cfg["text_cfg"][k] for k in ("context_length", "vocab_size", "width", "heads", "layers")}
    if cfg.get("quick_gelu"):
        kw["quick_gelu"] = True
    m = CLIP(embed_dim=cfg["embed_dim"], vision_cfg=v, text_cfg=t, output_dict=True, **kw)
    missing, unexpected = m.load_state_dict(state, strict=True)
    return m.float().train()


def run_reference(cfg, state, batch, siglip=False):
    """Reference forward+backward through its own task layer (clip_task.py:41-46 / siglip_task.py)."""
    import_reference()
    from open_clip.task import CLIPTask, SigLIPTask

    model = ref_model(cfg, state, siglip)
    task = (SigLIPTask if siglip else CLIPTask)(model, rank=0, world_size=1)
    task.train()
    losses, report = task({"image": batch["image"].float(), "text": batch["text"]})
    losses["loss"].backward()
    with torch.no_grad():
        mo = model(image=batch["image"].float(), text=batch["text"])
        logits = mo["logit_scale"] * mo["image_features"] @ mo["text_features"].t()
        if siglip:
            logits = logits + mo["logit_bias"]
    grads = {k: p.grad for k, p in model.named_parameters()}
    outs = {
        "image_features": mo["image_features"], "text_features": mo["text_features"],
        "logits_per_image": logits, "loss": losses["loss"].detach(), "logit_scale_exp": mo["logit_scale"],
    }
    return outs, grads


def fp16_representable(sd):
    return {k: (v.half().float() if v.dtype.is_floating_point else v) for k, v in sd.items()}


def make_tiny(siglip=False):
    cfg = get_model_config("tiny-test")
    state = fp16_representable(init_state_dict(cfg, seed=7, perturb=True, siglip=siglip))
    batch = synthetic_batch(cfg, 6, seed=99)
    batch["image"] = batch["image"].half().float()
    outs, grads = run_reference(cfg, state, batch, siglip)
    out = {"image": batch["image"].numpy().astype(np.float16), "text": batch["text"].numpy()}
    for k, v in state.items():
        out["w/" + k] = v.numpy().astype(np.float16) if v.ndim > 0 else v.numpy().astype(np.float32)
    for k, v in outs.items():
        out["out/" + k] = v.detach().numpy()
    pack_grads(out, grads)
    name = "tiny_siglip.npz" if siglip else "tiny_clip.npz"
    np.savez_compressed(os.path.join(GOLD, name), **out)
    print(name, "loss", float(outs["loss"]))


def make_tiny_quickgelu():
    """the `quick_gelu` configs (ViT-B-32-quickgelu.json): QuickGELU in both towers, reference ``CLIP(quick_gelu=True)``"""
    cfg = dict(get_model_config("tiny-test"), quick_gelu=True)
    state = fp16_representable(init_state_dict(cfg, seed=8, perturb=True))
    batch = synthetic_batch(cfg, 6, seed=98)
    batch["image"] = batch["image"].half().float()
    outs, grads = run_reference(cfg, state, batch)
    out = {"image": batch["image"].numpy().astype(np.float16), "text": batch["text"].numpy()}
    for k, v in state.items():
        out["w/" + k] = v.numpy().astype(np.float16) if v.ndim > 0 else v.numpy().astype(np.float32)
    for k, v in outs.items():
        out["out/" + k] = v.detach().numpy()
    pack_grads(out, grads)
    np.savez_compressed(os.path.join(GOLD, "tiny_quickgelu.npz"), **out)
    print("tiny_quickgelu.npz loss", float(outs["loss"]))


def make_tiny_hd88():
    """head_width 88 / non-integer mlp_ratio (ViT-g-14.json): pins the oracle's (and the kernels') handling of both against the reference"""
    cfg = get_model_config("hd88-test")
    state = fp16_representable(init_state_dict(cfg, seed=9, perturb=True))
    batch = synthetic_batch(cfg, 6, seed=97)
    batch["image"] = batch["image"].half().float()
    outs, grads = run_reference(cfg, state, batch)
    out = {"image": batch["image"].numpy().astype(np.float16), "text": batch["text"].numpy()}
    for k, v in state.items():
        out["w/" + k] = v.numpy().astype(np.float16) if v.ndim > 0 else v.numpy().astype(np.float32)
    for k, v in outs.items():
        out["out/" + k] = v.detach().numpy()
    pack_grads(out, grads)
    np.savez_compressed(os.path.join(GOLD, "tiny_hd88.npz"), **out)
    print("tiny_hd88.npz loss", float(outs["loss"]))


def make_vitb32():
    cfg = get_model_config("ViT-B-32")
    state = init_state_dict(cfg, seed=0, perturb=True)
    batch = synthetic_batch(cfg, 8, seed=1234)
    outs, grads = run_reference(cfg, state, batch)
    out = {"text": batch["text"].numpy(), "image_checksum": np.float64(batch["image"].double().sum())}
    for k in ("visual.conv1.weight", "token_embedding.weight", "transformer.resblocks.11.mlp.c_fc.weight", "visual.proj"):
        out["wsum/" + k] = np.float64(state[k].double().sum())
    for k, v in outs.items():
        out["out/" + k] = v.detach().numpy()
    pack_grads(out, grads)
    np.savez_compressed(os.path.join(GOLD, "vitb32_b8.npz"), **out)
    print("vitb32_b8.npz loss", float(outs["loss"]))


# ------------------------------------------------------------------------------------------
# distributed loss semantics (gloo, CPU): reference ClipLoss / SigLipLoss per rank
# ------------------------------------------------------------------------------------------
def _dist_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import_reference()
    from open_clip.loss import ClipLoss, SigLipLoss

    B, E = 5, 32
    res = {}
    g = torch.Generator().manual_seed(4242)
    feats = torch.nn.functional.normalize(torch.randn(world, 2, B, E, generator=g), dim=-1)
    scale = torch.tensor(9.5)
    bias = torch.tensor(-3.0)
    for name, kw in (("global", dict(local_loss=False, gather_with_grad=False)),
                     ("local_gwg", dict(local_loss=True, gather_with_grad=True)),
                     ("local_nograd", dict(local_loss=True, gather_with_grad=False)),
                     ("global_gwg", dict(local_loss=False, gather_with_grad=True))):
        img = feats[rank, 0].clone().requires_grad_(True)
        txt = feats[rank, 1].clone().requires_grad_(True)
        s = scale.clone().requires_grad_(True)
        loss = ClipLoss(rank=rank, world_size=world, **kw)(img, txt, s)
        loss.backward()
        res[f"clip/{name}/loss"] = loss.detach().numpy()
        res[f"clip/{name}/dimg"] = img.grad.numpy()
        res[f"clip/{name}/dtxt"] = txt.grad.numpy()
        res[f"clip/{name}/dscale"] = s.grad.numpy()
    img = feats[rank, 0].clone().requires_grad_(True)
    txt = feats[rank, 1].clone().requires_grad_(True)
    s = scale.clone().requires_grad_(True)
    b = bias.clone().requires_grad_(True)
    loss = SigLipLoss(rank=rank, world_size=world, dist_impl="bidir")(img, txt, s, b)
    loss.backward()
    res["siglip/bidir/loss"] = loss.detach().numpy()
    res["siglip/bidir/dimg"] = img.grad.numpy()
    res["siglip/bidir/dtxt"] = txt.grad.numpy()
    res["siglip/bidir/dscale"] = s.grad.numpy()
    res["siglip/bidir/dbias"] = b.grad.numpy()
    # the memory-efficient chunked evaluation (loss.py:369-404) and ClipLoss with a logit_bias (loss.py:111-113: real, zero gradient)
    img = feats[rank, 0].clone().requires_grad_(True)
    txt = feats[rank, 1].clone().requires_grad_(True)
    s = scale.clone().requires_grad_(True)
    b = bias.clone().requires_grad_(True)
    loss = SigLipLoss(rank=rank, world_size=world, dist_impl="bidir", chunk_size=2)(img, txt, s, b)
    loss.backward()
    res["siglip/chunked/loss"] = loss.detach().numpy()
    res["siglip/chunked/dimg"] = img.grad.numpy()
    res["siglip/chunked/dtxt"] = txt.grad.numpy()
    res["siglip/chunked/dscale"] = s.grad.numpy()
    res["siglip/chunked/dbias"] = b.grad.numpy()
    img = feats[rank, 0].clone().requires_grad_(True)
    txt = feats[rank, 1].clone().requires_grad_(True)
    s = scale.clone().requires_grad_(True)
    b = bias.clone().requires_grad_(True)
    loss = ClipLoss(rank=rank, world_size=world, local_loss=True, gather_with_grad=True)(img, txt, s, b)
    loss.backward()
    res["clip/local_gwg_bias/loss"] = loss.detach().numpy()
    res["clip/local_gwg_bias/dimg"] = img.grad.numpy()
    res["clip/local_gwg_bias/dbias"] = b.grad.numpy()
    q.put((rank, res, feats.numpy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def make_dist(world, port):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join()
    out = {"scale": np.float32(9.5), "bias": np.float32(-3.0)}
    for rank, res, feats in got:
        if feats is not None:
            out["feats"] = feats
        for k, v in res.items():
            out[f"r{rank}/{k}"] = v
    np.savez_compressed(os.path.join(GOLD, f"dist_loss_w{world}.npz"), **out)
    print(f"dist_loss_w{world}.npz", {k: float(v) for k, v in out.items() if k.endswith("/loss")})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.use_deterministic_algorithms(True)
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["tiny", "siglip", "quickgelu", "hd88", "dist", "vitb32"]
    if "tiny" in which:
        make_tiny(False)
    if "siglip" in which:
        make_tiny(True)
    if "quickgelu" in which:
        make_tiny_quickgelu()
    if "hd88" in which:
        make_tiny_hd88()
    if "dist" in which:
        make_dist(2, 29611)
        make_dist(3, 29612)
        make_dist(8, 29613)  # the 8-rank world of BASELINE config 3
    if "vitb32" in which:
        make_vitb32()
