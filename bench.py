"""bench.py -- image-text pairs/sec of the native CLIP training step (ViT-B-32, local batch 4096 per GPU).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 launched under
``python -m torch.distributed.run --nproc-per-node N`` (one rank per GPU, RCCL).  One "step" = zero_grad ->
forward (both towers) -> ClipLoss (packed feature all-gather + global logits when N>1) -> backward (DDP bucketed
grad all-reduce overlapped) -> AdamW step -> logit_scale clamp, on synthetic inputs already resident in HBM.
Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel (the NT MFMA GEMM) from HIP events recorded
around every one of its launches inside the timed region; ``cpu_baseline`` times the CPU oracle (port of the
reference path) on the host cores for a bounded sample (N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md
FWD_GFLOP_PER_PAIR = {"ViT-B-32": 14.78, "ViT-L-14": 175.33, "ViT-H-14": 381.68}  # docs/model_profile.csv (reference)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="ViT-B-32")
    ap.add_argument("--local-batch", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--grad-checkpointing", action="store_true")
    ap.add_argument("--siglip", action="store_true", help="SigLIPTask-equivalent step (sigmoid pairwise loss, logit_bias; BASELINE config 5)")
    ap.add_argument("--naive-global-loss", action="store_true", help="N>1: every rank evaluates the full N x N logits (the reference's "
                    "redundant form) instead of its own rows (same loss and gradients; tests/test_dist_loss_gloo.py, test_ddp_gpu.py)")
    ap.add_argument("--gemm-variant", type=int, default=0, help="developer: value for ocn_set_gemm_variant (kernel choice / ablation knobs)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE", help="developer: ocn_set_tuning(KEY, VALUE) before the run")
    ap.add_argument("--dist-backend", default="nccl", help="developer: 'gloo' + OCN_BENCH_ONE_DEVICE=1 runs N ranks on one GPU")
    return ap.parse_args()


class GemmTimer:
    """HIP events around every ocn_gemm_nt / ocn_gemm_tn_accum launch (same stream as the launch)."""

    def __init__(self):
        self.rec = []
        self.on = False

    def install(self):
        from open_clip_amd import ops
        nt, tn = ops.gemm_nt, ops.gemm_tn_accum
        timer = self

        def gemm_nt(epi, a, b, out, **kw):
            if not timer.on:
                return nt(epi, a, b, out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = nt(epi, a, b, out, **kw)
            e1.record()
            timer.rec.append(("nt", 2.0 * a.shape[0] * b.shape[0] * a.shape[1], e0, e1))
            return r

        def gemm_tn(a, b, dw, dbias=None, alpha=1.0):
            if not timer.on:
                return tn(a, b, dw, dbias, alpha)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = tn(a, b, dw, dbias, alpha)
            e1.record()
            timer.rec.append(("tn", 2.0 * a.shape[0] * a.shape[1] * b.shape[1], e0, e1))
            return r

        ops.gemm_nt, ops.gemm_tn_accum = gemm_nt, gemm_tn

    def summary(self):
        out = {}
        for kind in ("nt", "tn"):
            fl = sum(r[1] for r in self.rec if r[0] == kind)
            ms = sum(r[2].elapsed_time(r[3]) for r in self.rec if r[0] == kind)
            n = sum(1 for r in self.rec if r[0] == kind)
            out[kind] = {"launches": n, "tflop": fl / 1e12, "ms": ms, "tflops": (fl / 1e12) / (ms / 1e3) if ms > 0 else 0.0}
        return out


def cpu_baseline(model_name, seconds=20.0):
    """CPU oracle (port of the reference hot path) fwd+bwd+AdamW at the reference's CPU config (bs 32, fp32)."""
    from oracle import clip_oracle as O
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.synth import init_state_dict, synthetic_batch
    cfg = get_model_config(model_name)
    state = init_state_dict(cfg, seed=0)
    bs = 32
    batch = synthetic_batch(cfg, bs, seed=1234)
    params = {k: v.clone() for k, v in state.items()}
    plist = [torch.nn.Parameter(v) for v in params.values()]
    opt = torch.optim.AdamW(plist, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
    times = []
    t_end = time.time() + seconds
    while time.time() < t_end or len(times) < 2:
        t0 = time.time()
        outs, grads = O.train_forward_backward(batch["image"], batch["text"], {k: p.detach() for k, p in zip(params, plist)}, cfg)
        for p, k in zip(plist, params):
            p.grad = grads[k]
        opt.step()
        times.append(time.time() - t0)
        if len(times) >= 12:
            break
    warm = sorted(times[1:])
    med = warm[len(warm) // 2]
    return {"value": bs / med, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} steps of CPU-oracle fwd+bwd+AdamW, {model_name} fp32, batch {bs}; median of warm steps ({med:.2f} s/step)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if os.environ.get("OCN_BENCH_ONE_DEVICE") == "1":  # developer mode: every rank on GPU 0 (needs --dist-backend gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from open_clip_amd.configs import get_model_config
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.model import NativeCLIP
    from open_clip_amd.optim import NativeAdamW, param_groups_like_reference, weight_caches_of
    from open_clip_amd.synth import init_state_dict, synthetic_batch

    if args.gemm_variant or args.tuning:
        from open_clip_amd import _lib
        if args.gemm_variant:
            _lib.call("ocn_set_gemm_variant", args.gemm_variant)
        for kv in args.tuning:
            k, v = kv.split("=")
            _lib.call("ocn_set_tuning", int(k), int(v))
    cfg = get_model_config(args.model)
    torch.manual_seed(0)
    extra = dict(init_logit_scale=math.log(10), init_logit_bias=-10.0) if args.siglip else {}  # main.py:259-261
    model = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=True, **extra)
    model.load_state_dict(init_state_dict(cfg, seed=0, siglip=args.siglip))
    model = model.to(dev).train()
    if args.grad_checkpointing:
        model.set_grad_checkpointing(True)
    B = args.local_batch
    batch = synthetic_batch(cfg, B, seed=1234, rank=rank, device=dev)
    if args.siglip:
        from open_clip_amd.loss import NativeSigLipLoss
        loss_fn = NativeSigLipLoss(rank=rank, world_size=world)
    else:
        loss_fn = NativeClipLoss(local_loss=False, gather_with_grad=False, rank=rank, world_size=world,
                                 row_sharded=(world > 1 and not args.naive_global_loss))
    opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_caches=weight_caches_of(model))
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], bucket_cap_mb=128, gradient_as_bucket_view=True)

    timer = GemmTimer()
    if not args.no_roofline:
        timer.install()

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(image=batch["image"], text=batch["text"])
        loss = loss_fn(**out)
        loss.backward()
        opt.step()
        with torch.no_grad():
            model.logit_scale.clamp_(0, math.log(100))  # image_text_task.py:91-101
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    # GEMM launches carry HIP events on every second timed step (the events cost ~0.8 % of a step when every launch has them)
    timed_steps = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.on = (not args.no_roofline) and (i % 2 == 0)
        timed_steps += int(timer.on)
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    timer.on = False
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * world / (elapsed / args.steps)
        flops_pair = 3 * FWD_GFLOP_PER_PAIR.get(args.model, 0.0) * (4 / 3 if args.grad_checkpointing else 1.0)
        line = {
            "metric": ("image-text pairs/sec (whole node), ViT-B-32 gbs=32768 at 1/2/4/8 GPUs" if args.model == "ViT-B-32" and not args.siglip
                       else f"image-text pairs/sec (whole node), {args.model}{' SigLIP' if args.siglip else ''}"), "value": round(value, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model} {'SigLIPTask' if args.siglip else 'CLIPTask'}-equivalent train step (fwd+{'SigLipLoss' if args.siglip else 'ClipLoss'}+bwd+AdamW+clamp), amp_bf16 policy, "
                                   f"local_bs={B}, global_bs={B * world}, "
                                   + (("gather_features all-gather + global logits" + ("" if args.naive_global_loss else " (row-sharded across ranks)"))
                                      if world > 1 else "world_size 1 (no all-gather)"),
                       "model": args.model, "global_batch": B * world, "local_batch": B, "parallelism": f"dp{world}",
                       "random_init_weights": True, "final_loss": round(final_loss, 4)},
            "step_model_tflops_per_gpu": round(value / world * flops_pair / 1e3, 1),
            "peak_hbm_gb_rank0": round(torch.cuda.max_memory_allocated() / 1e9, 1),
        }
        if not args.no_roofline:
            s = timer.summary()
            nt, tn = s["nt"], s["tn"]
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written by tools/pmc_stats.py from the PMC passes
            if os.path.exists(tpath) and args.model == "ViT-B-32" and B == 4096:
                rec = json.load(open(tpath))
                traffic, traffic_src = round(rec["bytes_per_launch"]), rec["source"]
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_nt5_kernel (all epilogues)", "achieved": round(nt["tflops"], 1),
                                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(nt["tflops"] / PEAK_BF16_TFLOPS, 4),
                                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src, "launches": nt["launches"], "avg_launch_ms": round(nt["ms"] / max(nt["launches"], 1), 4),
                                "algorithmic_tflop_per_launch_avg": round(nt["tflop"] / max(nt["launches"], 1), 4),
                                "gemm_tn_kernel": {"achieved": round(tn["tflops"], 1), "frac": round(tn["tflops"] / PEAK_BF16_TFLOPS, 4),
                                                   "launches": tn["launches"], "avg_launch_ms": round(tn["ms"] / max(tn["launches"], 1), 4),
                                                   "note": "wgrad launches run on a side stream UNDER the LayerNorm / attention backward kernels of "
                                                           "their block (model.py::_Paired), so these events time a co-scheduled kernel; alone it "
                                                           "runs at ~1.2 PFLOP/s (profiles/r01_gemm_shapes.txt, OCN_WGRAD_STREAM=0)"},
                                "event_timed_steps": timed_steps,
                                "gemm_share_of_step": round((nt["ms"] + tn["ms"]) / (elapsed * 1e3 * timed_steps / args.steps), 3)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
