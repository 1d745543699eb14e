"""bench.py -- image-text pairs/sec of the native CLIP training step (ViT-B-32, local batch 4096 per GPU).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  For N>1 either a launcher started this file once per GPU
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or -- called
as plain ``python bench.py --gpus N`` -- it starts the N ranks itself the same way (``launcher_command``) and hands their exit code on; a
WORLD_SIZE that is not N, or fewer visible GPUs than N, is an error, never a silently smaller run.  One rank per GPU, RCCL.  One "step" = zero_grad ->
forward (both towers) -> ClipLoss (packed feature all-gather + global logits when N>1) -> backward (DDP bucketed
grad all-reduce overlapped) -> AdamW step -> logit_scale clamp, on synthetic inputs already resident in HBM.
Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel (the NT MFMA GEMM) from HIP events recorded
around every one of its launches inside the timed region; ``cpu_baseline`` times the reference's own train_one_epoch
(from oracle/_ref/reference_src.zip on the GPU box; the CPU oracle -- the port -- when that is missing) on the host cores for a bounded sample (N=1 only).
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
HBM_SPEC_GBPS, HBM_ACHIEVABLE_GBPS = 8000.0, 6300.0  # MI355X_MICROARCH.md: 8 TB/s spec, 6.29 TB/s measured with a float4 copy


def code_identity():
    """(git sha of the tree or None, sha256[:16] over the kernel sources): the second is what profiles/pmc_traffic.json is compared with --
    a PMC figure taken on other kernel code is marked stale on the line"""
    import hashlib
    import subprocess
    sha = None
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        sha = None
    if sha is None and os.path.exists(os.path.join(ROOT, ".head_sha")):
        sha = open(os.path.join(ROOT, ".head_sha")).read().strip() or None
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "open_clip_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return sha, h.hexdigest()[:16]
FWD_GFLOP_PER_PAIR = {"ViT-B-32": 14.78, "ViT-L-14": 175.33, "ViT-H-14": 381.68}  # docs/model_profile.csv (reference); others: configs.forward_gflops_per_pair


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="ViT-B-32")
    ap.add_argument("--local-batch", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the plain PyTorch-ROCm eager step timed beside the native one (N=1)")
    ap.add_argument("--eager-batch", type=int, default=4096, help="batch of the eager baseline (the bench's own batch: like for like; halved on out-of-memory)")
    ap.add_argument("--deterministic", action="store_true", help="the reproducible step (NativeCLIP(deterministic=True) + NativeClipLoss(deterministic=True)): no "
                    "fp32 atomic in any gradient or loss sum -- bit-identical from run to run")
    ap.add_argument("--accum-freq", type=int, default=1, help="reference --accum-freq semantics (train.py:236-311): F micro-batches of "
                    "--local-batch per optimizer step (features cached under no_grad, every micro-batch re-run with gradient against the "
                    "concatenation) -> global batch = local_batch * F * N; --accum-freq 8 is the metric's gbs=32768 on ONE GPU")
    ap.add_argument("--h2d", action="store_true", help="feed every step from pinned HOST memory: uint8 [B,H,W,3] pixels + tokens, double-buffered "
                    "async copies on a copy stream, normalisation inside the patch kernel (open_clip_amd/input_pipeline.py); the copies are "
                    "inside the timed region")
    ap.add_argument("--force-ddp", action="store_true", help="developer: with one process, still wrap the model in DistributedDataParallel over a "
                    "one-rank RCCL group (what the gradient buckets, their copies and the reducer hooks cost without any communication)")
    ap.add_argument("--native-allreduce", action="store_true", help="gradient all-reduce WITHOUT DistributedDataParallel: per-block in-place all-reduce of the "
                    "backward's own gradient arenas from post-accumulate hooks (open_clip_amd/grad_sync.py; RCCL through the C ABI, or the process "
                    "group with --dist-backend gloo); with one process: a one-rank communicator (what the hooks and launches cost)")
    ap.add_argument("--native-comm", action="store_true", help="the loss's feature all-gather / reduce-scatter / scalar all-reduce through the C ABI's RCCL communicator "
                    "(ocn_comm_*) instead of torch.distributed's process group; with one process: a one-rank communicator -- the loss runs its distributed "
                    "(row-sharded) form with identity collectives (what the marshalling and the extra launches cost)")
    ap.add_argument("--native", action="store_true", help="= --native-comm --native-allreduce: every collective of the step through the C ABI's RCCL communicator "
                    "(ocn_comm_*); DistributedDataParallel + torch.distributed stay the fallback when the communicator cannot be created")
    ap.add_argument("--torch-comm", action="store_true", help="N > 1: the loss collectives through torch.distributed's process group, too (default for N > 1 with the "
                    "nccl backend: the loss's feature all-gather / reduce-scatter / scalar all-reduce through ocn_comm_*, the gradient all-reduce through DDP)")
    ap.add_argument("--tile-rescue", choices=("auto", "on", "off"), default="auto", help="the persistent GEMMs' multi-GPU form (ocn_set_tile_rescue: workgroups that "
                    "finish hand out the shares of workgroups whose CU is held by a collective's kernel); auto = on when N > 1")
    ap.add_argument("--bucket-cap-mb", type=int, default=128)
    ap.add_argument("--data-ranks", type=int, default=1, help="developer: with one process, use the concatenation of the batches R ranks would "
                    "get (what a world_size-R run sees as its global batch; tests/test_bench_gpu.py)")
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--lr-warmup-steps", type=int, default=10000, help="linear warm-up as the reference schedules it (params.py:288, scheduler.py:6-15)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-clock-sample", action="store_true", help="do not read the GPU's clock / power with rocm-smi during the timed steps")
    ap.add_argument("--grad-checkpointing", action="store_true")
    ap.add_argument("--keep-blocks", default="auto",
                    help="with --grad-checkpointing: blocks per tower whose activations are kept instead of recomputed -- 'auto' (as many as the "
                         "free HBM holds, NativeCLIP.plan_grad_checkpointing), 'V,T' (image, text), '0' (recompute every block as the reference)")
    ap.add_argument("--serial-towers", action="store_true", help="image and text tower on ONE stream (default: the image tower on a stream of its own next "
                    "to the text tower, NativeCLIP._tower_side; the steps whose GEMM launches carry HIP events always run serially)")
    ap.add_argument("--no-wgrad-pair", action="store_true", help="with --serial-towers: no wgrad side stream either (every kernel alone on the chip: the "
                    "configuration of the event-timed steps, used for the rocprofv3 / PMC passes)")
    ap.add_argument("--no-dense-text-line", action="store_true", help="skip the extra --dense-text timing that the default line carries")
    ap.add_argument("--no-extra-lines", action="store_true", help="skip the two further samples the default line carries: `reference_work` (dense text tower AND "
                    "full last blocks: exactly the rows the reference executes) and `accum8_gbs32768` (the metric's own global batch on one GPU, --accum-freq 8)")
    ap.add_argument("--no-config-lines", action="store_true", help="skip the two samples of BASELINE configs 4 / 5 (ViT-L-14 with block recompute; ViT-H-14 + SigLIP) "
                    "that the default ViT-B-32 line carries as `config4_vitl14` / `config5_vith14_siglip` (each a subprocess of this file)")
    ap.add_argument("--dense-text", action="store_true", help="run all context_length positions of every caption through the text tower like the reference "
                    "does (default: packed -- only the tokens up to the pooled EOT exist; same features, loss and gradients, see model.py::_TextPack)")
    ap.add_argument("--image-stream", default="bf16", choices=["fp32", "bf16", "bf16-fp32grad"],
                    help="dtype of the IMAGE tower's residual stream: bf16 = what the reference's autocast runs there (transformer.py:794, layers.py:23-26; stream and "
                         "its gradient in bf16), fp32 = the stricter native form, bf16-fp32grad = bf16 stream with an fp32 residual-gradient path")
    ap.add_argument("--siglip", action="store_true", help="SigLIPTask-equivalent step (sigmoid pairwise loss, logit_bias; BASELINE config 5)")
    ap.add_argument("--naive-global-loss", action="store_true", help="N>1: every rank evaluates the full N x N logits (the reference's "
                    "redundant form) instead of its own rows (same loss and gradients; tests/test_dist_loss_gloo.py, test_ddp_gpu.py)")
    ap.add_argument("--gemm-variant", type=int, default=0, help="developer: value for ocn_set_gemm_variant (kernel choice / ablation knobs)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE", help="developer: ocn_set_tuning(KEY, VALUE) before the run")
    ap.add_argument("--dist-backend", default="nccl", help="developer: 'gloo' + OCN_BENCH_ONE_DEVICE=1 runs N ranks on one GPU")
    return ap.parse_args()


NT_KERNEL = {0: "gemm_nt5_kernel<0,false,0> (plain bf16 out)", 1: "gemm_nt5_kernel<1,false,2> (bias + GELU, saves gelu' in 8 bits)",
             2: "gemm_nt5_kernel<2,false,8> (bias + fp32 residual)", 3: "gemm_nt5_kernel<3,false,10> (x saved 8-bit gelu')",
             4: "gemm_nt5_kernel<4,false,0> (fp32 out)", 5: "gemm_nt5_kernel<5,false,0> (logits: CE statistics)",
             6: "gemm_nt5_kernel<6,false,0> (logits: CE gradient)", 8: "gemm_nt5_kernel<8,false,8> (bias + bf16 residual)"}


class GemmTimer:
    """HIP events around every ocn_gemm_nt / ocn_gemm_tn_accum launch (same stream as the launch), grouped by kernel instantiation
    (the names rocprofv3 shows: the NT kernel's epilogue / cache-policy template arguments, the TN kernel with / without bias row).
    Each record carries the launch's algorithmic FLOPs (2 M N K) and algorithmic BYTES (every operand and every output once:
    DESIGN.md section 4), so that a kernel can be priced against both roofs."""

    def __init__(self):
        self.rec = []
        self.on = False

    def install(self):
        from open_clip_amd import ops
        nt, tn, tn2 = ops.gemm_nt, ops.gemm_tn_accum, ops.gemm_tn_accum2
        timer = self

        def timed(kind, name, flops, nbytes, fn, *args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*args, **kw)
            e1.record()
            timer.rec.append((kind, name, flops, e0, e1, nbytes))
            return r

        def gemm_nt(epi, a, b, out, **kw):
            if not timer.on:
                return nt(epi, a, b, out, **kw)
            M, K, N = a.shape[0], a.shape[1], b.shape[0]
            name = "gemm_nt5_kernel<0,false,2> (bf16 out with N >= 1024, non-temporal stores: the QKV projections of ViT-B-32)" if (epi == 0 and N >= 1024) else NT_KERNEL.get(epi, f"gemm_nt epi {epi}")
            if epi == 3 and N < 1024:
                name = "gemm_nt5_kernel<3,false,8> (x saved 8-bit gelu', N < 1024)"
            nbytes = 2.0 * M * K + 2.0 * N * K + M * N * (4.0 if epi in (2, 4) else 2.0)  # A, B once; the output
            nbytes += (4.0 * M * N if epi == 2 else 0.0) + (2.0 * M * N if epi == 8 else 0.0) + (1.0 * M * N if epi in (1, 3) else 0.0)  # fp32 / bf16 residual read; 8-bit gelu' written / read
            return timed("nt", name, 2.0 * M * N * K, nbytes, nt, epi, a, b, out, **kw)

        def gemm_tn(a, b, dw, dbias=None, *rest, **kw):
            if not timer.on:
                return tn(a, b, dw, dbias, *rest, **kw)
            M, N, K = a.shape[0], a.shape[1], b.shape[1]
            name = "gemm_tn5_kernel<true> (wgrad + bias gradient)" if dbias is not None else "gemm_tn5_kernel<false> (wgrad)"
            return timed("tn", name, 2.0 * M * N * K, 2.0 * M * (N + K) + 4.0 * N * K, tn, a, b, dw, dbias, *rest, **kw)

        def gemm_tn2(a1, b1, dw1, db1, a2, b2, dw2, db2, *rest, **kw):
            if not timer.on:
                return tn2(a1, b1, dw1, db1, a2, b2, dw2, db2, *rest, **kw)
            M, K, N = a1.shape[0], b1.shape[1], a1.shape[1] + a2.shape[1]
            name = "gemm_tn5_kernel<true> (wgrad + bias gradient)" if db1 is not None else "gemm_tn5_kernel<false> (wgrad)"
            return timed("tn", name, 2.0 * M * N * K, 2.0 * M * (N + 2 * K) + 4.0 * N * K, tn2, a1, b1, dw1, db1, a2, b2, dw2, db2, *rest, **kw)

        ops.gemm_nt, ops.gemm_tn_accum, ops.gemm_tn_accum2 = gemm_nt, gemm_tn, gemm_tn2

    @staticmethod
    def _agg(rows):
        fl, ms, n, by = sum(r[2] for r in rows), sum(r[3].elapsed_time(r[4]) for r in rows), len(rows), sum(r[5] for r in rows)
        return {"launches": n, "tflop": fl / 1e12, "ms": ms, "tflops": (fl / 1e12) / (ms / 1e3) if ms > 0 else 0.0,
                "gbytes": by / 1e9, "gbps": (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0}

    def summary(self):
        out = {kind: self._agg([r for r in self.rec if r[0] == kind]) for kind in ("nt", "tn")}
        out["all"] = self._agg(self.rec)
        out["by_kernel"] = {name: self._agg([r for r in self.rec if r[1] == name]) for name in sorted({r[1] for r in self.rec})}
        return out


def roof(a):
    """price an aggregate against BOTH roofs: the nearer one is the kernel's bound.  HBM fraction = algorithmic bytes / time over the
    6.3 TB/s a copy kernel achieves on this chip (the fraction of the 8 TB/s spec rides beside it)"""
    fm = a["tflops"] / PEAK_BF16_TFLOPS
    fh = a["gbps"] / HBM_ACHIEVABLE_GBPS
    return {"bound": "mfma" if fm >= fh else "hbm", "frac": round(max(fm, fh), 4), "frac_mfma": round(fm, 4), "frac_hbm": round(fh, 4),
            "frac_hbm_of_spec": round(a["gbps"] / HBM_SPEC_GBPS, 4), "algorithmic_gb_per_s": round(a["gbps"], 1)}


def _reference_cpu_record():
    """the reference's own train_one_epoch timed in the build container in round 2 (kept beside the live number: another host)"""
    path = os.path.join(ROOT, "profiles", "r02_reference_cpu_train_one_epoch.json")
    if not os.path.exists(path):
        return None
    r = json.load(open(path))
    return {"value": r["pairs_per_s"], "unit": "pairs/s", "kind": "reference", "cores": r["torch_threads"], "nproc": r["nproc"], "where": r["where"],
            "what": r["what"], "oracle_port_same_process_pairs_per_s": r["oracle_port_same_process"]["pairs_per_s"]}


def _reference_cpu_live(threads, steps=8, timeout=180):
    """the REFERENCE's own ``open_clip_train.train.train_one_epoch`` (train.py:337) on ITS ``CLIP`` + ``CLIPTask`` + AdamW, ViT-B-32 fp32
    batch 32, on this box's host cores -- in a subprocess (``oracle/ref_cpu_baseline.py``; the reference's packages come from /root/reference or, on
    the GPU box, from ``oracle/_ref/reference_src.zip``: oracle/fetch_ref.py).  None when neither is here or the run fails (the port stays)."""
    import subprocess
    import tempfile
    if not (os.path.isdir("/root/reference/src/open_clip") or os.path.exists(os.path.join(ROOT, "oracle", "_ref", "reference_src.zip"))):
        return None
    out = os.path.join(tempfile.mkdtemp(prefix="ocn_refcpu_"), "ref.json")
    cmd = [sys.executable, "-m", "oracle.ref_cpu_baseline", "--steps", str(steps), "--threads", str(threads), "--out", out, "--no-port"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        if r.returncode != 0 or not os.path.exists(out):
            return {"error": (r.stderr or r.stdout)[-300:]}
        return json.load(open(out))
    except Exception as e:  # the baseline must never take the bench line down with it
        return {"error": repr(e)[:300]}


def cpu_baseline(model_name, seconds=30.0):
    """The CPU baseline beside the native step, on this box's host cores, at the reference's CPU configuration (ViT-B-32 fp32, batch 32):
    ``kind: "reference"`` -- the reference's own train_one_epoch (``_reference_cpu_live``), its pairs/s from the outer wall clock and from the
    reference's own log line -- whenever the reference's packages are here; the CPU oracle (port of the reference path, fwd+bwd+AdamW) is timed
    first either way: its short thread sweep picks the thread count (batch 32 does not scale to every core of a large host), its own pairs/s rides
    along as ``port``, and it IS the baseline (``kind: "port"``) when the reference is not available."""
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

    def one():
        t0 = time.time()
        outs, grads = O.train_forward_backward(batch["image"], batch["text"], {k: p.detach() for k, p in zip(params, plist)}, cfg)
        for p, k in zip(plist, params):
            p.grad = grads[k]
        opt.step()
        return time.time() - t0

    ncpu = os.cpu_count() or 1
    t_start = time.time()
    one()  # first touch (allocator, thread pool)
    sweep = {}
    for th in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu}):  # (batch 32 stops scaling at 8-32 threads on every host seen so far)
        torch.set_num_threads(th)
        one()
        sweep[th] = round(one(), 3)
        if time.time() - t_start > seconds or (len(sweep) >= 2 and sweep[th] > 1.15 * min(sweep.values())):
            break  # out of budget, or past the best setting
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [one() for _ in range(3)]
    med = sorted(times)[len(times) // 2]
    port = {"value": round(bs / med, 2), "unit": "pairs/s", "cores": best, "host_cores": ncpu, "kind": "port",
            "sample": f"{len(times)} steps of CPU-oracle fwd+bwd+AdamW, {model_name} fp32, batch {bs}, {best} threads (sweep s/step: {sweep}); median {med:.2f} s/step"}
    ref = _reference_cpu_live(best) if model_name == "ViT-B-32" else None
    if ref is not None and "pairs_per_s" in ref:
        rec = {"value": ref["pairs_per_s"], "unit": "pairs/s", "cores": ref["torch_threads"], "host_cores": ref["nproc"], "kind": "reference",
               "value_reference_log_line": ref.get("pairs_per_s_reference_log_line"),
               "sample": f"{ref['steps']} steps of the reference's open_clip_train.train.train_one_epoch (train.py:337) on its own CLIP + CLIPTask + AdamW, ViT-B-32 fp32, "
                         f"batch {bs}, world_size 1, {ref['torch_threads']} threads (chosen by the port's sweep), synthetic in-memory batches; median of the warm "
                         f"steps {ref['sec_per_step_median_warm']:.2f} s/step by an outer wall clock; `value_reference_log_line` = median of the reference's own "
                         f"console `N/s` (train.py:456); {ref['where']}",
               "sec_per_step_all": ref.get("sec_per_step_all"), "port": port}
    else:
        rec = dict(port)
        if ref is not None:
            rec["reference_run_error"] = ref.get("error")
    old = _reference_cpu_record()
    if old is not None:
        rec["reference_in_build_container_round2"] = old
    return rec


def torch_eager_baseline(model_name, batch_size, dev):
    """the same training step as plain PyTorch-ROCm eager ops under bf16 autocast on this GPU (oracle/torch_eager.py)"""
    from oracle import torch_eager
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.synth import init_state_dict, synthetic_batch
    cfg = get_model_config(model_name)
    batch = synthetic_batch(cfg, batch_size, seed=1234, device=dev)
    torch.cuda.reset_peak_memory_stats()
    sec, loss, peak = torch_eager.time_step(cfg, init_state_dict(cfg, seed=0), batch, steps=5, warmup=2)
    return {"value": round(batch_size / sec, 1), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 2), "ms_per_step_is": "median of 5 separately timed steps",
            "s_per_step_sorted": getattr(torch_eager.time_step, "last_all", None), "batch": batch_size, "final_loss": round(loss, 4),
            "peak_hbm_gb": round(peak / 1e9, 1),
            "what": "same step (ViT tower + text tower + ClipLoss + backward + torch.optim.AdamW + clamp) as plain PyTorch-ROCm eager ops under "
                    "torch.amp.autocast(bf16): F.linear / F.scaled_dot_product_attention / F.layer_norm / F.gelu / F.cross_entropy on this GPU "
                    "(oracle/torch_eager.py; the reference itself is not on the GPU box)"}


class ClockSampler:
    """shader clock and socket power of GPU 0 while the timed steps run, read with ``rocm-smi`` from a host thread (rank 0, one GPU only): the
    2.5 PFLOP/s MFMA peak of the roofline is the 2.4 GHz figure, and this step runs the socket into its power limit (profiles/
    r04_clock_power_under_load.txt).  Reported next to the roofline, never used to rescale ``peak`` or ``frac``.  Costs nothing on the GPU:
    the thread sleeps between two ``rocm-smi`` child processes; any failure (tool missing, format changed) yields ``None``."""

    def __init__(self, period=0.5):
        import threading
        self.period, self.rows, self._stop = period, [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def read():
        import re
        import subprocess
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        clk = re.search(r"sclk clock level:[^\n]*\((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        return (int(clk.group(1)) if clk else None, float(pw.group(1)) if pw else None)

    def _run(self):
        while not self._stop.is_set():
            try:
                self.rows.append(self.read())
            except Exception:
                return
            self._stop.wait(self.period)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=10)
        clk = [c for c, _ in self.rows if c]
        pw = [w for _, w in self.rows if w]
        if not clk:
            return None
        mean = sum(clk) / len(clk)
        return {"sclk_mhz_mean": round(mean), "sclk_mhz_min": min(clk), "sclk_mhz_max": max(clk),
                "socket_power_w_mean": round(sum(pw) / len(pw)) if pw else None, "samples": len(clk),
                "mfma_dense_bf16_tflops_at_this_clock": round(2500.0 * mean / 2400.0, 1),
                "how": "rocm-smi --showclocks --showpower every 0.5 s from a host thread during the K timed steps; the roofline's peak stays the 2.4 GHz "
                       "figure (2.5 PFLOP/s), this is what the power-limited clock of THIS run allows"}


def launcher_command(gpus, argv, port=None):
    """the command ``python bench.py --gpus N ...`` turns into when no launcher started it: one rank per GPU under ``torch.distributed.run`` on
    this node (the reference's launch contract, open_clip_train/distributed.py:80-166: torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), the
    rendezvous on 127.0.0.1 and a free port"""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def check_launch(gpus, env):
    """-> "spawn" (``--gpus N`` > 1 and no launcher: this process must start the N ranks itself), "rank" (started by a launcher whose
    WORLD_SIZE is N) or "single" (N = 1, no launcher).  Anything else is an error: a line that claims N GPUs is never timed on fewer."""
    launched = "RANK" in env or "LOCAL_RANK" in env
    world = int(env.get("WORLD_SIZE", "1"))
    if not launched:
        if world != 1:
            raise SystemExit(f"bench.py: WORLD_SIZE={world} without RANK / LOCAL_RANK in the environment: not a torchrun launch")
        return "spawn" if gpus > 1 else "single"
    if world != gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {gpus}: the line would not describe the run")
    return "rank" if world > 1 else "single"


def dist_facts(world, dev, native_comm, elapsed, backend):
    """what the TRANSPORTS report about the run, for the line of an N > 1 run (collective; every rank calls it): the process group's world size,
    the rank count of the RCCL communicator itself (ncclCommCount through ocn_comm_count; None when the ranks do not talk RCCL: gloo developer modes),
    the number of ranks that took part in a collective on the data path (an all-reduce of ones) and the per-rank times of the timed region"""
    import torch.distributed as dist
    ones = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(ones)
    times = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(times, torch.tensor([elapsed], device=dev, dtype=torch.float64))
    times = [float(t) for t in times]
    rccl = None
    if native_comm is not None:
        rccl = native_comm.count()[0]
        check = torch.ones(1, device=dev, dtype=torch.float32)
        native_comm.all_reduce_sum(check)  # the same count through the C ABI's own collective
        rccl = rccl if int(check.item()) == rccl else -1
    return {"dist_world_size": dist.get_world_size(), "transport_ranks": int(round(float(ones))), "rccl_ranks": rccl,
            "rccl_ranks_is": ("ncclCommCount of the C ABI's communicator (ocn_comm_count), confirmed by an ocn_comm_allreduce_sum of ones" if rccl is not None else
                              f"not available: backend {backend!r} / no C-ABI communicator in this run (torch's process group does not expose ncclCommCount)"),
            "elapsed_s_per_rank_min": round(min(times), 4), "elapsed_s_per_rank_max": round(max(times), 4)}


def main():
    args = parse()
    mode = check_launch(args.gpus, os.environ)
    if mode == "spawn":
        # `python bench.py --gpus N` as the driver calls it: start the N ranks here and hand their exit code on; rank 0 prints the one JSON line
        import subprocess
        raise SystemExit(subprocess.run(launcher_command(args.gpus, sys.argv[1:]), env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    one_device = os.environ.get("OCN_BENCH_ONE_DEVICE") == "1"  # developer mode: every rank on GPU 0 (needs --dist-backend gloo)
    if one_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but this node shows {torch.cuda.device_count()} GPU(s): one rank per GPU is the contract "
                         "(OCN_BENCH_ONE_DEVICE=1 with --dist-backend gloo is the developer mode that shares one)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world == 1 and args.force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29777")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    rank_devices = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
        assert dist.get_world_size() == args.gpus
        # rank -> device map on the line: the line's n_gpus is what the process group itself reports
        mine = {"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(local_rank),
                "pci_bus_id": getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None)}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)

    line = run(args, rank, local_rank, world, dev, rank_devices, one_device)
    if line is not None:
        # BASELINE configs 4 and 5 on the same line: after the timed ViT-B-32 steps and their baselines, with this process's HBM
        # handed back first (run() has returned: model, optimizer and activations are gone), each in a process of its own
        want = (world == 1 and args.model == "ViT-B-32" and not args.siglip and args.local_batch == 4096 and args.accum_freq == 1 and not args.h2d
                and not args.grad_checkpointing and args.data_ranks == 1 and not args.no_config_lines and not args.no_extra_lines)
        if want:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            line["config4_vitl14"] = config_line(["--model", "ViT-L-14", "--local-batch", "2048", "--grad-checkpointing"],
                                                 "BASELINE config 4 on one GPU: ViT-L-14, local batch 2048 (gbs 16384 / 8), block recompute "
                                                 "(--grad-checkpointing; --keep-blocks auto keeps what the free HBM holds), ClipLoss")
            line["config5_vith14_siglip"] = config_line(["--model", "ViT-H-14", "--siglip", "--local-batch", "1024", "--grad-checkpointing"],
                                                        "BASELINE config 5 on one GPU: ViT-H-14 + SigLipLoss, local batch 1024 (gbs 8192 / 8), block recompute")
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_ddp:
        torch.distributed.destroy_process_group()


def config_line(extra_args, what, steps=4, warmup=1, timeout=300):
    """one of the other BASELINE configurations through this same file in a subprocess (fresh HBM: --keep-blocks auto plans against what is free):
    `steps` timed steps of which the first carries HIP events on every GEMM launch (its towers one at a time), so the record has the per-kernel
    roofline of THAT model; returns the sub-record that rides on the default line, or {"error": ...} -- a sample never takes the line down"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-eager-baseline",
           "--no-dense-text-line", "--no-extra-lines", "--no-clock-sample", "--no-config-lines"] + extra_args
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or len(lines) != 1:
            return {"error": (r.stderr or r.stdout)[-400:], "returncode": r.returncode}
        d = json.loads(lines[0])
        rf = d.get("roofline", {})
        keep = ("kernel", "achieved", "bound", "frac", "frac_mfma", "frac_hbm", "launches", "avg_launch_ms", "share_of_gemm_time")
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "what": what,
                "model_tflops_per_gpu": d.get("step_model_tflops_per_gpu", d.get("step_dense_equivalent_model_tflops_per_gpu")),
                "mfu_of_2500_tflops": round((d.get("step_model_tflops_per_gpu") or d.get("step_dense_equivalent_model_tflops_per_gpu") or 0.0) / PEAK_BF16_TFLOPS, 4),
                "local_batch": d["config"]["local_batch"], "grad_checkpointing": d["config"]["grad_checkpointing"], "final_loss": d["config"]["final_loss"],
                "peak_hbm_gb": d.get("peak_hbm_gb_rank0"), "wall_s": round(time.perf_counter() - t0, 1),
                "note": f"{steps} timed steps, the first event-timed with the towers one at a time (its GEMM launches carry HIP events: the roofline below); "
                        "a sanity sample beside the headline metric, not the metric",
                "roofline": {"dominant": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_mfma", "frac_hbm", "avg_launch_ms", "launches")},
                             "by_kernel": [{k: e.get(k) for k in keep} for e in rf.get("by_kernel", [])[:8]],
                             "all_gemm_launches": rf.get("all_gemm_launches"), "gemm_share_of_step": rf.get("gemm_share_of_step")}}
    except Exception as e:
        return {"error": repr(e)[:400]}


def run(args, rank, local_rank, world, dev, rank_devices, one_device):
    """the measured body: builds the model, times the K steps, the extra samples and the baselines; returns rank 0's JSON line (None elsewhere)"""
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
    model = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=True, image_stream=args.image_stream, **extra)
    model.load_state_dict(init_state_dict(cfg, seed=0, siglip=args.siglip))
    model = model.to(dev).train()
    if args.dense_text:
        model.pack_text = False
    if args.serial_towers:
        model.tower_streams = False
    overlap_towers = model.tower_streams
    model.pair_wgrad = not args.no_wgrad_pair
    model.deterministic = args.deterministic
    B = args.local_batch
    F_ACC = max(1, args.accum_freq)
    if args.data_ranks > 1:
        assert world == 1
        parts = [[synthetic_batch(cfg, B, seed=1234 + 1000 * j, rank=r, device=dev) for r in range(args.data_ranks)] for j in range(F_ACC)]
        micro = [{k: torch.cat([p[k] for p in ps]) for k in ("image", "text")} for ps in parts]
        B = B * args.data_ranks
    else:
        micro = [synthetic_batch(cfg, B, seed=1234 + 1000 * j, rank=rank, device=dev) for j in range(F_ACC)]
    batch = micro[0]
    model_ref = model
    ctx_len = cfg["text_cfg"]["context_length"]
    kept_blocks, recompute_share = (0, 0), 0.0
    if args.grad_checkpointing:
        rows_t = max(int((m["text"].argmax(dim=-1) + 1).sum()) for m in micro) if model.pack_text else None
        if args.keep_blocks == "auto":
            # what is not there yet when this runs: gradients, the optimizer's two moments, the bf16 operand copies (16 B / parameter);
            # 15 % of the rest stays free for the allocator's fragmentation, the loss and the input batches
            free = torch.cuda.mem_get_info(dev)[0] - 16 * sum(p.numel() for p in model.parameters())
            kept_blocks = model.plan_grad_checkpointing(B, int(0.85 * free), text_rows=rows_t)
        else:
            kb = [int(v) for v in args.keep_blocks.split(",")]
            kept_blocks = (kb[0], kb[-1])
            model.set_grad_checkpointing(True, keep_last=kept_blocks)
        bv, bt = model.activation_bytes_per_block(B, rows_t)  # bytes ~ rows x width: the towers' FLOPs split (rows x width^2) follows from them
        nv, nt = len(model.visual.transformer.resblocks), len(model.transformer.resblocks)
        fv, ft = bv * model.visual.transformer.width, bt * model.transformer.width
        recompute_share = ((nv - min(nv, kept_blocks[0])) * fv + (nt - min(nt, kept_blocks[1])) * ft) / (nv * fv + nt * ft)
    if model.pack_text:
        kept = sum(int((m["text"].argmax(dim=-1) + 1).sum()) for m in micro)
        text_rows_note = (f"packed: only the tokens up to the pooled EOT exist in the text tower ({kept // len(micro)} of {B * ctx_len} rows per batch, mean caption "
                          f"{kept / (len(micro) * B):.1f} of {ctx_len} tokens; synthetic EOT position ~ U[8, {ctx_len - 1}]); features / loss / gradients "
                          f"identical to the padded tower (tests/test_model_gpu.py::test_packed_text_tower_equals_dense_text_tower); --dense-text runs all rows")
    else:
        text_rows_note = f"dense: all {ctx_len} positions of every caption (as the reference)"
    from open_clip_amd.model import _pooled_last_block_ok
    last_block_note = ("out-projection, LN2 and MLP of each tower's last block on the pooled rows only (the only rows the poolers read; same features and "
                       "gradients, tests/test_model_gpu.py::test_pooled_last_block_equals_full_block); pooled_last_block=False runs every row"
                       if (_pooled_last_block_ok(model) and _pooled_last_block_ok(model.visual)) else "every row (as the reference)")
    pipe = None
    if args.h2d:
        # decoded pixels as a loader hands them over: uint8 [B,H,W,3] in pinned host memory (synthetic; a pool of 2 batches is cycled)
        from open_clip_amd.input_pipeline import DeviceBatchPipeline
        S = cfg["vision_cfg"]["image_size"]
        gh = torch.Generator().manual_seed(99 + rank)
        host_pool = [(torch.randint(0, 256, (B, S, S, 3), generator=gh, dtype=torch.uint8).pin_memory(), micro[j % F_ACC]["text"].cpu().pin_memory())
                     for j in range(2)]
        # the packed text layout is computed on the host with the batch (HostTextPlan): the step has no host synchronisation left
        pipe = DeviceBatchPipeline(dev, (B, S, S, 3), (B, cfg["text_cfg"]["context_length"]), depth=max(2, F_ACC + 1),
                                   plan_text_vocab=(cfg["text_cfg"]["vocab_size"] if model.pack_text else None), attn_buckets=model.attn_buckets)
        pipe.submit(*host_pool[0])
    if args.native:
        args.native_comm = args.native_allreduce = True
    tile_rescue = args.tile_rescue == "on" or (args.tile_rescue == "auto" and world > 1)
    from open_clip_amd import ops as _ops
    if not tile_rescue:
        os.environ["OCN_TILE_RESCUE"] = "0"  # (the distributed losses / NativeGradSync would switch it on for world_size > 1: ops.multi_gpu_defaults)
    _ops.set_tile_rescue(tile_rescue)
    # N > 1 over RCCL: the loss collectives go through the C ABI's communicator by default (north_star: "an RCCL all-gather ... through a thin C-ABI
    # extension"); --torch-comm keeps them on the process group, --native moves the gradient all-reduce there as well
    want_loss_comm = args.native_comm or (world > 1 and args.dist_backend == "nccl" and not args.torch_comm)
    native_comm, native_comm_error = None, None
    if (args.native_allreduce or want_loss_comm) and args.dist_backend == "nccl":
        from open_clip_amd.comm import NativeComm  # RCCL behind the C ABI; the 128-byte id travels through the process group once
        try:
            native_comm = NativeComm.from_process_group(rank, world) if world > 1 else NativeComm(NativeComm.make_unique_id(), 0, 1)
        except Exception as e:  # ocn_comm_init failed here: fall back to torch.distributed + DDP (every rank must take the same branch: agreed below)
            native_comm_error = repr(e)[:300]
        if world > 1:
            ok = torch.tensor([0 if native_comm is None else 1], device=dev, dtype=torch.int32)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if int(ok) == 0 and native_comm is not None:
                native_comm.close()
                native_comm, native_comm_error = None, "ocn_comm_init failed on another rank"
        if native_comm is None:
            if rank == 0:
                print(f"bench.py: the C ABI's RCCL communicator could not be created ({native_comm_error}): falling back to torch.distributed + DDP", file=sys.stderr)
            args.native_allreduce = False
    loss_comm = native_comm if (want_loss_comm and native_comm is not None) else None  # with one process: a one-rank communicator, the loss still runs its distributed form
    if args.siglip:
        from open_clip_amd.loss import NativeSigLipLoss
        loss_fn = NativeSigLipLoss(rank=rank, world_size=world, comm=loss_comm, deterministic=args.deterministic)
    else:
        loss_fn = NativeClipLoss(local_loss=False, gather_with_grad=False, rank=rank, world_size=world,
                                 row_sharded=((world > 1 or loss_comm is not None) and not args.naive_global_loss), comm=loss_comm, deterministic=args.deterministic)
    opt = NativeAdamW(param_groups_like_reference(model, 0.2), lr=args.lr, betas=(0.9, 0.98), eps=1e-6, weight_caches=weight_caches_of(model))
    net = model
    grad_sync = None
    if args.native_allreduce:
        from open_clip_amd.grad_sync import NativeGradSync
        grad_sync = NativeGradSync(model, world, comm=native_comm)
    elif world > 1 or args.force_ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], bucket_cap_mb=args.bucket_cap_mb, gradient_as_bucket_view=True)

    timer = GemmTimer()
    if not args.no_roofline:
        timer.install()

    step_no = [0]

    def micro_batches(mb):
        """device batches of this optimizer step; with --h2d each one arrives from pinned host memory while the previous one computes"""
        for j in range(len(mb)):
            if pipe is None:
                yield mb[j]
            else:
                b = pipe.next()
                pipe.submit(*host_pool[(step_no[0] * len(mb) + j + 1) % 2])  # next batch's copy runs under this batch's compute
                yield b

    def step(mb=None):
        """one optimizer step over the micro-batches ``mb`` (default: the F_ACC batches of this run)"""
        mb = micro if mb is None else mb
        # linear warm-up of the reference's schedule (scheduler.py:6-15: lr * (step + 1) / warmup_length)
        lr_t = args.lr * min(1.0, (step_no[0] + 1) / max(1, args.lr_warmup_steps))
        for g in opt.param_groups:
            g["lr"] = lr_t
        opt.zero_grad(set_to_none=True)
        if len(mb) == 1:
            b = next(iter(micro_batches(mb)))
            out = net(image=b["image"], text=b["text"])
            loss = loss_fn(**out)
            loss.backward()
            if grad_sync is not None:
                grad_sync.finish()
            if pipe is not None:
                pipe.release(b)
        else:
            # train.py:236-311: features of every micro-batch under no_grad, then every micro-batch again with gradient, the loss taken
            # over the concatenation (cached features stand in for the other micro-batches)
            held = list(micro_batches(mb))
            feats = {"image_features": [], "text_features": []}
            with torch.no_grad():
                for b in held:
                    o = net(image=b["image"], text=b["text"])
                    for k in feats:
                        feats[k].append(o[k])
            for j, b in enumerate(held):
                o = net(image=b["image"], text=b["text"])
                inputs = {k: torch.cat(feats[k][:j] + [o[k]] + feats[k][j + 1:]) for k in feats}
                extra_in = {"logit_bias": o["logit_bias"]} if "logit_bias" in o else {}
                loss = loss_fn(**inputs, logit_scale=o["logit_scale"], **extra_in)
                if grad_sync is not None and j + 1 < len(held):
                    with grad_sync.no_sync():  # the micro-batches before the last one only accumulate (DDP: no_sync)
                        loss.backward()
                else:
                    loss.backward()
                if pipe is not None:
                    pipe.release(b)
            if grad_sync is not None:
                grad_sync.finish()
        opt.step()
        with torch.no_grad():
            model.logit_scale.clamp_(0, math.log(100))  # image_text_task.py:91-101
        step_no[0] += 1
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        loss = step()
    barrier()
    # GEMM launches carry HIP events on every EV-th timed step.  On those steps the towers run ONE AT A TIME (model.tower_streams =
    # "serial": the same two streams and allocator pools as the shipped configuration, the image tower's stream held back behind the
    # text tower's in forward and backward) with nothing beside a GEMM, so that a launch's duration is the kernel's own (with the
    # towers overlapped two kernels share the chip and an event pair times the mix); the other steps run as shipped.  Every step is
    # inside the timed region and counts in ``value``.
    EV = 20 if overlap_towers else 2  # (an event-timed step runs its towers one at a time: ~8 ms slower than a shipped step, and it counts in `value`)
    timed_steps = 0
    sampler = ClockSampler().start() if (rank == 0 and world == 1 and not args.no_clock_sample) else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.on = (not args.no_roofline) and (i % EV == 0)
        timed_steps += int(timer.on)
        model.tower_streams = ("serial" if timer.on else True) if overlap_towers else False
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    clock_under_load = sampler.stop() if sampler is not None else None
    timer.on = False
    model.tower_streams = overlap_towers
    facts = None
    if world > 1:
        facts = dist_facts(world, dev, native_comm, elapsed, args.dist_backend)
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())
    if rank == 0:
        # the headline, as soon as the K timed steps are over: the one JSON line on stdout also carries the auxiliary samples and baselines
        # (minutes of further work) -- a driver that kills the process in that window still finds the measured value here
        print("bench.py headline (timed steps done; the JSON line follows on stdout): " + json.dumps(
            {"value": round(B * F_ACC * world / (elapsed / args.steps), 1), "unit": "pairs/s", "ms_per_step": round(elapsed / args.steps * 1e3, 2),
             "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "image_stream": args.image_stream}), file=sys.stderr, flush=True)

    # the same loop with every caption padded to context_length (what the reference executes), timed right behind the packed one so
    # that both numbers come from the same box and process; outside the K timed steps
    dense_text = None
    peak_bytes = torch.cuda.max_memory_allocated()
    if model.pack_text and world == 1 and not args.no_dense_text_line:
        model.pack_text = False
        n_dense = max(2, min(5, args.steps))
        step()
        torch.cuda.synchronize()
        td = time.perf_counter()
        for _ in range(n_dense):
            step()
        torch.cuda.synchronize()
        td = (time.perf_counter() - td) / n_dense
        model.pack_text = True
        dense_text = {"value": round(B * F_ACC / td, 1), "unit": "pairs/s", "ms_per_step": round(td * 1e3, 2), "steps": n_dense,
                      "what": "same step with --dense-text (all context_length positions of every caption through the text tower)"}

    def sample(n, mb=None):
        step(mb)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            step(mb)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    # Two more samples on the default line, same process, outside the K timed steps:
    #   reference_work    every row the reference executes: all context_length positions of every caption AND the full last block of both towers
    #                     (no packed text rows, no pooled last block) -- the native kernels on exactly the reference's work
    #   accum8_gbs32768   the metric's own global batch on ONE GPU with the reference's accumulation semantics (train.py:236-311): a no-grad feature
    #                     pass over 8 micro-batches of 4096, then each micro-batch again with gradient against the 32768 x 32768 logits
    reference_work = accum8 = fp32_stream = None
    extra_ok = (world == 1 and not args.no_extra_lines and not args.no_dense_text_line and args.model == "ViT-B-32" and not args.siglip and F_ACC == 1
                and pipe is None and not args.grad_checkpointing and args.data_ranks == 1)
    if extra_ok and model.pack_text:
        saved_flags = (model.pack_text, model.pooled_last_block, model.visual.pooled_last_block)
        try:
            model.pack_text = False
            model.pooled_last_block = model.visual.pooled_last_block = False
            td = sample(3)
            reference_work = {"value": round(B / td, 1), "unit": "pairs/s", "ms_per_step": round(td * 1e3, 2), "steps": 3,
                              "what": "same step on exactly the rows the reference executes: --dense-text AND pooled_last_block=False (every position of every "
                                      "caption through the text tower, the last block of both towers on every row)"}
        except Exception as e:  # a sample must never take the bench line down with it
            reference_work = {"error": repr(e)[:300]}
        finally:
            model.pack_text, model.pooled_last_block, model.visual.pooled_last_block = saved_flags
    if extra_ok and model.image_stream != "fp32":
        try:  # the same step with the image tower's residual stream in fp32 (the stricter native form; the headline runs the reference's bf16 policy there)
            model.image_stream = "fp32"
            td = sample(4)
            fp32_stream = {"value": round(B / td, 1), "unit": "pairs/s", "ms_per_step": round(td * 1e3, 2), "steps": 4,
                           "what": "same step with --image-stream fp32: the image tower's residual stream (and its gradient) in fp32 instead of the bf16 the "
                                   "reference's autocast runs there (transformer.py:794, layers.py:23-26)"}
        except Exception as e:
            fp32_stream = {"error": repr(e)[:300]}
        finally:
            model.image_stream = args.image_stream
    if extra_ok and B == 4096:
        try:
            mb8 = micro + [synthetic_batch(cfg, B, seed=1234 + 1000 * j, rank=rank, device=dev) for j in range(1, 8)]
            td = sample(2, mb8)
            accum8 = {"value": round(8 * B / td, 1), "unit": "pairs/s", "ms_per_step": round(td * 1e3, 1), "steps": 2, "global_batch": 8 * B, "accum_freq": 8,
                      "what": "the metric's global batch 32768 on ONE GPU: --accum-freq 8 with the reference's semantics (train.py:236-311: features of the 8 "
                              "micro-batches of 4096 under no_grad, then every micro-batch again with gradient against the 32768 x 32768 logits), one "
                              "optimizer step per 32768 pairs; 4 forward-equivalents per pair instead of 3"}
            del mb8
        except Exception as e:
            accum8 = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()

    line = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * F_ACC * world / (elapsed / args.steps)
        # forward passes per pair: 1 (+1 recompute with grad checkpointing) (+1 no_grad feature pass with accumulation) + 2 for the backward
        # (the recompute counts for the share of the towers' block FLOPs that is actually recomputed: --keep-blocks)
        from open_clip_amd.configs import forward_gflops_per_pair
        flops_pair = FWD_GFLOP_PER_PAIR.get(args.model, forward_gflops_per_pair(cfg)) * (3 + (recompute_share if args.grad_checkpointing else 0) + (1 if F_ACC > 1 else 0))
        line = {
            "metric": ("image-text pairs/sec (whole node), ViT-B-32 gbs=32768 at 1/2/4/8 GPUs" if args.model == "ViT-B-32" and not args.siglip
                       else f"image-text pairs/sec (whole node), {args.model}{' SigLIP' if args.siglip else ''}"), "value": round(value, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model} {'SigLIPTask' if args.siglip else 'CLIPTask'}-equivalent train step (fwd+{'SigLipLoss' if args.siglip else 'ClipLoss'}+bwd+AdamW+clamp), amp_bf16 policy, "
                                   f"local_bs={B}, " + (f"accum_freq={F_ACC} (train.py:236-311), " if F_ACC > 1 else "") + f"global_bs={B * F_ACC * world}, "
                                   + ("inputs from pinned host memory every step (uint8 pixels, async double-buffered H2D inside the timed region), " if args.h2d else "")
                                   + (("gather_features all-gather + global logits" + ("" if args.naive_global_loss else " (row-sharded across ranks)"))
                                      if world > 1 else "world_size 1 (no all-gather)"),
                       "dist_world_size": (facts["dist_world_size"] if facts else 1),
                       "rccl_ranks": (facts["rccl_ranks"] if facts else (native_comm.count()[0] if native_comm is not None else None)),
                       "transport_ranks": (facts["transport_ranks"] if facts else 1),
                       "rccl_ranks_is": (facts["rccl_ranks_is"] if facts else "one process"),
                       "elapsed_s_per_rank": ({"min": facts["elapsed_s_per_rank_min"], "max": facts["elapsed_s_per_rank_max"]} if facts else None),
                       "native_comm_fallback": native_comm_error,
                       "dist_backend": ((("rccl (torch.distributed 'nccl')" if args.dist_backend == "nccl" else args.dist_backend) if world > 1 else "none")),
                       "rank_devices": rank_devices, "one_device_developer_mode": one_device,
                       "model": args.model, "global_batch": B * F_ACC * world, "local_batch": B, "accum_freq": F_ACC, "parallelism": f"dp{world}",
                       "ddp": bool((world > 1 or args.force_ddp) and grad_sync is None), "bucket_cap_mb": args.bucket_cap_mb,
                       "gradient_allreduce": ("native per-block in-place all-reduce (open_clip_amd/grad_sync.py)" + (" over RCCL through the C ABI" if native_comm is not None else " over the process group")
                                              if grad_sync is not None else ("DistributedDataParallel" if (world > 1 or args.force_ddp) else "none (one process)")),
                       "loss_collectives": ("C ABI (ocn_comm_*)" + (" on a one-rank communicator" if world == 1 else "")) if loss_comm is not None else ("torch.distributed" if world > 1 else "none"),
                       "gemm_tile_rescue": ("on: finishing workgroups take over the shares of workgroups that have not started (CUs held by collectives' kernels)"
                                            if tile_rescue else "off (static shares: one process, nothing else holds CUs)"),
                       "lr": args.lr, "lr_warmup_steps": args.lr_warmup_steps, "input": "host_uint8_h2d" if args.h2d else "resident",
                       "image_residual_stream": {"fp32": "fp32 (stricter than the reference's autocast)",
                                                 "bf16": "bf16, stream and gradient (what the reference's autocast runs in the image tower: transformer.py:794, layers.py:23-26)",
                                                 "bf16-fp32grad": "bf16 stream, fp32 residual-gradient path"}[args.image_stream],
                       "text_residual_stream": "fp32 (as under the reference's autocast: the token embedding is fp32, model.py:399-401)",
                       "text_tower": text_rows_note,
                       "tower_streams": ("image tower on its own stream next to the text tower" if overlap_towers else "one stream"),
                       "last_block": last_block_note,
                       "grad_checkpointing": (f"block recompute (transformer.py:577-585) for all but the last {kept_blocks[0]} image / {kept_blocks[1]} text blocks, whose "
                                              f"activations stay in HBM (--keep-blocks {args.keep_blocks}); {recompute_share:.2f} of a forward is recomputed per step"
                                              if args.grad_checkpointing else "off"),
                       "random_init_weights": True, "final_loss": round(final_loss, 4)},
            # FLOPs of the model as the reference runs it (every caption padded to context_length); the packed text tower executes fewer
            ("step_model_tflops_per_gpu" if not model_ref.pack_text else "step_dense_equivalent_model_tflops_per_gpu"): round(value / world * flops_pair / 1e3, 1),
            "peak_hbm_gb_rank0": round(peak_bytes / 1e9, 1), "reserved_hbm_gb_rank0": round(torch.cuda.max_memory_reserved() / 1e9, 1),
        }
        if clock_under_load is not None:
            line["clock_under_load"] = clock_under_load
        if dense_text is not None:
            line["dense_text_tower"] = dense_text
        if fp32_stream is not None:
            line["image_stream_fp32"] = fp32_stream
        if reference_work is not None:
            line["reference_work"] = reference_work
        if accum8 is not None:
            line["accum8_gbs32768"] = accum8
        if not args.no_roofline:
            s = timer.summary()
            nt, tn, allg = s["nt"], s["tn"], s["all"]
            # the dominant kernel = the instantiation with the most time inside the event-timed steps (the same ranking as the rocprofv3
            # summary under profiles/); the other GEMM kernels, the NT family and all GEMM launches together ride beside it
            dom_name, dom = max(s["by_kernel"].items(), key=lambda kv: kv[1]["ms"])
            step_ms_ev = elapsed * 1e3 * timed_steps / args.steps  # (approximate: event-timed steps are a little longer than the others)
            traffic, traffic_src, traffic_by = None, None, {}
            head_sha, csrc_sha = code_identity()
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written by tools/pmc_stats.py from the PMC passes
            if os.path.exists(tpath) and args.model == "ViT-B-32" and B == 4096:
                rec = json.load(open(tpath))
                stale = rec.get("csrc_sha16") != csrc_sha
                traffic_src = {"how": rec["source"], "git_sha": rec.get("git_sha"), "csrc_sha16": rec.get("csrc_sha16"), "launches_profiled": rec.get("launches"),
                               "stale": stale, "note": ("PMC passes taken on OTHER kernel sources than the ones this line ran (csrc hash differs)" if stale else
                                                        "PMC passes taken on exactly the kernel sources this line ran")}
                traffic_by = rec.get("by_kernel", {})
                key = dom_name.split(" ")[0]
                traffic = round(traffic_by[key]["bytes_per_launch"]) if key in traffic_by else round(rec["bytes_per_launch"])

            def krec(name, a):
                r = {"kernel": name, "achieved": round(a["tflops"], 1), **roof(a), "launches": a["launches"],
                     "avg_launch_ms": round(a["ms"] / max(a["launches"], 1), 4), "algorithmic_tflop_per_launch_avg": round(a["tflop"] / max(a["launches"], 1), 4),
                     "algorithmic_gb_per_launch_avg": round(a["gbytes"] / max(a["launches"], 1), 4),
                     "share_of_gemm_time": round(a["ms"] / max(allg["ms"], 1e-9), 3)}
                t = traffic_by.get(name.split(" ")[0])
                if t:
                    r["traffic"] = round(t["bytes_per_launch"])
                return r

            dom_roof = roof(dom)
            is_mfma = dom_roof["bound"] == "mfma"
            line["roofline"] = {"bound": dom_roof["bound"], "kernel": dom_name,
                                "achieved": round(dom["tflops"], 1) if is_mfma else round(dom["gbps"], 1),
                                "peak": PEAK_BF16_TFLOPS if is_mfma else HBM_ACHIEVABLE_GBPS, "unit": "TFLOP/s" if is_mfma else "GB/s",
                                "frac": dom_roof["frac"], "frac_mfma": dom_roof["frac_mfma"], "frac_hbm": dom_roof["frac_hbm"],
                                "peaks": {"mfma_dense_bf16_tflops": PEAK_BF16_TFLOPS, "hbm_achievable_gbps": HBM_ACHIEVABLE_GBPS, "hbm_spec_gbps": HBM_SPEC_GBPS,
                                          "rule": "a kernel's bound is the roof it is nearer to: max(algorithmic FLOPs / time / MFMA peak, algorithmic bytes / time / "
                                                  "6.3 TB/s); both fractions ride on every entry"},
                                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src, "launches": dom["launches"],
                                "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                                "algorithmic_tflop_per_launch_avg": round(dom["tflop"] / max(dom["launches"], 1), 4),
                                "algorithmic_gb_per_launch_avg": round(dom["gbytes"] / max(dom["launches"], 1), 4),
                                "dominant_by": "largest total time among the GEMM kernel instantiations inside the event-timed steps",
                                "kernel_form": ("rescue (the instantiations carry a trailing `true`: gemm_tn5_kernel<.., true>, gemm_nt5_kernel<.., true>; the PMC traffic quoted "
                                                "beside them was taken on the static form: same operand and result bytes)" if tile_rescue else "static"),
                                "by_kernel": [krec(n, a) for n, a in sorted(s["by_kernel"].items(), key=lambda kv: -kv[1]["ms"])],
                                "gemm_nt_family": {"achieved": round(nt["tflops"], 1), **roof(nt), "launches": nt["launches"],
                                                   "avg_launch_ms": round(nt["ms"] / max(nt["launches"], 1), 4)},
                                "gemm_tn_family": {"achieved": round(tn["tflops"], 1), **roof(tn), "launches": tn["launches"],
                                                   "avg_launch_ms": round(tn["ms"] / max(tn["launches"], 1), 4)},
                                "all_gemm_launches": {"achieved": round(allg["tflops"], 1), **roof(allg), "launches": allg["launches"]},
                                "event_timed_steps": timed_steps,
                                "event_timed_steps_mode": ("one tower at a time (same two streams, the image tower's held back behind the text tower's), no wgrad "
                                                           f"side stream: every GEMM launch alone on the chip; the other {args.steps - timed_steps} timed steps "
                                                           "overlap the towers" if overlap_towers else
                                                           "as every step (wgrad launches run on a side stream under the LayerNorm / attention backward kernels of "
                                                           "their block unless --no-wgrad-pair: their events then time a co-scheduled kernel)"),
                                "gemm_share_of_step": round(allg["ms"] / step_ms_ev, 3)}
            line["code"] = {"git_sha": head_sha, "csrc_sha16": csrc_sha}
        if world == 1 and not args.no_eager_baseline and not args.siglip:
            micro.clear()
            del batch
            torch.cuda.empty_cache()
            eb = min(args.eager_batch, B)
            while True:  # the bench's own batch when it fits (like for like); halve on out-of-memory
                try:
                    rec = torch_eager_baseline(args.model, eb, dev)
                    break
                except torch.OutOfMemoryError:
                    torch.cuda.empty_cache()
                    if eb <= 256:
                        rec = {"error": "out of memory at batch 256"}
                        break
                    eb //= 2
                except Exception as e:  # the baseline must never take the bench line down with it
                    rec = {"error": repr(e)[:300]}
                    break
            if "value" in rec:
                # like for like: native and eager both at the bench's batch; `native_dense_over_eager` also runs the SAME work (all
                # context_length positions of every caption), `native_over_eager` includes what the packed text tower saves
                rec["native_over_eager"] = round(value / rec["value"], 2)
                rec["native_over_eager_is"] = "shipped native step (packed text tower, pooled last block) / eager step, both at the batch above" if model_ref.pack_text else "native / eager"
                if dense_text is not None:
                    rec["native_dense_over_eager"] = round(dense_text["value"] / rec["value"], 2)
                    rec["native_dense_over_eager_is"] = "native step with --dense-text (every position of every caption, as eager runs it) / eager step"
                rec["same_batch_as_native"] = bool(eb == B)
            line["torch_eager_baseline"] = rec
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model)
    if grad_sync is not None and rank == 0 and os.environ.get("OCN_BENCH_VERBOSE"):
        print("grad_sync stats:", {k: (v if not isinstance(v, list) else v[-30:]) for k, v in grad_sync.stats.items()}, file=sys.stderr)
    if native_comm is not None:
        native_comm.close()
    return line if rank == 0 else None


if __name__ == "__main__":
    main()
