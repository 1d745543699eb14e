"""bench.py's own code paths on the MI355X at a miniature size (the model 'small-test', a few steps): the world_size-2 path -- two
ranks on the one GPU of the test box over gloo (RCCL refuses two ranks on one device), DistributedDataParallel + the row-sharded
global ClipLoss -- must report the same loss as one process on the concatenation of the two ranks' batches (every rank evaluates the
FULL-batch loss in this mode, and Adam is invariant to the 1/W that DDP's mean puts on the gradients); the gradient-accumulation
mode (--accum-freq, train.py:236-311) and the host-fed mode (--h2d) run and report what they did."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--model", "small-test", "--local-batch", "8", "--steps", "3", "--warmup", "0", "--no-cpu-baseline", "--no-eager-baseline", "--no-roofline",
          "--lr", "1e-3", "--lr-warmup-steps", "1"]


def _bench(extra, nproc=1, env=None, roofline=False, launcher=True):
    e = dict(os.environ, **(env or {}))
    common = [a for a in COMMON if roofline is False or a != "--no-roofline"]
    if nproc == 1 or not launcher:  # launcher=False: plain `python bench.py --gpus N`, as the driver calls it
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + common + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", "29741",
               os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + common + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_world2_reports_the_full_batch_loss():
    # roofline=True: as the driver launches it -- the first timed step carries HIP events on every GEMM and runs the towers one at a time
    # (model.tower_streams = "serial"), the others overlap them; all of it under DistributedDataParallel
    two = _bench(["--dist-backend", "gloo"], nproc=2, env={"OCN_BENCH_ONE_DEVICE": "1"}, roofline=True)
    assert two["roofline"]["event_timed_steps"] == 1 and two["roofline"]["launches"] > 0 and two["roofline"]["gemm_tn_family"]["launches"] > 0 and len(two["roofline"]["by_kernel"]) >= 3
    one = _bench(["--data-ranks", "2"])
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 16 and one["config"]["global_batch"] == 16
    assert "row-sharded" in two["config"]["workload"]
    assert abs(two["config"]["final_loss"] - one["config"]["final_loss"]) < 3e-2, (two["config"]["final_loss"], one["config"]["final_loss"])


def test_bench_default_line_mixes_event_timed_and_overlapped_steps():
    """the default line (roofline on): step 0 of the timed region is event-timed with the towers one at a time, steps 1.. overlap them;
    the loss after the three optimizer steps must be the one of the run without events / without the stream switch"""
    with_ev = _bench(["--no-dense-text-line"], roofline=True)
    without = _bench(["--no-dense-text-line"])
    r = with_ev["roofline"]
    assert r["event_timed_steps"] == 1 and r["launches"] > 0 and r["achieved"] > 0 and "one tower at a time" in r["event_timed_steps_mode"]
    # (three optimizer steps at lr 1e-3 on the miniature model amplify the summation-order noise of the fp32 atomics: measured up to 4e-3)
    assert abs(with_ev["config"]["final_loss"] - without["config"]["final_loss"]) < 3e-2, (with_ev["config"]["final_loss"], without["config"]["final_loss"])
    assert with_ev["reserved_hbm_gb_rank0"] <= 1.5 * without["reserved_hbm_gb_rank0"] + 0.5  # one allocator pool per stream in BOTH kinds of step


def test_bench_accumulation_and_host_fed_modes():
    acc = _bench(["--accum-freq", "2"])
    assert acc["config"]["accum_freq"] == 2 and acc["config"]["global_batch"] == 16 and "accum_freq=2" in acc["config"]["workload"]
    import math
    assert math.isfinite(acc["config"]["final_loss"]) and acc["value"] > 0
    h2d = _bench(["--h2d"])
    assert h2d["config"]["input"] == "host_uint8_h2d" and math.isfinite(h2d["config"]["final_loss"]) and h2d["value"] > 0
    both = _bench(["--h2d", "--accum-freq", "2"])
    assert both["config"]["accum_freq"] == 2 and math.isfinite(both["config"]["final_loss"])


def test_bench_native_allreduce_paths():
    """bench.py --native-allreduce: (a) two ranks on the one GPU over gloo -- NativeGradSync instead of DistributedDataParallel -- report the
    loss of the DDP run and of one process on the concatenated batch; (b) one process with a ONE-rank RCCL communicator behind the C ABI
    (ocn_comm_allreduce_avg on every block's gradient arena, the loss collectives untouched) reproduces the plain one-process loss"""
    env = {"OCN_BENCH_ONE_DEVICE": "1"}
    ddp = _bench(["--dist-backend", "gloo"], nproc=2, env=env)
    nat = _bench(["--dist-backend", "gloo", "--native-allreduce"], nproc=2, env=env)
    assert nat["config"]["ddp"] is False and "native per-block" in nat["config"]["gradient_allreduce"] and ddp["config"]["ddp"] is True
    assert abs(nat["config"]["final_loss"] - ddp["config"]["final_loss"]) < 3e-2, (nat["config"]["final_loss"], ddp["config"]["final_loss"])
    plain = _bench([])
    one = _bench(["--native-allreduce"])
    assert "RCCL through the C ABI" in one["config"]["gradient_allreduce"]
    assert abs(one["config"]["final_loss"] - plain["config"]["final_loss"]) < 3e-2
    nc = _bench(["--native-comm"])  # one process, one-rank communicator: the loss's distributed (row-sharded) form with identity collectives
    assert "ocn_comm_" in nc["config"]["loss_collectives"] and abs(nc["config"]["final_loss"] - plain["config"]["final_loss"]) < 3e-2
    acc = _bench(["--native-allreduce", "--accum-freq", "2"])
    ref = _bench(["--accum-freq", "2"])
    assert abs(acc["config"]["final_loss"] - ref["config"]["final_loss"]) < 3e-2


def test_bench_world8_on_one_device():
    """bench.py's N = 8 launch -- the node size of BASELINE config 3 and of the driver's scaling run -- as eight ranks on the ONE GPU of the test
    box (gloo transport, OCN_BENCH_ONE_DEVICE=1; RCCL refuses more than one rank per device): local batch 4 per rank, the packed [B, 2E] feature
    all-gather + row-sharded global ClipLoss + reduce-scatter backward, the gradient all-reduce under DistributedDataParallel and under the native
    per-block all-reduce, the max-over-ranks timing and rank 0's single JSON line.  The loss after three optimizer steps must be the one of ONE
    process on the concatenation of the eight ranks' batches.  (Collective correctness only: unmeasured on multi-GPU hardware.)"""
    env = {"OCN_BENCH_ONE_DEVICE": "1"}
    small = ["--local-batch", "4"]
    one = _bench(small + ["--data-ranks", "8"])
    ddp = _bench(small + ["--dist-backend", "gloo"], nproc=8, env=env)
    nat = _bench(small + ["--dist-backend", "gloo", "--native-allreduce"], nproc=8, env=env)
    for rec in (ddp, nat):
        assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 32 and rec["config"]["parallelism"] == "dp8" and rec["scaling"] == "weak"
        assert "row-sharded" in rec["config"]["workload"]
        assert abs(rec["config"]["final_loss"] - one["config"]["final_loss"]) < 3e-2, (rec["config"]["final_loss"], one["config"]["final_loss"])
    assert one["config"]["global_batch"] == 32


def test_bench_gpus2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` exactly as the driver calls it (no torchrun, no RANK / WORLD_SIZE in the environment): bench.py must start
    the two ranks itself and the line must describe a two-rank run -- n_gpus, the process group's own world size and a rank -> device map
    (VERDICT r4 missing #2: this used to time ONE GPU and print n_gpus 1).  Two ranks on the one GPU of the test box over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OCN_BENCH_ONE_DEVICE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo"] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    two = json.loads(lines[0])
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["dist_world_size"] == 2 and cfg["global_batch"] == 16 and cfg["parallelism"] == "dp2"
    assert [d["rank"] for d in cfg["rank_devices"]] == [0, 1] and cfg["one_device_developer_mode"] is True
    launched = _bench(["--dist-backend", "gloo"], nproc=2, env={"OCN_BENCH_ONE_DEVICE": "1"})
    assert abs(two["config"]["final_loss"] - launched["config"]["final_loss"]) < 3e-2
    # and without the developer mode the same call must refuse to put two ranks on the one GPU of this box
    if __import__("torch").cuda.device_count() < 2:
        env.pop("OCN_BENCH_ONE_DEVICE")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode != 0 and "one rank per GPU is the contract" in r.stderr
