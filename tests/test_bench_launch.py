"""How bench.py gets its N ranks (`python bench.py --gpus 8` without a launcher used to time ONE GPU and print n_gpus 1).
The reference's launch contract is torchrun (open_clip_train/distributed.py:80-166: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment); the driver calls plain `python bench.py --gpus N`.  bench.py accepts both and nothing else: without a launcher it starts
the N ranks itself, under a launcher WORLD_SIZE must equal N, and fewer visible GPUs than ranks is an error.  CPU only: the GPU twin
(two self-launched ranks on one device over gloo) is tests/test_bench_gpu.py::test_bench_gpus2_without_a_launcher_starts_two_ranks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_check_launch_modes():
    assert bench.check_launch(1, {}) == "single"
    assert bench.check_launch(1, {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}) == "single"
    assert bench.check_launch(8, {}) == "spawn"
    assert bench.check_launch(8, {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"}) == "rank"
    for gpus, env in ((8, {"RANK": "0", "WORLD_SIZE": "1"}), (8, {"RANK": "0", "WORLD_SIZE": "2"}), (1, {"RANK": "0", "WORLD_SIZE": "8"}),
                      (8, {"WORLD_SIZE": "8"}), (1, {"WORLD_SIZE": "2"})):
        with pytest.raises(SystemExit):
            bench.check_launch(gpus, env)


def test_launcher_command_is_one_rank_per_gpu_on_this_node():
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    c = bench.launcher_command(2, [])
    p1, p2 = c[c.index("--master-port") + 1], bench.launcher_command(2, [])[c.index("--master-port") + 1]
    assert 1024 < int(p1) < 65536 and 1024 < int(p2) < 65536  # a free port is looked up for every launch


def test_gpus2_without_enough_devices_fails_loudly_and_hands_the_exit_code_on():
    """no GPU in this container: `python bench.py --gpus 2` must start two ranks (the ranks' own message shows it did), both must refuse
    to share what is not there, and the parent must hand the failure on -- not print a one-GPU line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OCN_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "one rank per GPU is the contract" in r.stderr and "--gpus 2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_wrong_world_size_is_an_error_before_any_device_is_touched():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29556")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 8" in r.stderr


def _facts_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = bench.dist_facts(world, torch.device("cpu"), None, 1.0 + 0.01 * rank, "gloo")
    q.put((rank, f))
    dist.barrier()
    dist.destroy_process_group()


def test_transport_facts_of_an_eight_rank_run_agree():
    """what `bench.py --gpus 8` puts on its line about the transports -- here 8 gloo ranks on the host: the process group's world
    size, the ranks counted by a collective on the data path and the global batch derived from them agree on every rank; `rccl_ranks` is the RCCL
    communicator's own count (ocn_comm_count) and says so when the ranks do not talk RCCL; the per-rank times of the timed region ride along"""
    import torch.multiprocessing as mp
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_facts_worker, args=(r, world, 29731, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert len(got) == world
    for rank, f in got.items():
        assert f["dist_world_size"] == f["transport_ranks"] == world
        assert f["rccl_ranks"] is None and "gloo" in f["rccl_ranks_is"]
        assert 4096 * f["transport_ranks"] == 32768  # config.global_batch = local_batch * accum_freq * world: the metric's gbs at 8 ranks
        assert abs(f["elapsed_s_per_rank_min"] - 1.0) < 1e-6 and abs(f["elapsed_s_per_rank_max"] - 1.07) < 1e-6
    assert len({tuple(sorted(f.items())) for f in got.values()}) == 1  # every rank holds the same record
