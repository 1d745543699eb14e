"""CPU baseline of the REFERENCE itself: ``open_clip_train.train.train_one_epoch`` (train.py:337) driving the reference's
own ``CLIP`` + ``CLIPTask`` + AdamW on the host cores -- ViT-B-32, fp32, batch 32, world_size 1, synthetic in-memory loader (workers 0)
-- timed by an outer wall clock per step AND read off the reference's own console line (train.py:456,496-502: ``samples_per_second =
step_batch_size * world_size / batch_time``), and, in the same process and thread count, the CPU oracle (``oracle/clip_oracle.py``, the port).
``bench.py``'s ``cpu_baseline`` leg runs this file as a subprocess on the GPU box's host cores (the reference comes out of
``oracle/_ref/reference_src.zip`` there: oracle/fetch_ref.py, oracle/ref_shim.py).  TEST / BENCH INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.ref_cpu_baseline [--steps 8] [--threads N] [--out FILE] [--no-port]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_shim import import_reference  # noqa: E402
from open_clip_amd.configs import get_model_config  # noqa: E402
from open_clip_amd.synth import synthetic_batch  # noqa: E402


class _Loader:
    """what train_one_epoch needs of data['train'] (train.py:372-375): set_epoch, .dataloader with num_batches / num_samples"""

    def __init__(self, batches, stamps):
        self.batches, self.stamps = batches, stamps
        self.num_batches, self.num_samples = len(batches), len(batches) * batches[0]["image"].shape[0]
        self.dataloader = self

    def set_epoch(self, e):
        pass

    def __iter__(self):
        for b in self.batches:
            self.stamps.append(time.perf_counter())  # the loop asks for batch i when step i-1 has finished
            yield b

    def __len__(self):
        return self.num_batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_cpu_train_one_epoch.json"))
    ap.add_argument("--no-port", action="store_true", help="skip the oracle port timed in the same process")
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    import logging
    import re
    from oracle.ref_shim import reference_available

    class _Lines(logging.Handler):  # the reference's own console lines ("... Batch (t): 1.312, 24.3902/s, 24.3902/s/gpu ...")
        def __init__(self):
            super().__init__(level=logging.INFO)
            self.rates = []

        def emit(self, record):
            m = re.search(r"Batch \(t\): [0-9.]+, ([0-9.eE+-]+)/s,", record.getMessage())
            if m:
                self.rates.append(float(m.group(1)))

    lines = _Lines()
    logging.getLogger().addHandler(lines)
    logging.getLogger().setLevel(logging.INFO)
    import_reference()
    import open_clip
    from open_clip.task import CLIPTask
    from open_clip_train.distributed import init_distributed_device
    from open_clip_train.params import parse_args
    from open_clip_train.train import TrainState, train_one_epoch
    from open_clip_train.optim import wd_param_groups

    bs = 32
    args = parse_args(["--model", "ViT-B-32", "--precision", "fp32", "--batch-size", str(bs), "--device", "cpu", "--lr", "5e-4", "--warmup", "2",
                       "--epochs", "1", "--log-every-n-steps", "1", "--skip-scheduler"])
    device = init_distributed_device(args)
    args.wandb = args.trackio = args.tensorboard = False
    args.distill = False
    torch.manual_seed(0)
    model = open_clip.create_model("ViT-B-32", output_dict=True)
    task = CLIPTask(model, rank=0, world_size=1, device=torch.device("cpu"), verbose=False)
    task.train()
    optimizer = torch.optim.AdamW(wd_param_groups(model, args.wd), lr=args.lr, betas=(args.beta1, args.beta2), eps=args.eps)  # optim.py:67-77
    cfg = get_model_config("ViT-B-32")
    batches = [synthetic_batch(cfg, bs, seed=1234 + i) for i in range(a.steps)]
    stamps = []
    loader = _Loader(batches, stamps)
    state = TrainState(task=task, optimizer=optimizer)
    t0 = time.perf_counter()
    train_one_epoch(state, {"train": loader}, args)
    t_end = time.perf_counter()
    stamps.append(t_end)
    steps = [stamps[i + 1] - stamps[i] for i in range(len(stamps) - 1)]
    warm = sorted(steps[2:]) if len(steps) > 3 else sorted(steps)
    med = warm[len(warm) // 2]
    logged = sorted(lines.rates[2:]) if len(lines.rates) > 3 else sorted(lines.rates)
    rec = {"what": "reference open_clip_train.train.train_one_epoch (train.py:337), ViT-B-32 fp32, batch 32, world_size 1, CPU, synthetic in-memory batches",
           "kind": "reference", "where": "build container (/root/reference)" if reference_available() else "this box's host cores (reference from oracle/_ref/reference_src.zip)",
           "nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(),
           "steps": a.steps, "sec_per_step_all": [round(s, 3) for s in steps], "sec_per_step_median_warm": round(med, 3), "pairs_per_s": round(bs / med, 2),
           "pairs_per_s_reference_log_line": (round(logged[len(logged) // 2], 2) if logged else None),  # median of the reference's own `N/s` (train.py:456)
           "torch": torch.__version__}
    if a.no_port:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)
        print(json.dumps(rec))
        return
    # the oracle (port) in the same process / thread count: the calibration between the two CPU baselines
    from oracle import clip_oracle as O
    from open_clip_amd.synth import init_state_dict
    st = init_state_dict(cfg, seed=0)
    plist = {k: torch.nn.Parameter(v.clone()) for k, v in st.items()}
    opt = torch.optim.AdamW(list(plist.values()), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
    ts = []
    for i in range(min(a.steps, 6)):
        t1 = time.perf_counter()
        outs, grads = O.train_forward_backward(batches[i]["image"], batches[i]["text"], {k: p.detach() for k, p in plist.items()}, cfg)
        for k, p in plist.items():
            p.grad = grads[k]
        opt.step()
        ts.append(time.perf_counter() - t1)
    w = sorted(ts[1:])
    rec["oracle_port_same_process"] = {"sec_per_step_median_warm": round(w[len(w) // 2], 3), "pairs_per_s": round(bs / w[len(w) // 2], 2)}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
