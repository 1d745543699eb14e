"""Multi-rank data-parallel step on REAL kernels: two ranks share the one MI355X of the test box (backend gloo, both
on cuda:0 -- RCCL refuses two ranks on one device), each runs the native model under DistributedDataParallel on its
half of a batch with NativeClipLoss in distributed mode, and the result is checked against the CPU oracle run
single-process on the full batch:
  * global loss (local_loss=False, gather_with_grad=False, BASELINE config 3): every rank's loss == full-batch loss;
    DDP-averaged tower gradients == full-batch gradients / W (logit_scale: the full gradient);
  * local_loss + gather_with_grad (the mode used at scale): mean of the rank losses == full-batch loss;
    DDP-averaged parameter gradients == full-batch gradients.
This exercises what the 8-GPU bench relies on (packed feature all-gather, reduce-scatter backward, DDP bucket hooks over
the block-granular autograd Functions, the bf16 gradient hand-off between blocks) with only the transport swapped."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WORLD, B_LOCAL = 2, 4
KEYS = ["visual.conv1.weight", "visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.1.mlp.c_fc.weight",
        "visual.transformer.resblocks.0.ln_1.weight", "transformer.resblocks.0.mlp.c_proj.weight", "transformer.resblocks.1.attn.out_proj.bias",
        "token_embedding.weight", "positional_embedding", "text_projection", "visual.proj", "logit_scale"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, mode, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.loss import NativeClipLoss, NativeSigLipLoss
    from open_clip_amd.model import NativeCLIP
    from open_clip_amd.synth import init_state_dict, synthetic_batch
    torch.cuda.set_device(0)
    cfg = get_model_config("small-test")
    siglip = mode == "siglip"
    state = init_state_dict(cfg, seed=3, perturb=True, siglip=siglip)
    batch = synthetic_batch(cfg, WORLD * B_LOCAL, seed=11)
    extra = dict(init_logit_scale=float(np.log(10)), init_logit_bias=-10.0) if siglip else {}
    model = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=True, **extra)
    model.load_state_dict(state)
    model = model.cuda().train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=1, gradient_as_bucket_view=True)
    if siglip:  # SigLIPTask's loss (siglip_task.py:35-43): every rank's images against every rank's texts, positives only locally
        loss_fn = NativeSigLipLoss(rank=rank, world_size=WORLD)
    else:
        kw = {"global": dict(local_loss=False, gather_with_grad=False),
              "global_rowsharded": dict(local_loss=False, gather_with_grad=False, row_sharded=True),
              "local_gwg": dict(local_loss=True, gather_with_grad=True)}[mode]
        loss_fn = NativeClipLoss(rank=rank, world_size=WORLD, **kw)
    lo, hi = rank * B_LOCAL, (rank + 1) * B_LOCAL
    losses = []
    for _ in range(2):  # two passes: the second one runs with DDP's rebuilt buckets and re-used gradient views
        net.zero_grad(set_to_none=True)
        out = net(image=batch["image"][lo:hi].cuda(), text=batch["text"][lo:hi].cuda())
        loss = loss_fn(**out)
        loss.backward()
        torch.cuda.synchronize()
        losses.append(float(loss.detach()))
    grads = {k: p.grad.detach().float().cpu().numpy() for k, p in model.named_parameters() if k in KEYS or k == "logit_bias"}
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), loss=np.array(losses), **grads)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["global", "global_rowsharded", "local_gwg", "siglip"])
def test_two_ranks_ddp_against_full_batch_oracle(mode, tmp_path):
    import torch.multiprocessing as mp
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.synth import init_state_dict, synthetic_batch
    from oracle import clip_oracle as O
    port = _free_port()
    mp.spawn(_worker, args=(port, mode, str(tmp_path)), nprocs=WORLD, join=True)
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=3, perturb=True, siglip=(mode == "siglip"))
    batch = synthetic_batch(cfg, WORLD * B_LOCAL, seed=11)
    ref, rgrads = O.train_forward_backward(batch["image"], batch["text"], state, cfg, siglip=(mode == "siglip"))
    r = [np.load(os.path.join(str(tmp_path), f"rank{i}.npz")) for i in range(WORLD)]
    for i in range(WORLD):
        assert abs(r[i]["loss"][0] - r[i]["loss"][1]) < 1e-5, "the two passes must agree (same weights, same data)"
    if mode.startswith("global"):  # the row-sharded evaluation must be indistinguishable from the redundant one
        for i in range(WORLD):
            assert abs(float(r[i]["loss"][1]) - float(ref["loss"])) < 2e-2, (r[i]["loss"], float(ref["loss"]))
        gscale = 1.0 / WORLD
    else:
        # local_loss + gather_with_grad, and SigLIP (loss.py:406-489: a rank's loss = its B images against all N texts, divided by
        # B, NOT averaged over ranks): the MEAN of the rank losses is the single-process full-batch loss, and so is the DDP mean
        # of the gradients (logit_scale / logit_bias included)
        assert abs(np.mean([float(x["loss"][1]) for x in r]) - float(ref["loss"])) < 2e-2
        gscale = 1.0
    for k in KEYS + (["logit_bias"] if mode == "siglip" else []):
        g0, g1 = torch.from_numpy(r[0][k]), torch.from_numpy(r[1][k])
        assert torch.equal(g0, g1), f"{k}: ranks disagree after the gradient all-reduce"
        # logit_scale: in the global mode every rank evaluates the FULL loss, so each holds the full d/ds and the DDP mean
        # leaves it unscaled (only the feature gradients are restricted to the local slice, loss.py:47-50)
        want = rgrads[k] * (1.0 if (k == "logit_scale" and mode.startswith("global")) else gscale)
        if mode == "global_rowsharded" and k == "logit_scale":
            pass  # every rank holds the all-reduced (full) d/ds, exactly as in the redundant evaluation
        rel = float((g0 - want).norm() / want.norm().clamp_min(1e-12))
        assert rel < 6e-2, (mode, k, rel)


def _nccl_one_rank_worker(rank, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from open_clip_amd.loss import _reduce_scatter_sum
    g = torch.Generator().manual_seed(5)
    inp = torch.randn(6, 10, generator=g).cuda()
    out = torch.empty(6, 10, device="cuda")
    _reduce_scatter_sum(out, inp)  # the RCCL branch (dist.reduce_scatter_tensor): the one line the gloo runs cannot execute
    packed = torch.randn(6, 8, generator=g).cuda()
    allp = torch.empty(6, 8, device="cuda")
    dist.all_gather_into_tensor(allp, packed)  # the packed feature all-gather of the loss forward, same process group
    torch.cuda.synchronize()
    ok = bool(torch.equal(out, inp)) and bool(torch.equal(allp, packed))
    open(os.path.join(outdir, "ok.txt"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_rccl_process_group_runs_the_loss_collectives(tmp_path):
    """A one-rank RCCL ("nccl") process group on the test box's GPU: the collectives the distributed loss issues
    (all_gather_into_tensor forward, reduce_scatter_tensor backward) run through RCCL itself -- with one rank they must be
    identities.  (Two ranks cannot share a device under RCCL; the multi-rank semantics are covered over gloo above.)"""
    import torch.multiprocessing as mp
    mp.spawn(_nccl_one_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(os.path.join(str(tmp_path), "ok.txt")).read() == "1"


def test_native_comm_one_rank_collectives_through_the_c_abi():
    """ocn_comm_* (RCCL bound at run time behind the C ABI) with a ONE-rank communicator on the test box's GPU: all-gather,
    reduce-scatter and all-reduce must be identities, in fp32 and bf16 (more ranks need more GPUs: RCCL refuses two per device)"""
    from open_clip_amd.comm import NativeComm
    torch.cuda.set_device(0)
    comm = NativeComm(NativeComm.make_unique_id(), 0, 1)
    g = torch.Generator().manual_seed(9)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(64, 48, generator=g).to(dt).cuda()
        out = torch.empty_like(x)
        comm.all_gather_into_tensor(out, x)
        rs = torch.empty_like(x)
        comm.reduce_scatter_sum(rs, x)
        ar = x.clone()
        comm.all_reduce_sum(ar)
        torch.cuda.synchronize()
        assert torch.equal(out, x) and torch.equal(rs, x) and torch.equal(ar, x), dt
        # one neighbour exchange (ocn_comm_sendrecv; loss.py:226-243): with one rank the neighbour on both sides is the rank itself
        got = torch.full_like(x, float("nan"))
        comm.sendrecv(x, 0, got, 0)
        torch.cuda.synchronize()
        assert torch.equal(got, x), dt
    ids = torch.arange(100, dtype=torch.int64).cuda()
    got = torch.zeros_like(ids)
    comm.sendrecv(ids, 0, got, 0)  # any other dtype travels as raw bytes
    torch.cuda.synchronize()
    assert torch.equal(got, ids)
    assert comm.count() == (1, 0)  # what RCCL itself reports (ncclCommCount / ncclCommUserRank): bench.py's `rccl_ranks`
    comm.close()


def test_native_comm_one_rank_at_config3_payloads_and_the_loss_through_it():
    """the argument marshalling of ocn_comm_* at BASELINE config 3's sizes (RCCL refuses more than one rank per device, so ONE rank:
    every collective is an identity whose counts, dtypes and pointers still travel the product's path): the packed [4096, 2 x 512] fp32 feature
    all-gather (8 MiB per rank), the [4096, 1024] reduce-scatter of the row-sharded loss, the 3-element scalar all-reduce, a 28 MB gradient-arena
    all-reduce (mean), the parameter broadcast in fp32 / bf16 / raw bytes (int64, bool) -- and NativeClipLoss(comm=...) itself with world_size 1
    forced through its world_size > 1 code (row-sharded: all-gather -> two [4096 x 4096] fused logits + CE -> reduce-scatter -> all-reduce) against
    the plain world_size-1 loss.  Unmeasured on more than one GPU."""
    from open_clip_amd import loss as L
    from open_clip_amd.comm import NativeComm
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    comm = NativeComm(NativeComm.make_unique_id(), 0, 1)
    g = torch.Generator(device=dev).manual_seed(21)
    packed = torch.randn(4096, 1024, device=dev, generator=g)
    out = torch.full_like(packed, float("nan"))
    comm.all_gather_into_tensor(out, packed)
    rs = torch.full_like(packed, float("nan"))
    comm.reduce_scatter_sum(rs, packed)
    acc = torch.randn(3, device=dev, generator=g)
    acc0 = acc.clone()
    comm.all_reduce_sum(acc)
    arena = torch.randn(7_087_872, device=dev, generator=g)  # one ViT-B-32 image block's gradient arena
    arena0 = arena.clone()
    comm.all_reduce_avg(arena)
    torch.cuda.synchronize()
    assert torch.equal(out, packed) and torch.equal(rs, packed) and torch.equal(acc, acc0) and torch.equal(arena, arena0)
    for t in (torch.randn(768, 3072, device=dev, generator=g), torch.randn(513, device=dev, generator=g).bfloat16(),
              torch.arange(77, device=dev), torch.tensor([True, False, True], device=dev), torch.randn(5, device=dev, generator=g).half()):
        t0 = t.clone()
        comm.broadcast(t, 0)
        torch.cuda.synchronize()
        assert torch.equal(t, t0), t.dtype
    # the loss's multi-rank branch over the communicator: with W = 1 it must reproduce the one-process loss (same kernels, same order)
    I = torch.nn.functional.normalize(torch.randn(4096, 512, device=dev, generator=g), dim=-1)
    T = torch.nn.functional.normalize(0.5 * I + torch.nn.functional.normalize(torch.randn(4096, 512, device=dev, generator=g), dim=-1), dim=-1)
    res = []
    for force in (False, True):
        Ii, Ti, s = I.clone().requires_grad_(True), T.clone().requires_grad_(True), torch.tensor(14.28, device=dev, requires_grad=True)
        if force:  # world_size 1 through the world_size > 1 code (a communicator is given): row-sharded, every collective through ocn_comm_*
            loss = L.NativeClipLoss(rank=0, world_size=1, row_sharded=True, comm=comm)(Ii, Ti, s)
        else:
            loss = L.NativeClipLoss()(Ii, Ti, s)
        loss.backward()
        res.append((float(loss), Ii.grad.clone(), Ti.grad.clone(), float(s.grad)))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 and abs(res[0][3] - res[1][3]) <= 1e-5 * abs(res[0][3]) + 1e-7
    for a, b in ((res[0][1], res[1][1]), (res[0][2], res[1][2])):
        assert float((a - b).norm() / a.norm()) <= 1e-4  # identical kernels; the two forms add the row / column parts in a different order
    comm.close()


def _sync_twin_worker(rank, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from open_clip_amd.configs import get_model_config
    from open_clip_amd.grad_sync import NativeGradSync
    from open_clip_amd.loss import NativeClipLoss
    from open_clip_amd.model import NativeCLIP
    from open_clip_amd.synth import init_state_dict, synthetic_batch
    torch.cuda.set_device(0)
    cfg = get_model_config("small-test")
    state = init_state_dict(cfg, seed=3, perturb=True)
    batch = synthetic_batch(cfg, WORLD * B_LOCAL, seed=11)
    lo, hi = rank * B_LOCAL, (rank + 1) * B_LOCAL
    loss_fn = NativeClipLoss(rank=rank, world_size=WORLD, local_loss=True, gather_with_grad=True)
    res = {}
    for kind in ("ddp", "native"):
        model = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=True)
        model.load_state_dict(state)
        model = model.cuda().train()
        if kind == "ddp":
            net, sync = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=1, gradient_as_bucket_view=True), None
        else:
            net, sync = model, NativeGradSync(model, WORLD, process_group=dist.new_group())
        for _ in range(2):  # the second pass re-uses the .grad tensors' storage decisions of the first
            net.zero_grad(set_to_none=True)
            out = net(image=batch["image"][lo:hi].cuda(), text=batch["text"][lo:hi].cuda())
            loss_fn(**out).backward()
            if sync is not None:
                sync.finish()
            torch.cuda.synchronize()
        res[kind] = {k: p.grad.detach().float().cpu().numpy() for k, p in model.named_parameters()}
        if sync is not None:
            res["ranges"] = np.array(sync.stats["ranges_per_group"])
            res["collectives"] = np.array([sync.stats["collectives"]])
    np.savez(os.path.join(outdir, f"twin{rank}.npz"), ranges=res["ranges"], collectives=res["collectives"],
             **{"ddp/" + k: v for k, v in res["ddp"].items()}, **{"native/" + k: v for k, v in res["native"].items()})
    dist.barrier()
    dist.destroy_process_group()


def test_native_grad_sync_equals_ddp_on_real_kernels(tmp_path):
    """two ranks on the one GPU (gloo transport): the per-block in-place gradient all-reduce of open_clip_amd/grad_sync.py leaves in .grad
    what DistributedDataParallel leaves there, for EVERY parameter (rel 2e-4: two separate backward passes differ by the summation order
    of their fp32 atomics), identically on both ranks -- and every residual block went out as ONE flat range (autograd kept the views of
    the block's gradient arena: no copy into buckets)"""
    import torch.multiprocessing as mp
    mp.spawn(_sync_twin_worker, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    r = [np.load(os.path.join(str(tmp_path), f"twin{i}.npz")) for i in range(WORLD)]
    names = [k[len("ddp/"):] for k in r[0].files if k.startswith("ddp/")]
    assert len(names) > 50
    for k in names:
        a, b = r[0]["native/" + k], r[1]["native/" + k]
        assert np.array_equal(a, b), f"{k}: ranks disagree after the native all-reduce"
        d = r[0]["ddp/" + k]
        denom = max(float(np.linalg.norm(d)), 1e-12)
        assert float(np.linalg.norm(a - d)) / denom < 2e-4, (k, float(np.linalg.norm(a - d)) / denom)
    # small-test has 2 + 2 residual blocks: per pass 4 block groups of one range each, plus the embedding / head leftovers
    ranges = r[0]["ranges"].tolist()
    assert ranges.count(1) >= 8, ranges
