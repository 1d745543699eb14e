"""CPU, world_size 2 over gloo: ``NativeGradSync`` (open_clip_amd/grad_sync.py -- per-block in-place gradient all-reduce from
post-accumulate hooks, the replacement for DistributedDataParallel's reducer, base_task.py:219-232) leaves in ``.grad`` exactly what
DDP leaves there: the mean over ranks, for arena-backed block gradients (one collective per block), separately allocated gradients,
0-d parameters, accumulation under ``no_sync()`` and parameters that get no gradient; and every rank starts from rank 0's weights."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class _ArenaBlockFn(torch.autograd.Function):
    """y = x @ W^T + b with BOTH parameter gradients written into one flat arena, as the native residual block does"""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        arena = torch.zeros(w.numel() + dy.shape[1])
        dw, db = arena[:w.numel()].view(w.shape), arena[w.numel():]
        dw += dy.t() @ x
        db += dy.sum(0)
        return dy @ w, dw, db


class _Block(nn.Module):
    def __init__(self, d, arena):
        super().__init__()
        self.fc = nn.Linear(d, d)
        self.arena = arena

    def forward(self, x):
        if self.arena:
            return torch.tanh(_ArenaBlockFn.apply(x, self.fc.weight, self.fc.bias))
        return torch.tanh(self.fc(x))


class _Toy(nn.Module):
    def __init__(self, d=6, with_unused=False):
        super().__init__()
        self.visual = nn.Module()
        self.visual.proj = nn.Parameter(torch.randn(d, d) * 0.3)
        if with_unused:
            self.visual.unused = nn.Parameter(torch.randn(3))  # never receives a gradient
        self.resblocks = nn.ModuleList([_Block(d, True), _Block(d, False), _Block(d, True)])
        self.logit_scale = nn.Parameter(torch.tensor(1.5))

    def forward(self, x):
        x = x @ self.visual.proj
        for b in self.resblocks:
            x = b(x)
        return (x * self.logit_scale.exp()).pow(2).mean()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from open_clip_amd.grad_sync import NativeGradSync
    torch.manual_seed(100 + rank)  # DIFFERENT initial weights per rank: both wrappers must start from rank 0's
    a, b = _Toy(), _Toy()
    b.load_state_dict(a.state_dict())
    g = torch.Generator().manual_seed(7 + rank)
    xs = [torch.randn(5, 6, generator=g) for _ in range(3)]
    ddp = nn.parallel.DistributedDataParallel(a)
    sync = NativeGradSync(b, world, process_group=dist.new_group())  # its own group: DDP keeps asynchronous collectives in flight on the default one
    same_start = all(torch.equal(p, q_) for p, q_ in zip(a.parameters(), b.parameters()))
    # step 1: plain; step 2: two accumulated micro-batches (no_sync on the first)
    ddp(xs[0]).backward()
    b(xs[0]).backward()
    sync.finish()
    g1 = [(n, p.grad.numpy().copy(), dict(b.named_parameters())[n].grad.numpy().copy()) for n, p in a.named_parameters() if p.grad is not None]
    ddp.zero_grad(set_to_none=True)
    b.zero_grad(set_to_none=True)
    with ddp.no_sync():
        ddp(xs[1]).backward()
    ddp(xs[2]).backward()
    with sync.no_sync():
        b(xs[1]).backward()
    b(xs[2]).backward()
    sync.finish()
    g2 = [(n, p.grad.numpy().copy(), dict(b.named_parameters())[n].grad.numpy().copy()) for n, p in a.named_parameters() if p.grad is not None]
    # a trainable parameter that never receives a gradient: find_unused_parameters=True on both sides leaves it None and reduces the rest
    # (behind a has-gradient bitmap, in finish()); the strict default raises instead of letting the ranks' collectives diverge
    sync.remove()
    torch.manual_seed(300 + rank)
    c, d = _Toy(with_unused=True), _Toy(with_unused=True)
    d.load_state_dict(c.state_dict())
    ddp_u = nn.parallel.DistributedDataParallel(c, find_unused_parameters=True)
    sync_u = NativeGradSync(d, world, process_group=dist.new_group(), find_unused_parameters=True)
    ddp_u(xs[0]).backward()
    d(xs[0]).backward()
    sync_u.finish()
    unused_ok = d.visual.unused.grad is None and c.visual.unused.grad is None
    for (n, p), (_, q_) in zip(c.named_parameters(), d.named_parameters()):
        if p.grad is not None:
            unused_ok = unused_ok and q_.grad is not None and bool(torch.allclose(p.grad, q_.grad, rtol=1e-5, atol=1e-7))
    # ... and the same under ACCUMULATION: two micro-batches, the first under no_sync(), then finish() -- the bitmap is taken over
    # the accumulated .grad, the unused parameter stays None, the rest equals DDP's (mean over ranks of the SUM over micro-batches)
    ddp_u.zero_grad(set_to_none=True)
    d.zero_grad(set_to_none=True)
    with ddp_u.no_sync():
        ddp_u(xs[1]).backward()
    ddp_u(xs[2]).backward()
    with sync_u.no_sync():
        d(xs[1]).backward()
    d(xs[2]).backward()
    sync_u.finish()
    unused_ok = unused_ok and d.visual.unused.grad is None and c.visual.unused.grad is None
    for (n, p), (_, q_) in zip(c.named_parameters(), d.named_parameters()):
        if p.grad is not None:
            unused_ok = unused_ok and q_.grad is not None and bool(torch.allclose(p.grad, q_.grad, rtol=1e-5, atol=1e-7))
    sync_u.remove()
    e_ = _Toy(with_unused=True)
    strict = NativeGradSync(e_, world, process_group=dist.new_group())
    e_(xs[0]).backward()
    try:
        strict.finish()
        unused_ok = False
    except RuntimeError as err:
        unused_ok = unused_ok and "visual.unused" in str(err)
    q.put((rank, same_start, unused_ok, g1, g2, dict(sync.stats)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29741), (8, 29751)])
def test_native_grad_sync_equals_ddp_over_gloo(world, port):
    """W = 8 is the node size of BASELINE config 3: the same checks on eight ranks, plus that every rank issued the SAME collectives (sizes) in the
    SAME order -- what an RCCL communicator needs from its ranks (unmeasured on hardware: the build container and the test box have one GPU)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
    for rank in range(world):
        same_start, unused_ok, g1, g2, stats = got[rank]
        assert same_start, "rank-0 parameter broadcast missing"
        assert unused_ok
        for step, gs in (("step1", g1), ("accumulated", g2)):
            for name, g_ddp, g_native in gs:
                assert np.allclose(g_native, g_ddp, rtol=1e-6, atol=1e-8), (rank, step, name)
        # the two arena-backed blocks went out as ONE flat range each, the module-autograd block as its two tensors
        assert sorted(stats["ranges_per_group"][:4]) == [1, 1, 1, 2], stats
        # ... and those two (36 + 6 elements, below PACK_BELOW) shared one packed collective: no rank ever sent the 6-element bias on its own
        assert 6 not in stats["order"] and 42 in stats["order"], stats["order"]
    # ranks agree with each other: gradients, and the order and sizes of the collectives they issued
    for r in range(1, world):
        for (n, _, a0), (_, _, a1) in zip(got[0][2], got[r][2]):
            assert np.array_equal(a0, a1), (r, n)
        assert got[r][4]["order"] == got[0][4]["order"] and len(got[0][4]["order"]) > 0, (r, got[r][4]["order"], got[0][4]["order"])


def test_flat_ranges_merges_only_adjacent_views_of_one_storage():
    from open_clip_amd.grad_sync import flat_ranges
    arena = torch.arange(20, dtype=torch.float32)
    a, b, c = arena[0:6].view(2, 3), arena[6:10], arena[12:20].view(2, 4)  # a|b adjacent, c behind a gap
    other = torch.ones(5)
    r = flat_ranges([c, other, b, a])
    sizes = sorted(f.numel() for f, _ in r)
    assert sizes == [5, 8, 10]
    flat = [f for f, m in r if f.numel() == 10][0]
    flat.mul_(2)  # in place through the flat view
    assert torch.equal(a.reshape(-1), torch.arange(6, dtype=torch.float32) * 2) and float(b[-1]) == 18.0 and float(arena[10]) == 10.0


def test_group_keys_split_head_and_embeddings():
    from open_clip_amd.grad_sync import _group_key
    assert _group_key("visual.transformer.resblocks.3.mlp.c_fc.weight") == "visual.transformer.resblocks.3"
    assert _group_key("transformer.resblocks.11.ln_1.bias") == "transformer.resblocks.11"
    for n in ("visual.ln_post.weight", "visual.proj"):
        assert _group_key(n) == "visual.head", n
    for n in ("visual.conv1.weight", "visual.class_embedding", "visual.positional_embedding", "visual.ln_pre.bias"):
        assert _group_key(n) == "visual.embed", n
    for n in ("ln_final.weight", "text_projection"):
        assert _group_key(n) == "text.head", n
    for n in ("token_embedding.weight", "positional_embedding"):
        assert _group_key(n) == "text.embed", n
