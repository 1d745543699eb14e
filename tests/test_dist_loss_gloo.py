"""CPU, world_size 2, 3 and 8 over gloo: the collective plumbing of NativeClipLoss / NativeSigLipLoss
(packed all-gather, which operands carry gradient per mode, reduce-scatter backward) reproduces the per-rank
losses and gradients the REFERENCE produced (tests/golden/dist_loss_w*.npz).

The HIP compute seam (`open_clip_amd.loss.PairTerm`) is replaced by an fp32 torch stand-in *inside this test* --
the product has no such path; the kernels themselves are checked on the GPU (tests/test_kernels_gpu.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden_util import load


class CpuPairTerm:
    """test double with the interface of open_clip_amd.loss._PairTerm"""

    def __init__(self, X, Y, scale):
        self.X, self.Y, self.s = X, Y, scale  # scale: 1-element tensor, as in the product (never read on the host)
        self.R, self.N, self.E = X.shape[0], Y.shape[0], X.shape[1]

    def compute_logits(self, bias=None):
        self.bias = 0.0 if bias is None else bias.detach()
        self.logits = (self.s * self.X) @ self.Y.t() + self.bias
        return self

    def softmax_ce(self, label_offset, loss_scale, grad_scale, acc):
        p = torch.softmax(self.logits, -1)
        lab = torch.arange(self.R) + label_offset
        lse = torch.logsumexp(self.logits, -1)
        acc[0] += ((lse - self.logits[torch.arange(self.R), lab]) * loss_scale).sum()
        onehot = torch.zeros_like(p)
        onehot[torch.arange(self.R), lab] = 1
        self.G = (p - onehot) * grad_scale
        acc[1] += (self.G * self.logits).sum()

    def siglip(self, label_offset, negative_only, loss_scale, grad_scale, acc):
        lab = -torch.ones_like(self.logits)
        if not negative_only:
            lab[torch.arange(self.R), torch.arange(self.R) + label_offset] = 1
        z = lab * self.logits
        acc[0] += (-torch.nn.functional.logsigmoid(z)).sum() * loss_scale
        self.G = -lab * torch.sigmoid(-z) * grad_scale
        acc[1] += (self.G * (self.logits - self.bias)).sum()  # the bias subtracted per element, as ocn_siglip_rows does
        acc[2] += self.G.sum()

    def dX(self):
        return self.s * self.G @ self.Y

    def dY(self, into=None):
        out = self.G.t() @ (self.s * self.X)
        if into is not None:
            into.copy_(out)
            return into
        return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import open_clip_amd.loss as L
    L.PairTerm = CpuPairTerm
    g = load(f"dist_loss_w{world}.npz")
    feats = torch.from_numpy(g["feats"])
    res = {}
    for mode, kw in (("global", dict(local_loss=False, gather_with_grad=False)),
                     ("local_gwg", dict(local_loss=True, gather_with_grad=True)),
                     ("local_nograd", dict(local_loss=True, gather_with_grad=False)),
                     ("global_gwg", dict(local_loss=False, gather_with_grad=True)),
                     ("global_rowsharded", dict(local_loss=False, gather_with_grad=False, row_sharded=True))):
        img = feats[rank, 0].clone().requires_grad_(True)
        txt = feats[rank, 1].clone().requires_grad_(True)
        s = torch.tensor(float(g["scale"]), requires_grad=True)
        loss = L.NativeClipLoss(rank=rank, world_size=world, **kw)(img, txt, s)
        loss.backward()
        res[mode] = (float(loss), img.grad.numpy(), txt.grad.numpy(), float(s.grad))
    img = feats[rank, 0].clone().requires_grad_(True)
    txt = feats[rank, 1].clone().requires_grad_(True)
    s = torch.tensor(float(g["scale"]), requires_grad=True)
    b = torch.tensor(float(g["bias"]), requires_grad=True)
    # the reference's four transports of the same sum (loss.py:410-477) are one implementation here: every name must give the 'bidir' vectors
    loss = L.NativeSigLipLoss(rank=rank, world_size=world, dist_impl={2: "shift", 3: "reduce", 8: "gather"}.get(world))(img, txt, s, b)
    loss.backward()
    res["siglip"] = (float(loss), img.grad.numpy(), txt.grad.numpy(), float(s.grad), float(b.grad))
    # SigLipLoss(chunk_size=2) (loss.py:369-404) and ClipLoss with a logit_bias (loss.py:111-113: a real, exactly zero gradient)
    img = feats[rank, 0].clone().requires_grad_(True)
    txt = feats[rank, 1].clone().requires_grad_(True)
    s = torch.tensor(float(g["scale"]), requires_grad=True)
    b = torch.tensor(float(g["bias"]), requires_grad=True)
    loss = L.NativeSigLipLoss(rank=rank, world_size=world, chunk_size=2)(img, txt, s, b)
    loss.backward()
    res["siglip_chunked"] = (float(loss), img.grad.numpy(), txt.grad.numpy(), float(s.grad), float(b.grad))
    img = feats[rank, 0].clone().requires_grad_(True)
    txt = feats[rank, 1].clone().requires_grad_(True)
    s = torch.tensor(float(g["scale"]), requires_grad=True)
    b = torch.tensor(float(g["bias"]), requires_grad=True)
    loss = L.NativeClipLoss(rank=rank, world_size=world, local_loss=True, gather_with_grad=True)(img, txt, s, b)
    loss.backward()
    res["clip_bias"] = (float(loss), img.grad.numpy(), None if b.grad is None else float(b.grad))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29721), (3, 29722), (8, 29723)])
def test_native_losses_reproduce_reference_collective_semantics(world, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    g = load(f"dist_loss_w{world}.npz")
    for rank in range(world):
        for mode in ("global", "local_gwg", "local_nograd", "global_gwg"):
            loss, di, dt, ds = got[rank][mode]
            pre = f"r{rank}/clip/{mode}/"
            assert abs(loss - float(g[pre + "loss"])) < 1e-5, (rank, mode)
            np.testing.assert_allclose(di, g[pre + "dimg"], atol=2e-6, err_msg=pre)
            np.testing.assert_allclose(dt, g[pre + "dtxt"], atol=2e-6, err_msg=pre)
            assert abs(ds - float(g[pre + "dscale"])) < 1e-5
        # the row-sharded evaluation of the global loss must give the reference's "global" numbers (value and local grads)
        loss, di, dt, ds = got[rank]["global_rowsharded"]
        pre = f"r{rank}/clip/global/"
        assert abs(loss - float(g[pre + "loss"])) < 1e-5, rank
        np.testing.assert_allclose(di, g[pre + "dimg"], atol=2e-6, err_msg="rowsharded " + pre)
        np.testing.assert_allclose(dt, g[pre + "dtxt"], atol=2e-6, err_msg="rowsharded " + pre)
        assert abs(ds - float(g[pre + "dscale"])) < 1e-5
        for key, pre in (("siglip", f"r{rank}/siglip/bidir/"), ("siglip_chunked", f"r{rank}/siglip/chunked/")):
            loss, di, dt, ds, db = got[rank][key]
            assert abs(loss - float(g[pre + "loss"])) < 1e-5, pre
            np.testing.assert_allclose(di, g[pre + "dimg"], atol=2e-6, err_msg=pre)
            np.testing.assert_allclose(dt, g[pre + "dtxt"], atol=2e-6, err_msg=pre)
            assert abs(ds - float(g[pre + "dscale"])) < 2e-5 and abs(db - float(g[pre + "dbias"])) < 2e-5, pre
        loss, di, db = got[rank]["clip_bias"]
        pre = f"r{rank}/clip/local_gwg_bias/"
        assert abs(loss - float(g[pre + "loss"])) < 1e-5
        np.testing.assert_allclose(di, g[pre + "dimg"], atol=2e-6, err_msg=pre)
        assert db is not None and abs(db - float(g[pre + "dbias"])) < 1e-6, "ClipLoss must hand logit_bias a (zero) gradient, as the reference does"
