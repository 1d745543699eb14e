"""Gradient all-reduce of the data-parallel step without ``DistributedDataParallel`` (reference: the DDP wrap of
``src/open_clip/task/base_task.py:219-232``).

DDP's reducer copies every gradient into its own buckets before it can reduce them (605 MB per step for ViT-B-32: +1.6 ... +6.9 ms on
ONE GPU before any byte moves over xGMI).  The native backward already leaves the gradients where a collective wants them: each residual
block writes its twelve parameter gradients into ONE zeroed fp32 arena (``model._grad_arena``), and autograd hands those views to
``.grad`` without copying.  ``NativeGradSync`` therefore reduces IN PLACE:

  * a post-accumulate hook per parameter counts a group (one residual block; a tower's head; a tower's embeddings) down;
  * when a group is complete its ``.grad`` tensors are merged into flat address ranges (one per block when autograd kept the arena
    views; whatever they are otherwise -- correctness never depends on the aliasing) and each range is all-reduced (mean) on a
    communication stream that waits for the streams the backward runs on: the collectives of block *i* run under the backward of
    blocks *i-1 ...*, in reverse layer order, as DDP's buckets do; ranges below ``PACK_BELOW`` elements (LayerNorm vectors, the class /
    positional embeddings outside an arena) are packed into one staging buffer per group and share ONE collective;
  * ``finish()`` (call before ``optimizer.step()``) reduces what is left (0-d parameters such as ``logit_scale``; groups that did not
    complete because a parameter got no gradient) and makes the current stream wait for the communication stream.

Transport: ``comm`` = an ``open_clip_amd.comm.NativeComm`` (RCCL through the C ABI, ``ocn_comm_allreduce_avg``) or, without it, the
default ``torch.distributed`` process group (sum + scale; what the two-ranks-on-one-GPU / CPU gloo tests use).  Gradient averaging
(sum / world_size) and the initial rank-0 parameter broadcast follow DDP's semantics.

Unmeasured on more than one GPU (no multi-GPU node was available to the builder); on one GPU it costs the hooks only.
"""
import re
from contextlib import contextmanager

import torch

from . import ops

_BLOCK = re.compile(r"^(.*resblocks\.\d+)\.")
_HEAD = re.compile(r"^(visual\.(ln_post|proj|attn_pool)|ln_final|text_projection)\b")
PACK_BELOW = 1 << 16  # elements: flat ranges smaller than this share one staging buffer and one collective per group


def _group_key(name: str) -> str:
    """one group per residual block; per tower, the HEAD (final LayerNorm, projection: their gradients exist before the last block's backward
    starts, so their collective runs under the whole tower's backward) and the EMBEDDINGS (patch / token / positional / class embedding,
    ln_pre: complete only when the tower's backward ends)"""
    m = _BLOCK.match(name)
    if m:
        return m.group(1)
    tower = "visual" if name.startswith("visual.") else "text"
    return f"{tower}.head" if _HEAD.match(name) else f"{tower}.embed"


def flat_ranges(tensors):
    """merge tensors into maximal address-contiguous flat views: [(flat_tensor, [members])].  Contiguous fp32 tensors that sit back to
    back in ONE storage (the views of a block's gradient arena) become one range; everything else is a range of its own.  The ranges
    come out in the order of their first member in ``tensors`` -- NOT in address order, which differs from rank to rank while every
    rank has to issue the same collectives in the same order."""
    items = []
    for pos, t in enumerate(tensors):
        if t.is_contiguous() and t.dtype == torch.float32 and t.numel() > 0:
            items.append((t.untyped_storage().data_ptr(), t.data_ptr(), pos, t))
        else:
            items.append((None, id(t), pos, t))
    items.sort(key=lambda it: (it[0] is None, it[0] or 0, it[1]))
    out, i = [], 0
    while i < len(items):
        base, ptr, pos, t = items[i]
        if base is None:
            out.append((pos, t, [t]))
            i += 1
            continue
        members, first, end, j = [t], pos, ptr + t.numel() * 4, i + 1
        while j < len(items) and items[j][0] == base and items[j][1] == end:
            members.append(items[j][3])
            first = min(first, items[j][2])
            end += items[j][3].numel() * 4
            j += 1
        if len(members) == 1:
            out.append((first, t.view(-1), members))
        else:
            n = (end - ptr) // 4
            flat = torch.empty(0, dtype=torch.float32, device=t.device).set_(t.untyped_storage(), (ptr - base) // 4, (n,), (1,))
            out.append((first, flat, members))
        i = j
    out.sort(key=lambda r: r[0])
    return [(flat, members) for _, flat, members in out]


class NativeGradSync:
    def __init__(self, model, world_size: int, comm=None, process_group=None, broadcast_parameters: bool = True, find_unused_parameters: bool = False):
        """``find_unused_parameters`` (DDP's flag of the same name): a trainable parameter may receive no gradient on some or all ranks.  Every
        reduction then waits for ``finish()``, which first all-reduces a per-parameter "has a gradient" bitmap so that all ranks reduce the same
        tensors in the same order (a parameter used nowhere keeps ``grad is None``, one used on some ranks takes part with zeros elsewhere) -- no
        overlap with the backward.  Without it (default) every registered parameter MUST receive a gradient in every synchronised backward:
        ``finish()`` raises otherwise, naming the parameters, instead of letting ranks issue different collectives (a silent RCCL hang)."""
        self.model, self.world_size, self.comm, self.pg = model, int(world_size), comm, process_group
        ops.multi_gpu_defaults(self.world_size)  # the all-reduces' kernels hold CUs while the backward's GEMMs are launched: their rescue form
        self.find_unused = bool(find_unused_parameters)
        self.enabled = True
        self.stats = {"collectives": 0, "elements": 0, "ranges_per_group": [], "order": []}
        self._stream = None
        self._groups, self._of = {}, {}
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            key = _group_key(name) if p.dim() > 0 else "scalars"  # 0-d parameters are reduced together in finish()
            self._groups.setdefault(key, []).append(p)
            self._of[p] = key
        self._pending = {}
        self._fired = {}
        self._reduced_by_hook = {}
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self._of]
        if broadcast_parameters and self.world_size > 1:
            self.broadcast_parameters()

    # ---- transport -------------------------------------------------------------------------------------------------------
    def _allreduce_mean(self, flat):
        if self.world_size == 1 and self.comm is None:
            return
        if self.comm is not None:
            self.comm.all_reduce_avg(flat)
        else:
            import torch.distributed as dist
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
            flat.mul_(1.0 / self.world_size)
        self.stats["collectives"] += 1
        self.stats["elements"] += flat.numel()
        self.stats["order"] = (self.stats["order"] + [flat.numel()])[-256:]  # sizes of the last collectives, in issue order (must agree across ranks)

    def _allreduce_sum_small(self, t):
        if self.world_size == 1 and self.comm is None:
            return
        if self.comm is not None:
            self.comm.all_reduce_sum(t)
        else:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    def broadcast_parameters(self):
        """every rank starts from rank 0's parameters and buffers (DDP does this at construction, base_task.py:227)"""
        import torch.distributed as dist
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                if self.comm is not None:  # every dtype travels (NativeComm.broadcast sends non-float tensors as raw bytes), as under the process group
                    buf = t.data if t.is_contiguous() else t.data.contiguous()
                    self.comm.broadcast(buf, 0)
                    if buf.data_ptr() != t.data.data_ptr():
                        t.data.copy_(buf)
                else:
                    dist.broadcast(t.data, src=0, group=self.pg)
        if hasattr(self.model, "invalidate_weight_caches"):
            self.model.invalidate_weight_caches()

    # ---- streams ---------------------------------------------------------------------------------------------------------
    def _comm_stream(self, dev):
        if dev.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        return self._stream

    def _reduce(self, grads):
        dev = grads[0].device
        cs = self._comm_stream(dev)
        ranges = flat_ranges(grads)
        self.stats["ranges_per_group"] = (self.stats["ranges_per_group"] + [len(ranges)])[-64:]  # bounded: the last step's groups
        if cs is None:
            self._reduce_ranges(ranges)
            return
        cs.wait_stream(torch.cuda.current_stream(dev))      # the stream the hook (and the producing backward) runs on
        side = getattr(self.model, "_tower_side", {}).get(dev)
        if side is not None:
            cs.wait_stream(side)                            # the image tower's backward, when the towers run on two streams
        with torch.cuda.stream(cs):
            self._reduce_ranges(ranges)
            for _, members in ranges:
                for m in members:
                    m.record_stream(cs)

    def _reduce_ranges(self, ranges):
        """large ranges in place, one collective each; the small ones of the group through ONE packed fp32 buffer (which ranges are small
        depends on their sizes only, so every rank packs the same ones in the same order)"""
        if self.world_size == 1 and self.comm is None:
            return  # no transport configured: nothing to reduce, and no packing work either
        small = [flat for flat, _ in ranges if flat.numel() < PACK_BELOW and flat.dim() == 1]
        if len(small) < 2:
            small = []
        packed_ids = {id(f) for f in small}
        for flat, _ in ranges:
            if id(flat) not in packed_ids:
                self._allreduce_mean(flat)
        if small:
            packed = torch.cat([f.float() for f in small])
            self._allreduce_mean(packed)
            off = 0
            for f in small:
                f.copy_(packed[off:off + f.numel()])
                off += f.numel()

    # ---- hooks -----------------------------------------------------------------------------------------------------------
    def _hook(self, p):
        if not self.enabled:
            return
        key = self._of[p]
        fired = self._fired.setdefault(key, [])
        fired.append(p)
        if key != "scalars" and not self.find_unused and len(fired) == len(self._groups[key]):
            self._fired[key] = []
            self._reduced_by_hook[key] = True
            self._reduce([q.grad for q in self._groups[key]])  # registration order: the same on every rank

    @contextmanager
    def no_sync(self):
        """gradient accumulation (train.py:236-311): the micro-batches before the last one only accumulate into ``.grad``"""
        old, self.enabled = self.enabled, False
        try:
            yield
        finally:
            self.enabled = old

    def finish(self):
        """reduce what the hooks have not (0-d parameters packed into one tensor; incomplete groups), then order the current stream
        behind the communication stream.  Call once per optimizer step, after the last backward."""
        names = None
        left = []
        if self.find_unused:
            # one bitmap over ALL registered parameters (same length and order on every rank), summed: who has a gradient anywhere
            allp = [q for key in self._groups for q in self._groups[key]]
            have = torch.tensor([0.0 if q.grad is None else 1.0 for q in allp], device=allp[0].device)
            self._allreduce_sum_small(have)
            for q, n in zip(allp, have.tolist()):
                if n > 0:
                    if q.grad is None:
                        q.grad = torch.zeros_like(q)
                    left.append(q)
        else:
            solo = self.world_size == 1 and self.comm is None  # nobody to disagree with: an unused parameter is not an error
            for key in self._groups:  # registration order, not hook order
                if self._reduced_by_hook.get(key):
                    continue
                missing = [q for q in self._groups[key] if q.grad is None]
                if missing and solo:
                    left += [q for q in self._groups[key] if q.grad is not None]
                    continue
                if missing:  # the other ranks may have reduced this group already: fail loudly, here, instead of hanging in RCCL
                    names = names or {id(p): n for n, p in self.model.named_parameters()}
                    raise RuntimeError("NativeGradSync: no gradient reached " + ", ".join(names.get(id(q), "?") for q in missing[:8]) + " on this rank; "
                                       "every trainable parameter must take part in every synchronised backward (freeze it before construction, or "
                                       "construct NativeGradSync(find_unused_parameters=True), which reduces in finish() behind a has-gradient bitmap)")
                left += self._groups[key]
        self._fired = {}
        self._reduced_by_hook = {}
        scalars = [q for q in left if q.dim() == 0]
        rest = [q.grad for q in left if q.dim() > 0]
        if rest:
            self._reduce(rest)
        if scalars:
            packed = torch.stack([q.grad.reshape(()).float() for q in scalars])
            self._reduce([packed])
            dev = packed.device
            cs = self._comm_stream(dev)
            ctx = torch.cuda.stream(cs) if cs is not None else _null()
            with ctx:
                for i, q in enumerate(scalars):
                    q.grad.copy_(packed[i].to(q.grad.dtype))
        if self._stream is not None:
            torch.cuda.current_stream(self._stream.device).wait_stream(self._stream)

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


@contextmanager
def _null():
    yield
