"""Host -> device input path of the hot loop: what ``TrainingTask.prepare_batch`` does
(reference ``src/open_clip/task/base_task.py:135-157``: ``tensor.to(device, non_blocking=True)`` per batch, images as the fp32
output of the transform) re-designed for the native path:

  * images travel as DECODED uint8 pixels ``[B, H, W, 3]`` (4x fewer PCIe bytes than fp32 ``[B, 3, H, W]``: 616 MB instead of
    2.47 GB per 4096-image batch); ToTensor + Normalize with the model's mean / std (``constants.py:1-2``) happen inside the
    patch-embedding kernel (``ocn_patchify_u8``), so no normalised image is ever materialised;
  * batches go pinned-host -> device with ``cudaMemcpyAsync`` on a dedicated COPY stream into one of ``depth`` device slots, while
    the compute stream works on the previous slot; the only coupling is two events per slot (copy done -> compute may read;
    compute done with the slot -> the copy stream may overwrite it).

    pipe = DeviceBatchPipeline(device, image_shape=(B, 224, 224, 3), text_shape=(B, 77))
    pipe.submit(u8_host, text_host)              # enqueue the copy of batch i+1 (returns immediately)
    batch = pipe.next()                          # device tensors of batch i, ordered after their copy on the current stream
    out = model(**batch)                         # NativeCLIP accepts the uint8 image directly
    pipe.release(batch)                          # any time after the forward is enqueued (records the slot's 'free' event): the backward
                                                 # reads nothing of the slot -- next() hands out per-step copies of the tokens / text layout

``prepare_batch`` keeps working on the result: it leaves integer tensors (uint8 pixels, int64 tokens) untouched.
PyTorch is plumbing here (pinned allocations, streams, events); there is no arithmetic in this file."""
import torch


class HostTextPlan:
    """The packed layout of one text batch (``model._TextPack``: only the tokens up to the pooled EOT exist in the text tower) computed on
    the HOST, where the tokens come from anyway (the tokenizer's output in the loader), so that the step never has to read the packed row
    count back from the device: index plumbing only (argmax / cumsum / a stable sort of B small integers), the tensors travel with the batch.
    ``device_views()`` are views of ONE int32 buffer + the packed token ids."""

    def __init__(self, text_host: torch.Tensor, vocab_size=None, buckets=True):
        B, L = text_host.shape
        eot = text_host.argmax(dim=-1)  # first maximum, as torch.argmax on the device (transformer.py:941-944)
        lens = eot + 1
        seq_off = torch.zeros(B + 1, dtype=torch.int64)
        seq_off[1:] = lens.cumsum(0)
        self.B, self.L, self.M = B, L, int(seq_off[-1])
        self.n_bad = int(((text_host < 0) | (text_host >= vocab_size)).sum()) if vocab_size is not None else 0
        keep = torch.arange(L)[None, :] < lens[:, None]
        self.tokens = text_host[keep]                                  # [M] int64
        posidx = torch.arange(L, dtype=torch.int32).expand(B, L)[keep]  # [M] int32
        nbk = (L + 31) // 32
        if buckets:
            nb = (lens + 31) // 32
            order = torch.sort(nb, stable=True).indices
            self.counts = torch.bincount(nb - 1, minlength=nbk).tolist()
        else:
            order, self.counts = torch.zeros(0, dtype=torch.int64), None
        # one int32 image: eot [B] | seq_off [B+1] | last_row [B] | order [B or 0] | posidx [M]
        self.ints = torch.cat([eot.int(), seq_off.int(), (seq_off[1:] - 1).int(), order.int(), posidx])
        self.has_order = bool(buckets)

    @staticmethod
    def capacity(B, L):
        return 4 * B + 1 + B * L

    def device_views(self, ints_dev, tokens_dev):
        B, M = self.B, self.M
        o = 0
        eot = ints_dev[o:o + B]; o += B
        seq_off = ints_dev[o:o + B + 1]; o += B + 1
        last_row = ints_dev[o:o + B]; o += B
        order = None
        if self.has_order:
            order = ints_dev[o:o + B]; o += B
        posidx = ints_dev[o:o + M]
        return {"eot": eot, "seq_off": seq_off, "last_row": last_row, "order": order, "posidx": posidx, "tokens": tokens_dev[:M],
                "M": M, "counts": self.counts, "n_bad": self.n_bad}


class DeviceBatchPipeline:
    def __init__(self, device, image_shape, text_shape, depth: int = 2, image_dtype=torch.uint8, plan_text_vocab=None, attn_buckets=True):
        """``plan_text_vocab`` = the model's vocab_size: every submitted batch also carries its packed text layout computed on the host
        (``HostTextPlan``), attached to the device text tensor; ``NativeCLIP`` then runs the step without any host synchronisation."""
        self.device = torch.device(device)
        self.depth = depth
        self.plan_text_vocab, self.attn_buckets = plan_text_vocab, attn_buckets
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = []
        for _ in range(depth):
            self.slots.append({
                "image": torch.empty(image_shape, dtype=image_dtype, device=self.device),
                "text": torch.empty(text_shape, dtype=torch.int64, device=self.device),
                "h_image": torch.empty(image_shape, dtype=image_dtype).pin_memory(),
                "h_text": torch.empty(text_shape, dtype=torch.int64).pin_memory(),
                "ready": torch.cuda.Event(), "free": torch.cuda.Event(), "host_done": None, "plan": None,
            })
            if plan_text_vocab is not None:
                cap = HostTextPlan.capacity(*text_shape)
                self.slots[-1].update(plan_ints=torch.empty(cap, dtype=torch.int32, device=self.device), h_plan_ints=torch.empty(cap, dtype=torch.int32).pin_memory(),
                                      plan_tokens=torch.empty(text_shape[0] * text_shape[1], dtype=torch.int64, device=self.device),
                                      h_plan_tokens=torch.empty(text_shape[0] * text_shape[1], dtype=torch.int64).pin_memory())
        self._w = 0      # next slot to fill
        self._r = 0      # next slot to hand out
        self._queued = 0

    def staging(self):
        """the pinned host buffers of the slot the next ``submit`` will use -- a loader can decode straight into them and
        then call ``submit()`` without arguments (no extra host copy)"""
        s = self.slots[self._w]
        if s["host_done"] is not None and not s["host_done"].query():
            s["host_done"].synchronize()  # the previous transfer out of this pinned buffer has finished
        return s["h_image"], s["h_text"]

    def submit(self, image_host=None, text_host=None):
        """enqueue host -> device of one batch on the copy stream.  ``image_host`` / ``text_host`` = any host tensors (copied into
        the slot's pinned staging buffers first) or None when the caller filled ``staging()`` in place."""
        if self._queued == self.depth:
            raise RuntimeError("DeviceBatchPipeline: all slots are in flight; call next() / release() first")
        s = self.slots[self._w]
        src_i, src_t = s["h_image"], s["h_text"]
        if image_host is not None:
            if image_host.is_pinned() and text_host.is_pinned():
                src_i, src_t = image_host, text_host  # already page-locked (DataLoader(pin_memory=True)): DMA straight out of it
            else:
                hi, ht = self.staging()
                hi.copy_(image_host)
                ht.copy_(text_host)
        plan = None
        if self.plan_text_vocab is not None:
            if s["host_done"] is not None and not s["host_done"].query():
                s["host_done"].synchronize()  # the slot's PREVIOUS transfer (``depth`` batches ago) still reads the pinned plan buffers: rare
            plan = HostTextPlan(src_t, self.plan_text_vocab, self.attn_buckets)
            s["h_plan_ints"][:plan.ints.numel()].copy_(plan.ints)
            s["h_plan_tokens"][:plan.M].copy_(plan.tokens)
        s["plan"] = plan
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(s["free"])  # the forward that read this slot last has consumed it
            s["image"].copy_(src_i, non_blocking=True)
            s["text"].copy_(src_t, non_blocking=True)
            if plan is not None:
                n = plan.ints.numel()
                s["plan_ints"][:n].copy_(s["h_plan_ints"][:n], non_blocking=True)
                s["plan_tokens"][:plan.M].copy_(s["h_plan_tokens"][:plan.M], non_blocking=True)
            s["ready"].record(self.copy_stream)
            s["host_done"] = s["ready"]
        self._w = (self._w + 1) % self.depth
        self._queued += 1

    def next(self):
        """device batch ``{'image': uint8 [B,H,W,3], 'text': int64 [B,L]}`` of the oldest submitted slot; the current stream is made
        to wait for its copy (no host synchronisation)"""
        if self._queued == 0:
            raise RuntimeError("DeviceBatchPipeline: nothing submitted")
        s = self.slots[self._r]
        torch.cuda.current_stream(self.device).wait_event(s["ready"])
        self._r = (self._r + 1) % self.depth
        self._queued -= 1
        # What the BACKWARD reads again -- the token ids (embedding backward) and, with a host plan, the whole addressing layout of the packed text
        # tower (seq_off / order / last_row / posidx / packed ids: varlen attention backward, embedding backward, every recomputed block under grad
        # checkpointing) -- is copied out of the slot here, on the compute stream, into tensors of this step (a few MB): the slot then holds nothing
        # that outlives the forward, and ``release()`` right after the forward can never let the copy stream overwrite addressing data under a
        # running backward.  Only the pixels are read in place (by the patch kernel, forward only).
        text = s["text"].clone()
        if s["plan"] is not None:  # rides on the tensor object (prepare_batch leaves a device int64 tensor as it is): model._TextPack picks it up
            plan = s["plan"]
            text._ocn_host_plan = plan.device_views(s["plan_ints"][:plan.ints.numel()].clone(), s["plan_tokens"][:plan.M].clone())
        return {"image": s["image"], "text": text, "_slot": s}

    def release(self, batch):
        """call once the forward that reads ``batch`` has been enqueued: the slot's pixels are read by the patch kernel only, and everything
        the backward reads again (token ids, the packed text layout) was copied out of the slot by ``next()``"""
        batch["_slot"]["free"].record(torch.cuda.current_stream(self.device))
