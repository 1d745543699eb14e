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
    pipe.release(batch)                          # after the forward has consumed the pixels (records the slot's 'free' event)

``prepare_batch`` keeps working on the result: it leaves integer tensors (uint8 pixels, int64 tokens) untouched.
PyTorch is plumbing here (pinned allocations, streams, events); there is no arithmetic in this file."""
import torch


class DeviceBatchPipeline:
    def __init__(self, device, image_shape, text_shape, depth: int = 2, image_dtype=torch.uint8):
        self.device = torch.device(device)
        self.depth = depth
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = []
        for _ in range(depth):
            self.slots.append({
                "image": torch.empty(image_shape, dtype=image_dtype, device=self.device),
                "text": torch.empty(text_shape, dtype=torch.int64, device=self.device),
                "h_image": torch.empty(image_shape, dtype=image_dtype).pin_memory(),
                "h_text": torch.empty(text_shape, dtype=torch.int64).pin_memory(),
                "ready": torch.cuda.Event(), "free": torch.cuda.Event(), "host_done": None,
            })
        self._w = 0      # next slot to fill
        self._r = 0      # next slot to hand out
        self._queued = 0

    def staging(self):
        """the pinned host buffers of the slot the next ``submit`` will use -- a loader can decode straight into them and
        then call ``submit()`` without arguments (no extra host copy)"""
        s = self.slots[self._w]
        if s["host_done"] is not None:
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
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(s["free"])  # the forward that read this slot last has consumed it
            s["image"].copy_(src_i, non_blocking=True)
            s["text"].copy_(src_t, non_blocking=True)
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
        return {"image": s["image"], "text": s["text"], "_slot": s}

    def release(self, batch):
        """call once the kernels that read ``batch`` have been enqueued (after the forward; the image is only read by the patch
        kernel, the tokens by the embedding / argmax kernels and again by the embedding backward -- release after backward when
        the text tower trains)"""
        batch["_slot"]["free"].record(torch.cuda.current_stream(self.device))
