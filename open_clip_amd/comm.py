"""RCCL communicator behind the C ABI (``ocn_comm_*``, include/openclip_hip.h): the collectives of ``gather_features`` (reference
``src/open_clip/loss.py:23-54``) issued directly on the compute stream, without a process-group object in between.

    comm = NativeComm.from_process_group(rank, world_size)      # the 128-byte id travels through torch.distributed once (the reference
    loss = NativeClipLoss(rank=rank, world_size=world_size, comm=comm, ...)   # broadcasts such things the same way, distributed.py:186-193)

Without ``comm=`` the losses use ``torch.distributed`` (backend "nccl" = RCCL underneath) -- that is what the multi-rank tests and the
bench exercise (two ranks on one GPU need gloo; RCCL refuses two ranks per device).  The native communicator has been run on
hardware with ONE rank only (tests/test_ddp_gpu.py); ``bench.py --native`` selects it (with the native gradient all-reduce) and falls back to
torch.distributed + DDP when the communicator cannot be created."""
import ctypes

import torch

from . import _lib

F32, BF16 = torch.float32, torch.bfloat16


def _dt(t):
    if t.dtype == F32:
        return 0
    if t.dtype == BF16:
        return 1
    raise RuntimeError(f"NativeComm: dtype {t.dtype} not supported (fp32 / bf16)")


def _check(t, name):
    if not (t.is_cuda and t.is_contiguous()):
        raise RuntimeError(f"NativeComm: '{name}' must be a contiguous device tensor")
    return t.data_ptr()


class NativeComm:
    def __init__(self, unique_id: bytes, rank: int, world_size: int):
        assert len(unique_id) == 128
        self.rank, self.world_size = rank, world_size
        buf = ctypes.create_string_buffer(unique_id, 128)
        out = ctypes.c_void_p()
        _lib.call("ocn_comm_init", ctypes.cast(buf, ctypes.c_void_p), rank, world_size, ctypes.cast(ctypes.byref(out), ctypes.c_void_p))
        self._comm = out.value

    @staticmethod
    def make_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.call("ocn_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return buf.raw

    @classmethod
    def from_process_group(cls, rank: int, world_size: int):
        """rank 0 creates the id, torch.distributed carries it to the other ranks (any initialised backend)"""
        import torch.distributed as dist
        box = [cls.make_unique_id() if rank == 0 else None]
        if world_size > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(box[0], rank, world_size)

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def all_gather_into_tensor(self, out, inp):
        assert out.numel() == inp.numel() * self.world_size and out.dtype == inp.dtype
        _lib.call("ocn_comm_allgather", self._comm, _check(inp, "input"), _check(out, "output"), inp.numel(), _dt(inp), self._stream())

    def reduce_scatter_sum(self, out, inp):
        assert inp.numel() == out.numel() * self.world_size and out.dtype == inp.dtype
        _lib.call("ocn_comm_reduce_scatter_sum", self._comm, _check(inp, "input"), _check(out, "output"), out.numel(), _dt(inp), self._stream())

    def all_reduce_sum(self, t):
        _lib.call("ocn_comm_allreduce_sum", self._comm, _check(t, "tensor"), t.numel(), _dt(t), self._stream())

    def all_reduce_avg(self, t):
        _lib.call("ocn_comm_allreduce_avg", self._comm, _check(t, "tensor"), t.numel(), _dt(t), self._stream())

    def broadcast(self, t, root=0):
        """every dtype: fp32 / bf16 as such, anything else (integer / bool buffers, fp16 parameters) as its raw bytes -- a broadcast moves bits"""
        if t.dtype in (F32, BF16):
            n, dt = t.numel(), _dt(t)
        else:
            n, dt = t.numel() * t.element_size(), 2
        if n:
            _lib.call("ocn_comm_broadcast", self._comm, _check(t, "tensor"), n, dt, int(root), self._stream())

    def count(self):
        """(ranks, own rank) as the COMMUNICATOR reports them (ncclCommCount / ncclCommUserRank) -- not what the caller passed in"""
        n, r = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.call("ocn_comm_count", self._comm, ctypes.cast(ctypes.byref(n), ctypes.c_void_p), ctypes.cast(ctypes.byref(r), ctypes.c_void_p))
        return n.value, r.value

    def sendrecv(self, send, to_rank, recv, from_rank):
        """one neighbour exchange (loss.py:226-243): ``send`` goes to ``to_rank``, ``recv`` is filled from ``from_rank``; one grouped RCCL operation"""
        assert send.numel() == recv.numel() and send.dtype == recv.dtype
        n, dt = (send.numel(), _dt(send)) if send.dtype in (F32, BF16) else (send.numel() * send.element_size(), 2)
        _lib.call("ocn_comm_sendrecv", self._comm, _check(send, "send"), int(to_rank), _check(recv, "recv"), int(from_rank), n, dt, self._stream())

    def close(self):
        """destroys the communicator; pending collectives are waited for first (ncclCommDestroy does not order itself behind the streams:
        ocn_comm_destroy drains the device before it calls it)"""
        if self._comm:
            _lib.call("ocn_comm_destroy", self._comm)
            self._comm = None
