"""ctypes face of the ``sh_comm_*`` entry points (include/seghiero_hip.h): the RCCL communicator a host WITHOUT PyTorch would drive the
data-parallel step with.  The package's own multi-GPU path is ``ddp.py`` on ``torch.distributed`` (backend "nccl" = RCCL);
this class exists so that the C ABI is exercised from the test-suite and documents the call order for such a host:

    id = Comm.unique_id()                 # rank 0; the launcher hands the 128 bytes to every rank
    comm = Comm(id, world, rank)          # every rank, on its own HIP device (collective)
    comm.broadcast(weights, 0)            # start-up: weights, BatchNorm running statistics, step
    comm.all_reduce_async(bucket)         # per gradient bucket, from inside backward: runs on the communicator's side stream
    comm.wait()                           # before the optimizer step: the compute stream waits for the side stream
"""
import ctypes

import torch

from ._lib import LIB, SegHieroHipError, status_text

_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int64: 2}
_OPS = {"sum": 0, "min": 1, "max": 2}


def _check(name, rc):
    if rc != 0:
        raise SegHieroHipError(f"{name} failed with status {rc} ({status_text(rc)})")


def _stream(t):
    return torch._C._cuda_getCurrentRawStream(t.device.index)


class Comm:
    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _check("sh_comm_unique_id", LIB.raw("sh_comm_unique_id")(buf))
        return buf.raw

    def __init__(self, unique_id, world, rank):
        if len(unique_id) != 128:
            raise SegHieroHipError("a communicator id is 128 bytes")
        h = ctypes.c_void_p()
        _check("sh_comm_init", LIB.raw("sh_comm_init")(ctypes.c_char_p(unique_id), int(world), int(rank), ctypes.byref(h)))
        self._h, self.world, self.rank = h, int(world), int(rank)

    def close(self):
        if self._h is not None:
            _check("sh_comm_destroy", LIB.raw("sh_comm_destroy")(self._h))
            self._h = None

    def _args(self, t, op):
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPES:
            raise SegHieroHipError("collectives take dense f32 / f64 / i64 device tensors")
        return self._h, t.data_ptr(), t.numel(), _DTYPES[t.dtype], _OPS[op]

    def all_reduce(self, t, op="sum"):
        """in place, in order on torch's current stream"""
        _check("sh_comm_all_reduce", LIB.raw("sh_comm_all_reduce")(*self._args(t, op), _stream(t)))
        return t

    def all_reduce_async(self, t, op="sum"):
        """in place on the communicator's side stream, after everything torch's current stream has queued; `wait()` before reading t"""
        _check("sh_comm_all_reduce_async", LIB.raw("sh_comm_all_reduce_async")(*self._args(t, op), _stream(t)))
        return t

    def wait(self, device=None):
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        _check("sh_comm_wait", LIB.raw("sh_comm_wait")(self._h, torch._C._cuda_getCurrentRawStream(idx)))

    def broadcast(self, t, root=0):
        if not t.is_cuda or not t.is_contiguous():
            raise SegHieroHipError("broadcast takes a dense device tensor")
        _check("sh_comm_broadcast", LIB.raw("sh_comm_broadcast")(self._h, t.data_ptr(), t.numel() * t.element_size(), int(root), _stream(t)))
        return t
