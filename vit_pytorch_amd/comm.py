"""RCCL through libvitk's own C entry points (`vitk_comm_*`, include/vitk.h): what a host that binds the library WITHOUT
torch.distributed uses for the data-parallel gradient exchange (train_vit_decorr.py:68-78).  `parallel.DataParallel` keeps
torch.distributed as its transport (backend "nccl" is RCCL on ROCm); this module is the thin Python face of the C boundary.

    uid = NativeComm.unique_id() on rank 0, handed to the other ranks by the launcher (file, socket, MPI ...)
    comm = NativeComm(rank, world, uid); comm.all_reduce(flat_grads, average=True); comm.close()
"""
import ctypes

import torch

from . import _lib as L
from . import kernels as K

ID_BYTES = 128


class NativeComm:
    def __init__(self, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != ID_BYTES:
            raise ValueError(f"unique id must be {ID_BYTES} bytes")
        self._lib = L.load()
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), ID_BYTES)
        K.check(self._lib.vitk_comm_init(ctypes.cast(buf, ctypes.c_void_p), self.rank, self.world, ctypes.cast(ctypes.byref(h), ctypes.c_void_p)),
                "comm_init")
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(ID_BYTES)
        K.check(L.load().vitk_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "comm_unique_id")
        return buf.raw

    def all_reduce(self, t: torch.Tensor, average: bool = True) -> torch.Tensor:
        """In-place sum (or average) of a contiguous float32 / bfloat16 tensor over the ranks, on the current stream."""
        if not t.is_contiguous() or not t.is_cuda:
            raise L.VitkError("comm.all_reduce: a contiguous GPU tensor is required")
        if t.dtype == torch.float16:
            raise L.VitkError("comm.all_reduce: float16 buffers go through libvitk_f16.so (not wired here)")
        K.check(self._lib.vitk_comm_allreduce(self._h, t.data_ptr(), t.numel(), K.dt(t), int(bool(average)), K._stream()), "comm_allreduce")
        return t

    def close(self):
        if self._h is not None:
            K.check(self._lib.vitk_comm_destroy(self._h), "comm_destroy")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
