"""Minimal ctypes binding to RCCL for the one collective on the path.

``torch.distributed`` (backend "nccl" = RCCL) sets the job up and remains the fallback, but each
c10d collective costs tens of microseconds of host time -- more than the 50 us frame it would
ride on.  The visible-set exchange therefore calls ``ncclAllGather`` of the RCCL library that
torch already loaded, directly, on the caller's HIP stream.  The communicator is bootstrapped
through the existing process group (rank 0's ncclUniqueId is broadcast once).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_INT64 = 4


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    L = C.CDLL(path if os.path.exists(path) else "librccl.so")
    L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclGetErrorString.argtypes = [C.c_int]
    L.ncclGetErrorString.restype = C.c_char_p
    return L


class Communicator:
    """One RCCL communicator over the ranks of the default process group."""

    def __init__(self, rank, world, device):
        self.L = _load()
        self.rank, self.world = rank, world
        uid = _UniqueId()
        if rank == 0:
            self._check(self.L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        # (a c_char array FIELD reads back NUL-truncated: take the raw 128 bytes)
        t = torch.frombuffer(bytearray(C.string_at(C.byref(uid), 128)), dtype=torch.uint8).to(device)
        dist.broadcast(t, src=0)                                # through the c10d group, once
        raw = bytes(t.cpu().numpy().tobytes())
        C.memmove(C.byref(uid), raw, 128)
        self.comm = C.c_void_p()
        self._check(self.L.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.L.ncclGetErrorString(rc).decode()}")

    def allgather_i64(self, send, recv, stream):
        """recv[r * n : (r + 1) * n] = rank r's send (int64 tensors), on `stream` (torch.cuda.Stream)."""
        self._check(self.L.ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), NCCL_INT64, self.comm,
                                         C.c_void_p(stream.cuda_stream)), "ncclAllGather")

    def destroy(self):
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
