"""Collectives for the sharded prover (``ts_comm`` of include/tapstark.h) over torch.distributed.

One process per GPU.  With the ``nccl`` backend (RCCL over xGMI on a real node) the library's device
buffers are wrapped zero-copy (``__cuda_array_interface__``) and the collective is enqueued behind
the library's HIP stream (``torch.cuda.ExternalStream``), so nothing blocks the host.  With ``gloo``
(CPU rehearsal: several ranks sharing one GPU in the tests) the buffers are staged through host
memory.  PyTorch is plumbing here: device memory views, streams and the process group.
"""
from __future__ import annotations

import ctypes as C
import traceback

import torch
import torch.distributed as dist

from . import _lib


class _DevPtr:
    """A raw device pointer as a ``__cuda_array_interface__`` provider (bytes)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 2, "strides": None}


def _wrap(ptr: int, nbytes: int, device) -> torch.Tensor:
    return torch.as_tensor(_DevPtr(ptr, nbytes), device=device)


class TorchComm:
    """``ts_comm`` backed by a torch.distributed process group."""

    def __init__(self, device: torch.device | int | None = None, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.calls = {"all_gather": 0, "broadcast": 0, "bytes": 0}
        self.error = None
        # keep the ctypes thunks alive as long as the struct
        self._ag = _lib.ALL_GATHER_FN(self._all_gather)
        self._bc = _lib.BROADCAST_FN(self._broadcast)
        self.c = _lib.CommC(self.rank, self.world, None, self._ag, self._bc, _lib.ABORT_FN())

    # -- callbacks (called from inside ts_prove_sharded, on the calling thread) -------------------
    def _all_gather(self, _user, send, recv, nbytes, stream):
        try:
            self.calls["all_gather"] += 1
            self.calls["bytes"] += nbytes * self.world
            ext = torch.cuda.ExternalStream(stream, device=self.device)
            s = _wrap(send, nbytes, self.device)
            r = _wrap(recv, nbytes * self.world, self.device)
            if self.backend == "nccl":
                with torch.cuda.stream(ext):
                    dist.all_gather_into_tensor(r, s, group=self.group)
            else:
                ext.synchronize()
                hs = s.cpu()
                parts = [torch.empty_like(hs) for _ in range(self.world)]
                dist.all_gather(parts, hs, group=self.group)
                with torch.cuda.stream(ext):
                    r.copy_(torch.cat(parts))
                ext.synchronize()
            return 0
        except Exception:  # never let an exception cross the C boundary
            self.error = traceback.format_exc()
            return 1

    def _broadcast(self, _user, buf, nbytes, root, stream):
        try:
            self.calls["broadcast"] += 1
            self.calls["bytes"] += nbytes
            ext = torch.cuda.ExternalStream(stream, device=self.device)
            b = _wrap(buf, nbytes, self.device)
            src = dist.get_global_rank(self.group, root) if self.group is not None else root
            if self.backend == "nccl":
                with torch.cuda.stream(ext):
                    dist.broadcast(b, src=src, group=self.group)
            else:
                ext.synchronize()
                hb = b.cpu()
                dist.broadcast(hb, src=src, group=self.group)
                if self.rank != root:
                    with torch.cuda.stream(ext):
                        b.copy_(hb)
                    ext.synchronize()
            return 0
        except Exception:
            self.error = traceback.format_exc()
            return 1
