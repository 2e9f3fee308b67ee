"""Native communicators of the library (csrc/comm.cpp) for ``prove_sharded``: no torch needed.

``LocalCommGroup(G)``  G ranks as G host threads of this process (one context each).
``RcclComm``           RCCL over xGMI, one process per GPU; the 128-byte unique id travels by
                       whatever channel the launcher has (a file, torch.distributed, MPI...).
"""
from __future__ import annotations

import ctypes as C

from . import _lib


class _NativeComm:
    def __init__(self, c: "_lib.CommC", keep):
        self.c = c
        self.rank, self.world = int(c.rank), int(c.world)
        self._keep = keep
        self.error = None


class LocalCommGroup:
    def __init__(self, world: int):
        self.world = world
        self.h = C.c_void_p()
        rc = _lib.lib().ts_comm_local_group_create(world, C.byref(self.h))
        if rc:
            raise _lib.TsError(rc, "ts_comm_local_group_create")

    def comm(self, rank: int) -> _NativeComm:
        c = _lib.CommC()
        rc = _lib.lib().ts_comm_local_get(self.h, rank, C.byref(c))
        if rc:
            raise _lib.TsError(rc, "ts_comm_local_get")
        return _NativeComm(c, self)

    def reset(self):
        """Make a group usable again after an aborted proof (one-shot failure semantics otherwise,
        include/tapstark.h): only once every rank's prove call has returned."""
        rc = _lib.lib().ts_comm_local_group_reset(self.h)
        if rc:
            raise _lib.TsError(rc, "ts_comm_local_group_reset")

    def set_timeout(self, seconds: int):
        rc = _lib.lib().ts_comm_local_group_set_timeout(self.h, int(seconds))
        if rc:
            raise _lib.TsError(rc, "ts_comm_local_group_set_timeout")

    def __del__(self):
        try:
            if self.h:
                _lib.lib().ts_comm_local_group_destroy(self.h)
                self.h = None
        except Exception:
            pass


def rccl_available() -> bool:
    return bool(_lib.lib().ts_rccl_available())


def rccl_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    rc = _lib.lib().ts_rccl_unique_id(buf)
    if rc:
        raise _lib.TsError(rc, "ts_rccl_unique_id")
    return bytes(buf)


class RcclComm(_NativeComm):
    def __init__(self, ctx, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        c = _lib.CommC()
        self.h = C.c_void_p()
        idb = (C.c_uint8 * 128)(*unique_id)
        rc = _lib.lib().ts_comm_rccl_create(ctx.h, idb, rank, world, C.byref(c), C.byref(self.h))
        if rc:
            raise _lib.TsError(rc, "ts_comm_rccl_create")
        super().__init__(c, ctx)

    def info(self) -> dict:
        """What RCCL itself reports (ncclCommCount, ncclCommUserRank, ...) beside the arguments."""
        i = _lib.RcclInfoC()
        rc = _lib.lib().ts_comm_rccl_info(self.h, C.byref(i))
        if rc:
            raise _lib.TsError(rc, "ts_comm_rccl_info")
        return {k: int(getattr(i, k)) for k, _ in _lib.RcclInfoC._fields_}

    def close(self):
        if self.h:
            _lib.lib().ts_comm_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
