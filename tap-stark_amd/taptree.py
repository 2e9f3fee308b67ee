"""Taptree-compatible commitment on the device (``ts_tap*`` of include/tapstark.h): mirrors
``CompleteTaptree`` (basic/src/tcs/complete_taptree.rs) and ``TapTreeMmcs``
(basic/src/mmcs/taptree_mmcs.rs:24-119).  Digests are 32-byte strings (Bitcoin's byte order)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .stark import Context, DeviceMatrix, _p, _u32, default_context


def _check(rc: int, what: str, ctx: Context | None = None):
    if rc:
        msg = (ctx._l.ts_last_error(ctx.h) or b"").decode() if ctx is not None else what
        raise _lib.TsError(rc, msg or what)


def _u8(n):
    return (C.c_uint8 * n)()


def tapleaf_hash(script: bytes) -> bytes:
    out = _u8(32)
    _check(_lib.lib().ts_tapleaf_hash(script, len(script), out), "ts_tapleaf_hash")
    return bytes(out)


def tapbranch_hash(a: bytes, b: bytes) -> bytes:
    out = _u8(32)
    _check(_lib.lib().ts_tapbranch_hash((C.c_uint8 * 32)(*a), (C.c_uint8 * 32)(*b), out), "ts_tapbranch_hash")
    return bytes(out)


def winternitz_lock_script(secret: bytes, u32_count: int = 1) -> bytes:
    """Stand-in for ``locking_script_with_type(CompressType::U32)`` (un-vendored crate), built from
    the reference's local copy of the construction (scripts/src/bit_comm/winternitz.rs:171-274)."""
    n = C.c_size_t()
    buf = _u8(1 << 14)
    _check(_lib.lib().ts_tap_winternitz_lock_script(secret, len(secret), u32_count, buf, len(buf), C.byref(n)),
           "ts_tap_winternitz_lock_script")
    return bytes(buf[: n.value])


def _pack(scripts):
    offs = np.zeros(len(scripts) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in scripts])
    return b"".join(scripts), offs


def leaf_script(locks, index: int, values, u32_size: int = 1) -> bytes:
    """``CommitedLeaf::generate_script`` (tcs/mod.rs:197-225): locks[0] = index lock."""
    blob, offs = _pack(locks)
    vals = _u32(values).reshape(-1)
    n_evals = len(locks) - 1
    assert len(vals) == n_evals * u32_size
    n = C.c_size_t()
    cap = len(blob) + 8 * len(vals) + 64
    buf = _u8(cap)
    _check(_lib.lib().ts_tap_leaf_script(blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)), n_evals, u32_size,
                                         index, _p(vals) if len(vals) else None, buf, cap, C.byref(n)),
           "ts_tap_leaf_script")
    return bytes(buf[: n.value])


def verify_inclusion(root: bytes, leaf_hash: bytes, path) -> bool:
    pb = b"".join(path)
    return bool(_lib.lib().ts_taptree_verify_inclusion((C.c_uint8 * 32)(*root), (C.c_uint8 * 32)(*leaf_hash),
                                                       (C.c_uint8 * max(len(pb), 1))(*pb), len(path)))


class CompleteTaptree:
    """complete_taptree.rs:5-161, hashed on the GPU."""

    def __init__(self, handle, ctx: Context | None):
        self.h, self.ctx = handle, ctx
        n = C.c_uint64()
        root = _u8(32)
        _check(_lib.lib().ts_taptree_info(handle, C.byref(n), root), "ts_taptree_info")
        self.leaf_count, self.root = int(n.value), bytes(root)

    @classmethod
    def new_with_scripts(cls, scripts, ctx: Context | None = None) -> "CompleteTaptree":
        ctx = ctx or default_context()
        blob, offs = _pack(scripts)
        h = C.c_void_p()
        _check(ctx._l.ts_taptree_from_scripts(ctx.h, blob or b"\0", offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              len(scripts), C.byref(h)), "ts_taptree_from_scripts", ctx)
        return cls(h, ctx)

    def combine(self, other: "CompleteTaptree") -> "CompleteTaptree":
        h = C.c_void_p()
        _check(_lib.lib().ts_taptree_combine(self.h, other.h, C.byref(h)), "ts_taptree_combine")
        t = CompleteTaptree(h, self.ctx)
        t._parts = (self, other)  # the C++ side shares ownership; keep the wrappers too
        return t

    def get_leaf_proof(self, index: int):
        """(leaf hash, [sibling hashes, leaf-most first])"""
        leaf = _u8(32)
        path = _u8(32 * 130)
        depth = C.c_uint32()
        _check(_lib.lib().ts_taptree_leaf_proof(self.h, index, leaf, path, 130, C.byref(depth)),
               "ts_taptree_leaf_proof", self.ctx)
        pb = bytes(path)
        return bytes(leaf), [pb[32 * k:32 * k + 32] for k in range(depth.value)]

    def __del__(self):
        try:
            if self.h:
                _lib.lib().ts_taptree_free(self.h)
                self.h = None
        except Exception:
            pass


class TapMmcsData:
    def __init__(self, ctx, handle, roots, lock_blob, lock_offs, u32_size):
        self.ctx, self.h, self.roots = ctx, handle, roots
        self.lock_blob, self.lock_offs, self.u32_size = lock_blob, lock_offs, u32_size
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(_lib.lib().ts_tap_mmcs_info(handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "info")
        self.n_mats, self.log_max_height, self.n_evals, self.num_queries = a.value, b.value, c.value, d.value

    def tree_locks(self, q: int):
        n_seg = 1 + self.n_evals
        o = self.lock_offs[q * n_seg:(q + 1) * n_seg + 1]
        return [self.lock_blob[int(o[i]):int(o[i + 1])] for i in range(n_seg)]

    def __del__(self):
        try:
            if self.h:
                _lib.lib().ts_tap_mmcs_free(self.h)
                self.h = None
        except Exception:
            pass


class TapTreeMmcs:
    """``TapTreeMmcs::new(manager, num_queries)`` (taptree_mmcs.rs:31-37).  ``locks_for_tree(q, n_evals)``
    plays the bit-commitment manager: it returns the 1 + n_evals lock scripts of tree q."""

    def __init__(self, num_queries: int, locks_for_tree, u32_size: int = 1, ctx: Context | None = None,
                 host_only: bool = False):
        self.num_queries, self.locks_for_tree, self.u32_size = num_queries, locks_for_tree, u32_size
        self.ctx = None if host_only else (ctx or default_context())  # verify_batch needs no GPU

    def commit(self, inputs):
        ctx = self.ctx
        mats = [m if isinstance(m, DeviceMatrix) else DeviceMatrix.upload(ctx, m) for m in inputs]
        total_w = sum(m.dims()[1] for m in mats)
        n_evals = total_w // self.u32_size
        locks = []
        for q in range(self.num_queries):
            lq = list(self.locks_for_tree(q, n_evals))
            assert len(lq) == 1 + n_evals
            locks += lq
        blob, offs = _pack(locks)
        arr = (C.c_void_p * len(mats))(*[m.h for m in mats])
        roots = _u8(32 * self.num_queries)
        h = C.c_void_p()
        _check(ctx._l.ts_tap_mmcs_commit(ctx.h, len(mats), arr, self.u32_size, self.num_queries, blob,
                                         offs.ctypes.data_as(C.POINTER(C.c_uint64)), roots, C.byref(h)),
               "ts_tap_mmcs_commit", ctx)
        rb = bytes(roots)
        return ([rb[32 * q:32 * q + 32] for q in range(self.num_queries)],
                TapMmcsData(ctx, h, None, blob, offs, self.u32_size))

    def open_batch(self, query_times_index: int, index: int, data: TapMmcsData, total_width: int):
        rows = np.zeros(max(total_width, 1), dtype=np.uint32)
        path = _u8(32 * max(data.log_max_height, 1))
        n = C.c_size_t()
        cap = len(data.lock_blob) // max(self.num_queries, 1) + 8 * total_width + 64
        script = _u8(cap)
        _check(_lib.lib().ts_tap_mmcs_open_batch(data.h, query_times_index, index, _p(rows), path, script, cap,
                                                 C.byref(n)), "ts_tap_mmcs_open_batch", self.ctx)
        pb = bytes(path)
        return (rows[:total_width], [pb[32 * k:32 * k + 32] for k in range(data.log_max_height)],
                bytes(script[: n.value]))

    def verify_batch(self, locks, index: int, opened_values, path, root: bytes) -> bool:
        blob, offs = _pack(locks)
        vals = _u32(opened_values).reshape(-1)
        pb = b"".join(path)
        ok = C.c_int()
        _check(_lib.lib().ts_tap_mmcs_verify_batch(blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   len(locks) - 1, self.u32_size, index, _p(vals),
                                                   (C.c_uint8 * max(len(pb), 1))(*pb), len(path),
                                                   (C.c_uint8 * 32)(*root), C.byref(ok)), "verify_batch")
        return bool(ok.value)


# ------------------------------------------------------------------ prove / verify over taptrees
def lock_table_shapes(width: int, quotient_degree: int, log_n: int):
    """(n_evals, u32_size) of every commitment of one proof, in commit order: trace, quotient chunks,
    then the log2(n) FRI rounds (rows of two EF4 elements)."""
    return [(width, 1), (4 * quotient_degree, 1)] + [(2, 4)] * log_n


def make_lock_table(num_queries: int, width: int, quotient_degree: int, log_n: int, lock_for):
    """The flat lock-script table ``ts_prove_tap`` takes: for every commitment, for every tree q, the
    index lock then one lock per evaluation.  ``lock_for(commit, q, slot, u32_count) -> bytes`` plays
    the reference's bit-commitment manager (tcs/mod.rs:251-260 ``assign_bc``)."""
    locks = []
    for ci, (n_evals, u32) in enumerate(lock_table_shapes(width, quotient_degree, log_n)):
        for q in range(num_queries):
            for s in range(1 + n_evals):
                locks.append(lock_for(ci, q, s, 1 if s == 0 else u32))
    return locks


def prove_tap(config, air, challenger, trace, public_values, locks, comm=None):
    """``uni_stark::prove`` with ``TapTreeMmcs`` as both MMCSs (``ts_prove_tap``); returns the
    TSPF v2 words.  With ``comm`` (``comm.LocalCommGroup.comm(r)``, ``comm.RcclComm``,
    ``dist.TorchComm``) the commitments of this ONE proof are split by tree over ``comm.world`` GPUs
    (``ts_prove_tap_sharded``): every rank passes the whole trace and gets the whole, identical proof."""
    from .air import BaseAir, air_tape
    from .stark import CompiledAir

    pcs = config.pcs
    ctx = pcs.ctx
    pis = _u32(public_values)
    if isinstance(air, BaseAir):
        air = CompiledAir(ctx, air_tape(air, len(pis)))
    if not isinstance(trace, DeviceMatrix):
        trace = DeviceMatrix.upload(ctx, trace)
    n, w = trace.dims()
    log_n = n.bit_length() - 1
    log_N = log_n + pcs.fri.log_blowup
    qd = 1 << air.log_quotient_degree
    Q = pcs.fri.num_queries
    cap = 64 + 8 * w + 16 * qd + 16 * Q + 8 * Q * log_n + Q * (16 + w + 5 * qd + 16 * log_N + log_n * (9 + 8 * log_N))
    out = np.zeros(cap, dtype=np.uint32)
    n_words = C.c_size_t()
    cfg = pcs.fri._c()
    blob, offs = _pack(locks)
    offs_p = offs.ctypes.data_as(C.POINTER(C.c_uint64))
    pis_p = _p(pis) if len(pis) else None
    if comm is None:
        rc = ctx._l.ts_prove_tap(ctx.h, C.byref(cfg), air.h, challenger.h, trace.h, pis_p, len(pis), blob,
                                 offs_p, len(locks), _p(out), cap, C.byref(n_words))
    else:
        rc = ctx._l.ts_prove_tap_sharded(ctx.h, C.byref(cfg), C.byref(comm.c), air.h, challenger.h, trace.h,
                                         pis_p, len(pis), blob, offs_p, len(locks), _p(out), cap,
                                         C.byref(n_words))
        if rc == 7 and getattr(comm, "error", None):
            raise RuntimeError("communicator callback failed:\n" + comm.error)
    ctx.check(rc)
    return out[: n_words.value].copy()


def verify_tap(config, air, challenger, proof_words, public_values, locks) -> int:
    """``uni_stark::verify`` for a TSPF v2 proof (host only): 0 = accept, else the verdict code."""
    from .air import BaseAir, air_tape
    from .stark import CompiledAir

    pis = _u32(public_values)
    if isinstance(air, BaseAir):
        air = CompiledAir(None, air_tape(air, len(pis)))
    words = _u32(proof_words)
    cfg = config.pcs.fri._c()
    blob, offs = _pack(locks)
    verdict = C.c_int(-1)
    _check(_lib.lib().ts_verify_tap(C.byref(cfg), air.h, challenger.h, _p(words), len(words),
                                    _p(pis) if len(pis) else None, len(pis), blob,
                                    offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(locks), C.byref(verdict)),
           "ts_verify_tap")
    return int(verdict.value)
