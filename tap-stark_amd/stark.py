"""Python binding of the host-side mirror of the reference's prover interface.

The prover itself (orchestration, transcript, kernels) is C++/HIP behind the C ABI
(include/tapstark.h); this module only gives it the reference's names so that tests and benches
read like the reference's own:

* ``FriConfig``            -- reference fri/src/config.rs:11-16
* ``TwoAdicFriPcs``        -- reference fri/src/two_adic_pcs.rs:38-61 (commit/open on the device)
* ``StarkConfig``          -- reference uni-stark/src/config.rs:64-101
* ``BfChallenger``         -- reference basic/src/challenger/mod.rs:67-137
* ``prove``                -- reference uni-stark/src/prover.rs:25-35
* ``Proof``                -- reference uni-stark/src/proof.rs:17-37 + fri/src/proof.rs:13-33
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .air import BaseAir, air_tape

P = 0x78000001
TSPF_MAGIC = 0x46505354


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def _p(a: np.ndarray):
    return a.ctypes.data_as(_lib.u32p)


class Context:
    """One per GPU (``ts_ctx``)."""

    def __init__(self, device: int = 0):
        self._l = _lib.lib()
        h = C.c_void_p()
        rc = self._l.ts_ctx_create(device, C.byref(h))
        if rc:
            raise _lib.TsError(rc, (self._l.ts_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def check(self, rc: int):
        if rc:
            raise _lib.TsError(rc, (self._l.ts_last_error(self.h) or b"").decode())

    def synchronize(self):
        self.check(self._l.ts_ctx_synchronize(self.h))

    @property
    def stream(self) -> int:
        return int(self._l.ts_ctx_stream(self.h) or 0)

    def alu_ceiling(self, kind: int) -> float:
        """Whole-chip NTT butterflies/s (kind 0) or Blake3 compressions/s (kind 1), no memory traffic."""
        r = C.c_double()
        self.check(self._l.ts_bench_alu(self.h, kind, C.byref(r)))
        return float(r.value)

    def bench_stage(self, stage: int, log_n: int, width: int, log_blowup: int, reps: int) -> float:
        """Mean ms of one repetition of a stage on resident data: 0 = coset LDE, 1 = Merkle hashing."""
        r = C.c_double()
        self.check(self._l.ts_bench_stage(self.h, stage, log_n, width, log_blowup, reps, C.byref(r)))
        return float(r.value)

    def set_timing(self, enabled: bool):
        self.check(self._l.ts_ctx_set_timing(self.h, int(enabled)))

    def take_timings(self) -> list[tuple[str, float]]:
        buf = C.create_string_buffer(1 << 16)
        self.check(self._l.ts_ctx_take_timings(self.h, buf, len(buf)))
        out = []
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.rsplit("=", 1)
                out.append((k, float(v)))
        return out

    def set_replay(self, mode: int):
        """``ts_ctx_set_replay`` (measurement aid: 1 record, 2 replay without the mid-proof syncs, 0 off)."""
        self.check(self._l.ts_ctx_set_replay(self.h, int(mode)))

    def set_kernel_timing(self, enabled: bool):
        self.check(self._l.ts_ctx_set_kernel_timing(self.h, int(enabled)))

    def take_kernel_timings(self) -> dict[str, tuple[int, float]]:
        """kernel name -> (launches, total ms), from HIP events on the context's stream."""
        buf = C.create_string_buffer(1 << 16)
        self.check(self._l.ts_ctx_take_kernel_timings(self.h, buf, len(buf)))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.rsplit("=", 1)
                cnt, ms = v.split(":")
                out[k] = (int(cnt), float(ms))
        return out

    def graph_stats(self) -> dict:
        """TS_FRI_GRAPH diagnostics (``ts_ctx_graph_stats``)."""
        out = (C.c_uint64 * 4)()
        self.check(self._l.ts_ctx_graph_stats(self.h, out))
        return {"replays": int(out[0]), "fallbacks": int(out[1]), "shapes": int(out[2]),
                "pool_bytes": int(out[3]), "reserve_failures": self.stat(4)}

    def stat(self, which: int) -> int:
        """``ts_ctx_stat``: 0-3 as graph_stats, 4 graph reservations refused, 5 local-quotient fall-backs."""
        v = C.c_uint64()
        self.check(self._l.ts_ctx_stat(self.h, which, C.byref(v)))
        return int(v.value)

    def close(self):
        if self.h:
            self._l.ts_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class PinnedHostMatrix:
    """A row-major u32 matrix in page-locked host memory (``ts_host_alloc``), as a numpy view."""

    def __init__(self, height: int, width: int):
        self.shape = (height, width)
        p = C.c_void_p()
        rc = _lib.lib().ts_host_alloc(height * width * 4, C.byref(p))
        if rc:
            raise _lib.TsError(rc, "ts_host_alloc")
        self.ptr = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(height, width))

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().ts_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class DeviceMatrix:
    """``RowMajorMatrix<Val>`` resident in HBM (``ts_matrix``)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.h = handle

    @classmethod
    def upload(cls, ctx: Context, values) -> "DeviceMatrix":
        values = _u32(values)
        assert values.ndim == 2
        h = C.c_void_p()
        ctx.check(ctx._l.ts_matrix_upload(ctx.h, _p(values), values.shape[0], values.shape[1],
                                          C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def upload_async(cls, ctx: Context, pinned: "PinnedHostMatrix") -> "DeviceMatrix":
        """H2D from page-locked memory without waiting (``ts_matrix_upload_async``)."""
        h = C.c_void_p()
        ctx.check(ctx._l.ts_matrix_upload_async(ctx.h, pinned.ptr, pinned.shape[0], pinned.shape[1], C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device_ptr(cls, ctx: Context, ptr: int, height: int, width: int) -> "DeviceMatrix":
        h = C.c_void_p()
        ctx.check(ctx._l.ts_matrix_from_device(ctx.h, C.c_void_p(ptr), height, width, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def fibonacci(cls, ctx: Context, a: int, b: int, n: int) -> "DeviceMatrix":
        """``generate_trace_rows(a, b, n)`` (uni-stark/tests/fib_air.rs:59-78) computed in HBM."""
        h = C.c_void_p()
        ctx.check(ctx._l.ts_trace_fibonacci(ctx.h, a, b, n, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def synth_mul(cls, ctx: Context, n: int, width: int = 64, seed: int | None = None) -> "DeviceMatrix":
        """The SynthMulAir-``width`` trace of ``airs.generate_synth_mul_trace``, computed in HBM."""
        from .airs import SPLITMIX_SEED
        h = C.c_void_p()
        ctx.check(ctx._l.ts_trace_synth_mul(ctx.h, n, width, SPLITMIX_SEED if seed is None else seed,
                                            C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def synth_ext(cls, ctx: Context, n: int, width: int = 163, seed: int | None = None) -> "DeviceMatrix":
        """The SynthExt-``width`` trace of ``airs.generate_synth_ext_trace``, computed in HBM."""
        from .airs import SPLITMIX_SEED
        h = C.c_void_p()
        ctx.check(ctx._l.ts_trace_synth_ext(ctx.h, n, width, SPLITMIX_SEED if seed is None else seed,
                                            C.byref(h)))
        return cls(ctx, h)

    def dims(self):
        hh, ww = C.c_uint64(), C.c_uint32()
        self.ctx.check(self.ctx._l.ts_matrix_dims(self.h, C.byref(hh), C.byref(ww)))
        return int(hh.value), int(ww.value)

    def download(self) -> np.ndarray:
        hh, ww = self.dims()
        out = np.zeros((hh, ww), dtype=np.uint32)
        self.ctx.check(self.ctx._l.ts_matrix_download(self.ctx.h, self.h, _p(out)))
        return out

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx._l.ts_matrix_free(self.ctx.h, self.h)
        except Exception:
            pass


class CompiledAir:
    """``ts_air``.  With ``ctx=None`` the AIR is host-only (degree rules, ``verify``): no GPU."""

    def __init__(self, ctx: "Context | None", tape):
        self.ctx = ctx
        self._l = _lib.lib()
        self.tape = _u32(tape)
        h = C.c_void_p()
        rc = self._l.ts_air_compile(ctx.h if ctx else None, _p(self.tape), len(self.tape), C.byref(h))
        if rc:
            msg = self._l.ts_last_error(ctx.h if ctx else None) or b""
            raise _lib.TsError(rc, msg.decode())
        self.h = h
        w, npub, deg, lqd = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._l.ts_air_info(h, C.byref(w), C.byref(npub), C.byref(deg), C.byref(lqd))
        self.width, self.n_public = int(w.value), int(npub.value)
        self.max_constraint_degree, self.log_quotient_degree = int(deg.value), int(lqd.value)

    @property
    def is_jit(self) -> bool:
        """The hiprtc-specialised quotient kernel is loaded (else the on-device interpreter runs;
        large programs are compiled in the background and switch over when ready)."""
        return bool(self._l.ts_air_is_jit(self.h))

    def jit_wait(self) -> tuple[int, float]:
        """Joins a background specialisation: (state 0 none | 3 loaded | 4 failed, compile seconds)."""
        st, secs = C.c_int(), C.c_double()
        rc = self._l.ts_air_jit_wait(self.ctx.h, self.h, C.byref(st), C.byref(secs))
        if rc:
            raise self._err(rc)
        return int(st.value), float(secs.value)

    def _err(self, rc: int):
        msg = self._l.ts_last_error(self.ctx.h if self.ctx else None) or b""
        return _lib.TsError(rc, msg.decode())

    def program(self) -> dict:
        """The register program the tape was lowered to (``ts_air_program``): ``n_regs``, ``code``
        (n_instr, 4) = {op, dst, a, b}, ``consts`` (canonical), ``const_public`` (index or 0xffffffff)."""
        n = C.c_size_t()
        self._l.ts_air_program(self.h, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=np.uint32)
        rc = self._l.ts_air_program(self.h, _p(out), len(out), C.byref(n))
        if rc:
            raise self._err(rc)
        n_regs, n_instr, n_consts = (int(v) for v in out[:3])
        code = out[3:3 + 4 * n_instr].reshape(n_instr, 4)
        consts = out[3 + 4 * n_instr:3 + 4 * n_instr + n_consts]
        return {"n_regs": n_regs, "code": code, "consts": consts, "const_public": out[3 + 4 * n_instr + n_consts:]}

    def jit_source(self) -> str:
        n = C.c_size_t()
        self._l.ts_air_jit_source(self.h, None, 0, C.byref(n))
        buf = C.create_string_buffer(n.value)
        rc = self._l.ts_air_jit_source(self.h, buf, n.value, C.byref(n))
        if rc:
            raise self._err(rc)
        return buf.raw[:n.value].decode()

    def jit_compile(self, arch: str = "gfx950") -> tuple[bytes, float]:
        """(code object, compile seconds) of the hiprtc-specialised quotient kernel; no GPU needed."""
        cap = 64 << 20
        buf = C.create_string_buffer(cap)
        n, secs = C.c_size_t(), C.c_double()
        rc = self._l.ts_air_jit_compile(self.h, arch.encode(), buf, cap, C.byref(n), C.byref(secs))
        if rc:
            raise self._err(rc)
        return buf.raw[:n.value], float(secs.value)

    def __del__(self):
        try:
            if self.h and (self.ctx is None or self.ctx.h):
                self._l.ts_air_free(self.ctx.h if self.ctx else None, self.h)
        except Exception:
            pass


@dataclass
class FriConfig:
    """reference fri/src/config.rs:11-16; ``mmcs`` is the built-in Blake3 Merkle MMCS."""
    log_blowup: int
    num_queries: int
    proof_of_work_bits: int

    def blowup(self) -> int:
        return 1 << self.log_blowup

    def _c(self):
        return _lib.FriConfigC(self.log_blowup, self.num_queries, self.proof_of_work_bits)


class PcsData:
    """``Pcs::ProverData``: committed LDEs + Merkle tree in HBM (``ts_pcs_data``)."""

    def __init__(self, ctx: Context, handle, root: np.ndarray):
        self.ctx, self.h, self.root = ctx, handle, root
        n, lh = C.c_uint32(), C.c_uint32()
        ctx.check(ctx._l.ts_pcs_data_info(handle, C.byref(n), C.byref(lh)))
        self.n_mats, self.log_height = int(n.value), int(lh.value)
        self.dims = []  # (LDE height, width) per committed matrix
        for i in range(self.n_mats):
            hh, ww = C.c_uint64(), C.c_uint32()
            ctx.check(ctx._l.ts_pcs_data_matrix_info(handle, i, C.byref(hh), C.byref(ww)))
            self.dims.append((int(hh.value), int(ww.value)))

    def lde(self, idx: int, width: int | None = None) -> np.ndarray:
        out = np.zeros(self.dims[idx], dtype=np.uint32)
        self.ctx.check(self.ctx._l.ts_pcs_data_lde(self.ctx.h, self.h, idx, _p(out)))
        return out

    def digests(self, level: int) -> np.ndarray:
        out = np.zeros(((1 << self.log_height) >> level, 8), dtype=np.uint32)
        self.ctx.check(self.ctx._l.ts_pcs_data_digests(self.ctx.h, self.h, level, _p(out)))
        return out

    def open_batch(self, index: int, total_width: int | None = None):
        if total_width is None:
            total_width = sum(w for _, w in self.dims)
        rows = np.zeros(max(total_width, 1), dtype=np.uint32)
        path = np.zeros((max(self.log_height, 1), 8), dtype=np.uint32)
        self.ctx.check(self.ctx._l.ts_pcs_open_batch(self.ctx.h, self.h, index, _p(rows), _p(path)))
        return rows[:total_width], path[: self.log_height]

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx._l.ts_pcs_data_free(self.ctx.h, self.h)
        except Exception:
            pass


class Blake3Mmcs:
    """``BFMmcs`` (basic/src/mmcs/bf_mmcs.rs:17-68) with the build-defined Blake3 Merkle tree, on
    the device: ``commit`` to matrices as given (no LDE), ``open_batch`` via the returned PcsData."""

    def __init__(self, ctx: Context | None = None):
        self.ctx = ctx or default_context()

    def commit(self, inputs) -> tuple[np.ndarray, PcsData]:
        ctx = self.ctx
        mats = [m if isinstance(m, DeviceMatrix) else DeviceMatrix.upload(ctx, m) for m in inputs]
        arr = (C.c_void_p * len(mats))(*[m.h for m in mats])
        root = np.zeros(8, dtype=np.uint32)
        h = C.c_void_p()
        ctx.check(ctx._l.ts_mmcs_commit(ctx.h, len(mats), arr, _p(root), C.byref(h)))
        return root, PcsData(ctx, h, root)

    def commit_matrix(self, m):
        return self.commit([m])

    def open_batch(self, index: int, data: PcsData):
        return data.open_batch(index)


class TwoAdicFriPcs:
    """reference fri/src/two_adic_pcs.rs:38-61,203-419 on the device."""

    def __init__(self, fri: FriConfig, ctx: Context | None = None, host_only: bool = False):
        self.fri = fri
        self.ctx = None if host_only else (ctx or default_context())

    def natural_domain_for_degree(self, degree: int):
        return (degree.bit_length() - 1, 1)  # (log_n, shift), two_adic_pcs.rs:220-226

    def commit(self, evaluations) -> tuple[np.ndarray, PcsData]:
        """``evaluations``: list of ((log_n, shift), DeviceMatrix | ndarray). Matrices are consumed."""
        ctx = self.ctx
        mats, shifts = [], []
        for (log_n, shift), m in evaluations:
            if not isinstance(m, DeviceMatrix):
                m = DeviceMatrix.upload(ctx, m)
            assert m.dims()[0] == 1 << log_n
            mats.append(m)
            shifts.append(shift)
        arr = (C.c_void_p * len(mats))(*[m.h for m in mats])
        sh = _u32(shifts)
        root = np.zeros(8, dtype=np.uint32)
        h = C.c_void_p()
        cfg = self.fri._c()
        ctx.check(ctx._l.ts_pcs_commit(ctx.h, C.byref(cfg), len(mats), arr, _p(sh), _p(root),
                                       C.byref(h)))
        return root, PcsData(ctx, h, root)

    def quotient_chunks(self, trace_data: PcsData, air: CompiledAir, public_values, alpha):
        qd = 1 << air.log_quotient_degree
        out = (C.c_void_p * qd)()
        pis = _u32(public_values)
        pis_p = _p(pis) if len(pis) else None
        self.ctx.check(self.ctx._l.ts_quotient_chunks(self.ctx.h, trace_data.h, self.fri.log_blowup,
                                                      air.h, pis_p, len(pis), _p(_u32(alpha)), out))
        return [DeviceMatrix(self.ctx, C.c_void_p(out[c])) for c in range(qd)]

    def open_reduce(self, trace_data: PcsData, quotient_data: PcsData, width: int, zeta, batch_alpha):
        qd = quotient_data.n_mats
        opened = np.zeros((2 * width + 4 * qd, 4), dtype=np.uint32)
        ro = np.zeros((1 << trace_data.log_height, 4), dtype=np.uint32)
        cfg = self.fri._c()
        self.ctx.check(self.ctx._l.ts_pcs_open_reduce(self.ctx.h, C.byref(cfg), trace_data.h,
                                                      quotient_data.h, _p(_u32(zeta)),
                                                      _p(_u32(batch_alpha)), _p(opened), _p(ro)))
        return opened, ro

    def open(self, rounds, challenger: "BfChallenger"):
        """``Pcs::open`` (two_adic_pcs.rs:260-419).  ``rounds``: list of (PcsData, points) with
        ``points[m]`` the list of EF4 opening points of matrix m.  Returns (opened, fri_proof):
        ``opened[r][m][p]`` is a (width, 4) array, ``fri_proof`` the FriProof words (TSPF v1)."""
        n_pts, pts, total = [], [], 0
        for data, points in rounds:
            assert len(points) == data.n_mats
            for (_, w), pl in zip(data.dims, points):
                n_pts.append(len(pl))
                pts.extend(_u32(z).reshape(4) for z in pl)
                total += w * len(pl)
        handles = (C.c_void_p * len(rounds))(*[d.h for d, _ in rounds])
        n_pts_a = _u32(n_pts)
        pts_a = _u32(np.concatenate(pts)) if pts else np.zeros(4, dtype=np.uint32)
        opened = np.zeros((max(total, 1), 4), dtype=np.uint32)
        # exact FriProof size (DESIGN.md "Proof format")
        log_max = max(d.log_height for d, _ in rounds)
        R, Q = max(log_max - self.fri.log_blowup, 0), self.fri.num_queries
        per_q = 1 + sum(2 + d.n_mats + sum(w for _, w in d.dims) + 8 * d.log_height for d, _ in rounds)
        per_q += sum(9 + 8 * (log_max - 1 - i) for i in range(R))
        cap = 2 + 8 * R + Q * per_q + 5
        cfg = self.fri._c()
        proof = np.zeros(cap, dtype=np.uint32)
        n_o, n_p = C.c_size_t(), C.c_size_t()
        self.ctx.check(self.ctx._l.ts_pcs_open(self.ctx.h, C.byref(cfg), challenger.h, len(rounds),
                                               handles, _p(n_pts_a), _p(pts_a), _p(opened),
                                               opened.size, C.byref(n_o), _p(proof), cap,
                                               C.byref(n_p)))
        assert n_o.value == 4 * total
        out, k = [], 0
        for data, points in rounds:
            r_out = []
            for (_, w), pl in zip(data.dims, points):
                m_out = []
                for _ in pl:
                    m_out.append(opened[k:k + w].copy())
                    k += w
                r_out.append(m_out)
            out.append(r_out)
        return out, proof[: n_p.value].copy()

    def verify(self, rounds, fri_proof, challenger: "BfChallenger") -> None:
        """``Pcs::verify`` (two_adic_pcs.rs:421-534), host only.  ``rounds``: list of
        (commitment, mats) with ``mats[m] = (log_degree, [(point, values (width, 4)), ...])``.
        Raises :class:`VerificationError`."""
        roots, per_round, logs, widths, n_pts, pts, vals = [], [], [], [], [], [], []
        for root, mats in rounds:
            roots.append(_u32(root).reshape(8))
            per_round.append(len(mats))
            for log_degree, openings in mats:
                logs.append(log_degree)
                n_pts.append(len(openings))
                width = None
                for z, v in openings:
                    v = _u32(v).reshape(-1, 4)
                    width = len(v)
                    pts.append(_u32(z).reshape(4))
                    vals.append(v.reshape(-1))
                widths.append(width or 0)
        fri_proof = _u32(fri_proof)
        verdict = C.c_int(9)
        cfg = self.fri._c()
        cat = lambda xs: _u32(np.concatenate(xs)) if xs else np.zeros(4, dtype=np.uint32)  # noqa: E731
        rc = _lib.lib().ts_pcs_verify(C.byref(cfg), challenger.h, len(rounds), _p(cat(roots)),
                                      _p(_u32(per_round)), _p(_u32(logs)), _p(_u32(widths)),
                                      _p(_u32(n_pts)), _p(cat(pts)), _p(cat(vals)), _p(fri_proof),
                                      len(fri_proof), C.byref(verdict))
        if rc:
            raise _lib.TsError(rc, "ts_pcs_verify")
        if verdict.value:
            raise VerificationError(verdict.value)

    def fri_prove(self, inputs, challenger: "BfChallenger") -> np.ndarray:
        """``bf_prove`` alone (fri/src/prover.rs:19-63) the way fri/tests/fri.rs:100-119 calls it:
        ``inputs`` are (len, 4) EF4 vectors of strictly descending power-of-two lengths; the input
        opening proof is the literal reduced openings.  Returns the FriProof words."""
        ins = [_u32(v).reshape(-1, 4) for v in inputs]
        logs = _u32([len(v).bit_length() - 1 for v in ins])
        ptrs = (_lib.u32p * len(ins))(*[_p(v) for v in ins])
        R = max(int(logs[0]) - self.fri.log_blowup, 0)
        cap = 16 + 8 * R + self.fri.num_queries * (1 + 5 * len(ins) + sum(9 + 8 * (int(logs[0]) - 1 - i)
                                                                           for i in range(R)))
        out = np.zeros(cap, dtype=np.uint32)
        n = C.c_size_t()
        cfg = self.fri._c()
        self.ctx.check(self.ctx._l.ts_fri_prove(self.ctx.h, C.byref(cfg), challenger.h, len(ins),
                                                _p(logs), ptrs, _p(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def fri_verify(self, proof, challenger: "BfChallenger") -> None:
        """fri/src/verifier.rs:20-98 for a proof of :meth:`fri_prove` (host only)."""
        proof = _u32(proof)
        verdict = C.c_int(9)
        cfg = self.fri._c()
        rc = _lib.lib().ts_fri_verify(C.byref(cfg), challenger.h, _p(proof), len(proof), C.byref(verdict))
        if rc:
            raise _lib.TsError(rc, "ts_fri_verify")
        if verdict.value:
            raise VerificationError(verdict.value)

    def fold_matrix(self, vec, beta) -> np.ndarray:
        vec = _u32(vec)
        h = vec.shape[0] // 2
        out = np.zeros((h, 4), dtype=np.uint32)
        self.ctx.check(self.ctx._l.ts_fri_fold(self.ctx.h, _p(vec), h, _p(_u32(beta)), _p(out)))
        return out


@dataclass
class StarkConfig:
    """reference uni-stark/src/config.rs:64-101 (Challenge = EF4, Challenger = BfChallenger)."""
    pcs: TwoAdicFriPcs


class BfChallenger:
    """reference basic/src/challenger/mod.rs:67-137 (host side, ``ts_challenger``)."""

    def __init__(self, permutation: int = 0, sample_ext: bool = True, _handle=None):
        self._l = _lib.lib()
        if _handle is None:
            h = C.c_void_p()
            rc = self._l.ts_chal_new(permutation, int(sample_ext), C.byref(h))
            if rc:
                raise _lib.TsError(rc, "ts_chal_new")
            _handle = h
        self.h = _handle

    def clone(self) -> "BfChallenger":
        h = C.c_void_p()
        rc = self._l.ts_chal_clone(self.h, C.byref(h))
        if rc:
            raise _lib.TsError(rc, "ts_chal_clone")
        return BfChallenger(_handle=h)

    def observe(self, word: int):
        self._l.ts_chal_observe(self.h, word)

    def observe_commitment(self, d):
        self._l.ts_chal_observe_commitment(self.h, _p(_u32(d)))

    def sample(self) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint32)
        self._l.ts_chal_sample(self.h, _p(out))
        return out

    def sample_bits(self, bits: int) -> int:
        return int(self._l.ts_chal_sample_bits(self.h, bits))

    def check_witness(self, bits: int, witness: int) -> bool:
        return bool(self._l.ts_chal_check_witness(self.h, bits, witness))

    def grind(self, bits: int) -> int:
        w = C.c_uint32()
        rc = self._l.ts_chal_grind(self.h, bits, C.byref(w))
        if rc:
            raise _lib.TsError(rc, "failed to find witness")
        return int(w.value)

    def state(self) -> np.ndarray:
        out = np.zeros(34, dtype=np.uint32)
        self._l.ts_chal_state(self.h, _p(out))
        return out

    def __del__(self):
        try:
            if self.h:
                self._l.ts_chal_free(self.h)
        except Exception:
            pass


# ------------------------------------------------------------------------ Proof
@dataclass
class BatchOpening:
    """reference fri/src/two_adic_pcs.rs:63-68"""
    opened_values: list
    opening_proof: np.ndarray


@dataclass
class QueryProof:
    """reference fri/src/proof.rs:28-33"""
    input_proof: list
    commit_phase_openings: list


class Proof:
    """reference uni-stark/src/proof.rs:17-37 (+ FriProof fri/src/proof.rs:13-21) over the TSPF v1
    words the library writes; ``words`` keeps the wire form.  The structured fields (``degree_bits``,
    ``trace_commit``, ``quotient_commit``, ``trace_local``, ``trace_next``, ``quotient_chunks``,
    ``commit_phase_commits``, ``query_proofs``, ``final_poly``, ``pow_witness``) are parsed on first
    access: ``prove()`` hands back the words without spending interpreter time on them."""

    _FIELDS = ("degree_bits", "trace_commit", "quotient_commit", "trace_local", "trace_next",
               "quotient_chunks", "commit_phase_commits", "query_proofs", "final_poly", "pow_witness")

    def __init__(self, words):
        self.words = np.asarray(words, dtype=np.uint32)

    def __getattr__(self, name):  # only reached for attributes not set yet
        if name in Proof._FIELDS:
            self._parse()
            return self.__dict__[name]
        raise AttributeError(name)

    def to_postcard(self) -> bytes:
        """postcard bytes of the reference's serde ``Proof`` (``ts_proof_to_postcard``)."""
        l = _lib.lib()
        w = _u32(self.words)
        out = np.zeros(5 * len(w) + 16, dtype=np.uint8)  # a varint takes at most 5 bytes per word
        n = C.c_size_t()
        rc = l.ts_proof_to_postcard(_p(w), len(w), out.ctypes.data_as(C.POINTER(C.c_uint8)), len(out),
                                    C.byref(n))
        if rc:
            raise _lib.TsError(rc, "ts_proof_to_postcard")
        return out[: n.value].tobytes()

    @classmethod
    def from_postcard(cls, data: bytes, version: int = 0) -> "Proof":
        """``version``: 0 infers TSPF v1 / v2 from the number of roots per commitment; 2 must be
        given for a proof over taptrees made with one query (``ts_proof_from_postcard_v``)."""
        l = _lib.lib()
        b = np.frombuffer(data, dtype=np.uint8).copy()
        out = np.zeros(len(b) + 16, dtype=np.uint32)  # every word takes at least one byte
        n = C.c_size_t()
        rc = l.ts_proof_from_postcard_v(b.ctypes.data_as(C.POINTER(C.c_uint8)), len(b), int(version),
                                        _p(out), len(out), C.byref(n))
        if rc:
            raise _lib.TsError(rc, "ts_proof_from_postcard")
        return cls.parse(out[: n.value].copy())

    @classmethod
    def parse(cls, words: np.ndarray) -> "Proof":
        """Eager form: raises ``ValueError`` on a malformed buffer."""
        pf = cls(words)
        pf._parse()
        return pf

    def _parse(self) -> None:
        w = self.words
        pos = 0

        def take(n):
            nonlocal pos
            out = w[pos:pos + n]
            if len(out) != n:
                raise ValueError("truncated proof")
            pos += n
            return out

        magic, version, degree_bits, width, qd = (int(x) for x in take(5))
        if magic != TSPF_MAGIC or version not in (1, 2):
            raise ValueError("not a TSPF v1/v2 proof")
        # v2 (proofs over the taptree MMCS, ts_prove_tap): num_queries roots per commitment, the
        # commitment fields are (num_queries, 8) arrays
        nr = int(take(1)[0]) if version == 2 else 1
        d = {"degree_bits": degree_bits, "query_proofs": [], "version": version}
        if version == 1:
            d["trace_commit"], d["quotient_commit"] = take(8), take(8)
        else:
            d["trace_commit"], d["quotient_commit"] = take(8 * nr).reshape(nr, 8), take(8 * nr).reshape(nr, 8)
        d["trace_local"] = take(4 * width).reshape(width, 4)
        d["trace_next"] = take(4 * width).reshape(width, 4)
        d["quotient_chunks"] = take(16 * qd).reshape(qd, 4, 4)
        R = int(take(1)[0])
        d["commit_phase_commits"] = (take(8 * R).reshape(R, 8) if version == 1
                                     else take(8 * nr * R).reshape(R, nr, 8))
        Q = int(take(1)[0])
        for _ in range(Q):
            nb = int(take(1)[0])
            batches = []
            for _ in range(nb):
                nm = int(take(1)[0])
                vals = [take(int(take(1)[0])) for _ in range(nm)]
                plen = int(take(1)[0])
                batches.append(BatchOpening(vals, take(8 * plen).reshape(plen, 8)))
            steps = []
            for _ in range(R):
                vals = take(8).reshape(2, 4)
                plen = int(take(1)[0])
                steps.append((vals, take(8 * plen).reshape(plen, 8)))
            d["query_proofs"].append(QueryProof(batches, steps))
        d["final_poly"] = take(4)
        d["pow_witness"] = int(take(1)[0])
        if pos != len(w):
            raise ValueError("trailing words in proof")
        self.__dict__.update(d)


def prove(config: StarkConfig, air, challenger: BfChallenger, trace, public_values) -> Proof:
    """``uni_stark::prove`` (reference uni-stark/src/prover.rs:25-35).

    ``air`` is a ``BaseAir`` (captured symbolically like ``get_symbolic_constraints``) or an
    already ``CompiledAir``; ``trace`` an (n, w) array or a ``DeviceMatrix`` (consumed).
    """
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
    R = log_N - pcs.fri.log_blowup
    Q = pcs.fri.num_queries
    cap = (64 + 8 * w + 16 * qd + 8 * R
           + Q * (16 + w + 5 * qd + 2 * 8 * log_N + R * (9 + 8 * log_N)))
    # One output buffer per context, kept for its lifetime (a context is driven by one thread): no
    # half-megabyte allocation, zero-fill and release per proof.
    out = getattr(ctx, "_proof_buf", None)
    if out is None or len(out) < cap:
        out = ctx._proof_buf = np.zeros(cap, dtype=np.uint32)
    n_words = C.c_size_t()
    cfg = pcs.fri._c()
    pis_p = _p(pis) if len(pis) else None
    ctx.check(ctx._l.ts_prove(ctx.h, C.byref(cfg), air.h, challenger.h, trace.h, pis_p, len(pis),
                              _p(out), len(out), C.byref(n_words)))
    return Proof(out[: n_words.value].copy())


def prove_stream(lanes, traces, lane_of, public_values, gate_ms: float = 0.0, want_times: bool = True):
    """``ts_prove_stream``: ``len(traces)`` independent proofs on the contexts of ``lanes`` = [(StarkConfig,
    CompiledAir), ...] (one context each, same device, same FriConfig), one host thread per lane INSIDE the
    library; proof i runs on lane ``lane_of[i]`` with ``traces[i]`` (a DeviceMatrix made on that lane's context;
    consumed) and a fresh challenger.  Returns (Proof of the highest index, start_ms array, wall_ms array)."""
    n_l, n = len(lanes), len(traces)
    l = _lib.lib()
    ctxs = (C.c_void_p * n_l)(*[conf.pcs.ctx.h for conf, _ in lanes])
    airs = (C.c_void_p * n_l)(*[a.h for _, a in lanes])
    mats = (C.c_void_p * max(n, 1))(*[t.h for t in traces])
    lo = _u32(lane_of)
    pis = _u32(public_values)
    pcs = lanes[0][0].pcs
    cfg = pcs.fri._c()
    ctx0 = pcs.ctx
    out = getattr(ctx0, "_proof_buf", None)
    if out is None or len(out) < (1 << 20):
        out = ctx0._proof_buf = np.zeros(1 << 20, dtype=np.uint32)
    n_words = C.c_size_t()
    st = np.zeros(max(n, 1), dtype=np.float64)
    wl = np.zeros(max(n, 1), dtype=np.float64)
    dp = C.POINTER(C.c_double)
    rc = l.ts_prove_stream(ctxs, airs, n_l, C.byref(cfg), mats, _p(lo), n, _p(pis) if len(pis) else None, len(pis),
                           float(gate_ms), _p(out), len(out), C.byref(n_words),
                           st.ctypes.data_as(dp) if want_times else None, wl.ctypes.data_as(dp) if want_times else None)
    if rc:
        for conf, _ in lanes:  # the failing lane's context holds the message
            msg = (l.ts_last_error(conf.pcs.ctx.h) or b"").decode()
            if msg:
                raise _lib.TsError(rc, msg)
        raise _lib.TsError(rc, "ts_prove_stream")
    return Proof(out[: n_words.value].copy()), st[:n], wl[:n]


def prove_sharded(config: StarkConfig, air, challenger: BfChallenger, trace_rows, public_values,
                  comm, min_local_log: int = 0, trace_replicated: bool = False,
                  local_quotient: bool = False, _options_struct_size: int | None = None) -> Proof:
    """One proof over ``comm.world`` GPUs (``ts_prove_sharded``).

    Every rank calls this with its own context, a challenger in the same state and its row slice
    ``trace_rows`` = natural rows [g n/G, (g+1) n/G) of the trace; every rank gets the whole proof,
    bit-identical to :func:`prove` on the whole trace.  ``comm`` is a ``dist.TorchComm``.
    ``trace_replicated``: ``trace_rows`` is the whole trace on every rank (no all-gather of it).
    ``local_quotient``: every rank computes the quotient on its own cosets (no chunk broadcast; for a
    trace that violates its constraints all ranks fall back to the broadcast path, so the proof is
    :func:`prove`'s for every trace -- ``ts_shard_options.local_quotient`` in include/tapstark.h).
    """
    pcs = config.pcs
    ctx = pcs.ctx
    pis = _u32(public_values)
    if isinstance(air, BaseAir):
        air = CompiledAir(ctx, air_tape(air, len(pis)))
    if not isinstance(trace_rows, DeviceMatrix):
        trace_rows = DeviceMatrix.upload(ctx, trace_rows)
    n_loc, w = trace_rows.dims()
    n = n_loc if trace_replicated else n_loc * comm.world
    log_N = n.bit_length() - 1 + pcs.fri.log_blowup
    qd = 1 << air.log_quotient_degree
    R = log_N - pcs.fri.log_blowup
    Q = pcs.fri.num_queries
    cap = (64 + 8 * w + 16 * qd + 8 * R
           + Q * (16 + w + 5 * qd + 2 * 8 * log_N + R * (9 + 8 * log_N)))
    out = np.zeros(cap, dtype=np.uint32)
    n_words = C.c_size_t()
    cfg = pcs.fri._c()
    pis_p = _p(pis) if len(pis) else None
    # (_options_struct_size: test hook -- a caller built against another layout of ts_shard_options)
    opts = _lib.ShardOptionsC(C.sizeof(_lib.ShardOptionsC) if _options_struct_size is None else _options_struct_size,
                              min_local_log, int(trace_replicated), int(local_quotient))
    rc = ctx._l.ts_prove_sharded(ctx.h, C.byref(cfg), C.byref(comm.c), air.h, challenger.h,
                                 trace_rows.h, pis_p, len(pis), C.byref(opts), _p(out), cap,
                                 C.byref(n_words))
    if rc == 7 and getattr(comm, "error", None):
        raise RuntimeError("communicator callback failed:\n" + comm.error)
    ctx.check(rc)
    return Proof(out[: n_words.value].copy())


VERIFY_ERRORS = {0: "Ok", 1: "InvalidProofShape", 2: "InvalidOpeningArgument(InvalidProofShape)",
                 3: "InvalidOpeningArgument(InvalidPowWitness)", 4: "InvalidOpeningArgument(InputError)",
                 5: "InvalidOpeningArgument(CommitPhaseMmcsError)",
                 6: "InvalidOpeningArgument(FinalPolyMismatch)", 7: "OodEvaluationMismatch",
                 8: "folded evaluation mismatch", 9: "malformed proof"}


class VerificationError(Exception):
    """reference uni-stark/src/verifier.rs:163-171 / fri/src/error.rs:21-29"""

    def __init__(self, code: int):
        super().__init__(VERIFY_ERRORS.get(code, str(code)))
        self.code = code


def verify(config: StarkConfig, air, challenger: BfChallenger, proof, public_values) -> None:
    """``uni_stark::verify`` (reference uni-stark/src/verifier.rs:19-25); host only, no GPU.
    Raises ``VerificationError``; returns None on acceptance (``Ok(())``)."""
    pis = _u32(public_values)
    if isinstance(air, BaseAir):
        air = CompiledAir(None, air_tape(air, len(pis)))
    words = _u32(proof.words if isinstance(proof, Proof) else proof)
    cfg = config.pcs.fri._c()
    verdict = C.c_int(-1)
    l = _lib.lib()
    rc = l.ts_verify(C.byref(cfg), air.h, challenger.h, _p(words), len(words),
                     _p(pis) if len(pis) else None, len(pis), C.byref(verdict))
    if rc:
        raise _lib.TsError(rc, "ts_verify")
    if verdict.value != 0:
        raise VerificationError(verdict.value)


def check_constraints(air, trace, public_values, ctx: Context | None = None) -> int:
    """reference uni-stark/src/check_constraints.rs:11-39 on the GPU.  Returns -1 if every
    constraint holds on every row, else ``row * 65536 + constraint_index`` of the first failure."""
    ctx = ctx or default_context()
    pis = _u32(public_values)
    if isinstance(air, BaseAir):
        air = CompiledAir(ctx, air_tape(air, len(pis)))
    if not isinstance(trace, DeviceMatrix):
        trace = DeviceMatrix.upload(ctx, trace)
    out = C.c_int64(-1)
    ctx.check(ctx._l.ts_check_constraints(ctx.h, air.h, trace.h, _p(pis) if len(pis) else None,
                                          len(pis), C.byref(out)))
    return int(out.value)
