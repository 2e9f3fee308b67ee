"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

u32p = C.POINTER(C.c_uint32)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class FriConfig(C.Structure):
    _fields_ = [("log_blowup", C.c_uint32), ("num_queries", C.c_uint32),
                ("proof_of_work_bits", C.c_uint32)]


class Challenger(C.Structure):
    _fields_ = [("state", C.c_uint32 * 16), ("in_buf", C.c_uint32 * 8), ("n_in", C.c_int),
                ("out_buf", C.c_uint32 * 8), ("n_out", C.c_int), ("perm_kind", C.c_int),
                ("sample_ext", C.c_int), ("n_perms", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ts_or_prove.restype = C.c_int64
        _lib.ts_or_check_constraints.restype = C.c_int64
        _lib.ts_or_chal_sample_bits.restype = C.c_uint64
        _lib.ts_or_chal_sample_base.restype = C.c_uint32
        _lib.ts_or_last_transcript.restype = C.c_size_t
        _lib.ts_or_mmcs_commit.restype = C.c_void_p
        _lib.ts_or_mmcs_layer.restype = u32p
        _lib.ts_or_eval_interpolant_naive.restype = C.c_uint32
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


# ---------------------------------------------------------------- primitives
def blake3(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().ts_or_blake3(data, C.c_size_t(len(data)), out)
    return bytes(out)


class OracleChallenger:
    """reference basic/src/challenger/mod.rs BfChallenger<F, U32, P, 16>."""

    def __init__(self, perm_kind: int = 0, sample_ext: bool = True):
        self.c = Challenger()
        lib().ts_or_chal_init(C.byref(self.c), perm_kind, int(sample_ext))

    def observe(self, word: int):
        lib().ts_or_chal_observe(C.byref(self.c), C.c_uint32(word))

    def observe_digest(self, d):
        d = _u32(d)
        lib().ts_or_chal_observe_digest(C.byref(self.c), _p(d))

    def sample_base(self) -> int:
        return int(lib().ts_or_chal_sample_base(C.byref(self.c)))

    def sample(self) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint32)
        lib().ts_or_chal_sample(C.byref(self.c), _p(out))
        return out

    def sample_bits(self, bits: int) -> int:
        return int(lib().ts_or_chal_sample_bits(C.byref(self.c), C.c_uint(bits)))

    def grind(self, bits: int) -> int:
        w = C.c_uint32()
        rc = lib().ts_or_chal_grind(C.byref(self.c), C.c_uint(bits), C.byref(w))
        if rc:
            raise RuntimeError("failed to find witness")
        return int(w.value)

    def check_witness(self, bits: int, witness: int) -> bool:
        return bool(lib().ts_or_chal_check_witness(C.byref(self.c), C.c_uint(bits),
                                                   C.c_uint32(witness)))

    @property
    def n_perms(self) -> int:
        return int(self.c.n_perms)

    def state_words(self) -> np.ndarray:
        """(state[16], n_in, in_buf[8], n_out, out_buf[8]) flattened, for state comparison."""
        return np.array(list(self.c.state) + [self.c.n_in] + list(self.c.in_buf)[: self.c.n_in]
                        + [self.c.n_out] + list(self.c.out_buf)[: self.c.n_out], dtype=np.uint64)


# ------------------------------------------------------------------- DFT/LDE
def naive_dft(m: np.ndarray, inverse: bool = False) -> np.ndarray:
    m = _u32(m)
    n, w = m.shape
    out = np.zeros_like(m)
    lib().ts_or_naive_dft(_p(m), _p(out), C.c_size_t(n), C.c_size_t(w), int(inverse))
    return out


def dft_batch(m: np.ndarray, inverse: bool = False) -> np.ndarray:
    m = _u32(m).copy()
    n, w = m.shape
    lib().ts_or_dft_batch(_p(m), C.c_size_t(n), C.c_size_t(w), int(inverse))
    return m


def coset_lde_batch(m: np.ndarray, added_bits: int, shift: int) -> np.ndarray:
    m = _u32(m)
    n, w = m.shape
    out = np.zeros((n << added_bits, w), dtype=np.uint32)
    lib().ts_or_coset_lde_batch(_p(m), C.c_size_t(n), C.c_size_t(w), C.c_uint(added_bits),
                                C.c_uint32(shift), _p(out))
    return out


def commit_lde(m: np.ndarray, domain_shift: int, log_blowup: int) -> np.ndarray:
    """Pcs::commit's LDE: (N, w), bit-reversed rows."""
    m = _u32(m)
    n, w = m.shape
    log_n = n.bit_length() - 1
    out = np.zeros((n << log_blowup, w), dtype=np.uint32)
    lib().ts_or_commit_lde(_p(m), C.c_uint(log_n), C.c_size_t(w), C.c_uint32(domain_shift),
                           C.c_uint(log_blowup), _p(out))
    return out


def eval_interpolant_naive(col: np.ndarray, domain_shift: int, x: int) -> int:
    col = _u32(col)
    return int(lib().ts_or_eval_interpolant_naive(_p(col), C.c_size_t(len(col)),
                                                  C.c_uint32(domain_shift), C.c_uint32(x)))


# ---------------------------------------------------------------------- MMCS
class OracleMmcs:
    def __init__(self, mats):
        self.mats = [_u32(m) for m in mats]
        n = len(self.mats)
        ptrs = (u32p * n)(*[_p(m) for m in self.mats])
        self.heights = (C.c_size_t * n)(*[m.shape[0] for m in self.mats])
        self.widths = (C.c_size_t * n)(*[m.shape[1] for m in self.mats])
        self.root = np.zeros(8, dtype=np.uint32)
        self.h = C.c_void_p(lib().ts_or_mmcs_commit(n, ptrs, self.heights, self.widths,
                                                    _p(self.root)))
        self.log_max_h = int(lib().ts_or_mmcs_log_max_height(self.h))

    def open(self, index: int):
        tot = sum(m.shape[1] for m in self.mats)
        rows = np.zeros(tot, dtype=np.uint32)
        path = np.zeros((max(self.log_max_h, 1), 8), dtype=np.uint32)
        lib().ts_or_mmcs_open(self.h, C.c_size_t(index), _p(rows), _p(path))
        return rows, path[: self.log_max_h]

    def layer(self, level: int) -> np.ndarray:
        ptr = lib().ts_or_mmcs_layer(self.h, C.c_uint(level))
        cnt = (1 << self.log_max_h) >> level
        return np.ctypeslib.as_array(ptr, shape=(cnt, 8)).copy()

    def verify(self, index: int, rows, path, root=None) -> bool:
        rows, path = _u32(rows), _u32(path)
        root = self.root if root is None else _u32(root)
        return bool(lib().ts_or_mmcs_verify(len(self.mats), self.heights, self.widths,
                                            C.c_size_t(index), _p(rows), _p(path),
                                            C.c_size_t(path.shape[0]), _p(root)))

    def __del__(self):
        try:
            lib().ts_or_mmcs_free(self.h)
        except Exception:
            pass


# ----------------------------------------------------------------------- AIR
def tape_validate(tape) -> int:
    tape = _u32(tape)
    return int(lib().ts_or_tape_validate(_p(tape), C.c_size_t(len(tape))))


def max_constraint_degree(tape) -> int:
    tape = _u32(tape)
    return int(lib().ts_or_air_max_constraint_degree(_p(tape), C.c_size_t(len(tape))))


def log_quotient_degree(tape) -> int:
    tape = _u32(tape)
    return int(lib().ts_or_air_log_quotient_degree(_p(tape), C.c_size_t(len(tape))))


def constraint_values(tape, local, nxt, pis, sels) -> np.ndarray:
    """(m, n_constraints): every constraint of the tape on m (local row, next row, selector triple)
    inputs, by direct evaluation of the DAG."""
    tape, local, nxt, pis, sels = _u32(tape), _u32(local), _u32(nxt), _u32(pis), _u32(sels)
    if len(pis) == 0:
        pis = np.zeros(1, dtype=np.uint32)
    m, k = local.shape[0], int(tape[5])
    out = np.zeros((m, k), dtype=np.uint32)
    rc = lib().ts_or_tape_constraint_values(_p(tape), C.c_size_t(len(tape)), _p(local), _p(nxt), C.c_size_t(m),
                                            _p(pis), _p(sels), _p(out))
    assert rc == 0, "malformed tape"
    return out


def check_constraints(tape, trace, pis) -> int:
    tape, trace, pis = _u32(tape), _u32(trace), _u32(pis)
    if len(pis) == 0:
        pis = np.zeros(1, dtype=np.uint32)
    return int(lib().ts_or_check_constraints(_p(tape), C.c_size_t(len(tape)), _p(trace),
                                             C.c_size_t(trace.shape[0]), _p(pis)))


# -------------------------------------------------------------------- stages
def quotient_values(tape, lde, log_n, log_blowup, pis, alpha) -> np.ndarray:
    tape, lde, pis, alpha = _u32(tape), _u32(lde), _u32(pis), _u32(alpha)
    if len(pis) == 0:
        pis = np.zeros(1, dtype=np.uint32)
    lqd = log_quotient_degree(tape)
    out = np.zeros(((1 << log_n) << lqd, 4), dtype=np.uint32)
    lib().ts_or_quotient_values(_p(tape), C.c_size_t(len(tape)), _p(lde), C.c_uint(log_n),
                                C.c_uint(log_blowup), _p(pis), _p(alpha), _p(out))
    return out


def split_quotient(qvals, log_n, log_qd) -> np.ndarray:
    qvals = _u32(qvals)
    out = np.zeros((1 << log_qd, 1 << log_n, 4), dtype=np.uint32)
    lib().ts_or_split_quotient(_p(qvals), C.c_uint(log_n), C.c_uint(log_qd), _p(out))
    return out


def fold_matrix(vec, beta) -> np.ndarray:
    vec, beta = _u32(vec), _u32(beta)
    h = vec.shape[0] // 2
    out = np.zeros((h, 4), dtype=np.uint32)
    lib().ts_or_fold_matrix(_p(vec), C.c_size_t(h), _p(beta), _p(out))
    return out


def fold_row(index, log_height, beta, e0, e1) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint32)
    lib().ts_or_fold_row(C.c_size_t(index), C.c_uint(log_height), _p(_u32(beta)), _p(_u32(e0)),
                         _p(_u32(e1)), _p(out))
    return out


def open_reduce(trace_lde, chunk_ldes, log_n, log_blowup, zeta, alpha):
    trace_lde = _u32(trace_lde)
    chunk_ldes = [_u32(c) for c in chunk_ldes]
    qd = len(chunk_ldes)
    log_qd = qd.bit_length() - 1
    w = trace_lde.shape[1]
    N = trace_lde.shape[0]
    ptrs = (u32p * qd)(*[_p(c) for c in chunk_ldes])
    opened = np.zeros((2 * w + 4 * qd, 4), dtype=np.uint32)
    ro = np.zeros((N, 4), dtype=np.uint32)
    lib().ts_or_open_reduce(_p(trace_lde), C.c_size_t(w), ptrs, C.c_uint(log_qd), C.c_uint(log_n),
                            C.c_uint(log_blowup), _p(_u32(zeta)), _p(_u32(alpha)), _p(opened),
                            _p(ro))
    return opened, ro


# --------------------------------------------------------------- whole proofs
def prove(cfg: FriConfig, tape, trace, pis, chal: OracleChallenger | None = None,
          cap_words: int = 1 << 24, debug_assertions: bool = True) -> np.ndarray:
    """debug_assertions=False: a release build of the reference (prover.rs:40-41 compiles
    check_constraints out), which proves an invalid trace without complaint."""
    tape, trace, pis = _u32(tape), _u32(trace), _u32(pis)
    if len(pis) == 0:
        pis = np.zeros(1, dtype=np.uint32)
    chal = chal or OracleChallenger()
    log_n = trace.shape[0].bit_length() - 1
    out = np.zeros(cap_words, dtype=np.uint32)
    lib().ts_or_set_debug_assertions(1 if debug_assertions else 0)
    try:
        n = lib().ts_or_prove(C.byref(cfg), _p(tape), C.c_size_t(len(tape)), C.byref(chal.c),
                              _p(trace), C.c_uint(log_n), _p(pis), _p(out), C.c_size_t(cap_words))
    finally:
        lib().ts_or_set_debug_assertions(1)
    if n < 0:
        raise RuntimeError(f"oracle prove failed: {n}")
    return out[:n].copy()


def verify(cfg: FriConfig, tape, proof, pis, chal: OracleChallenger | None = None) -> int:
    tape, proof, pis = _u32(tape), _u32(proof), _u32(pis)
    if len(pis) == 0:
        pis = np.zeros(1, dtype=np.uint32)
    chal = chal or OracleChallenger()
    return int(lib().ts_or_verify(C.byref(cfg), _p(tape), C.c_size_t(len(tape)),
                                  C.byref(chal.c), _p(proof), C.c_size_t(len(proof)), _p(pis)))


def last_transcript() -> dict:
    buf = np.zeros(1024, dtype=np.uint32)
    n = lib().ts_or_last_transcript(_p(buf), C.c_size_t(len(buf)))
    buf = buf[:n]
    R = int(buf[12])
    betas = buf[13:13 + 4 * R].reshape(R, 4)
    pos = 13 + 4 * R
    pow_witness = int(buf[pos])
    nq = int(buf[pos + 1])
    return {"alpha": buf[0:4].copy(), "zeta": buf[4:8].copy(), "batch_alpha": buf[8:12].copy(),
            "betas": betas.copy(), "pow_witness": pow_witness,
            "indices": buf[pos + 2:pos + 2 + nq].copy()}


def pcs_roundtrip(cfg: FriConfig, log_degrees_by_round, evals_by_round, tamper: int = 0) -> int:
    flat_logs, flat_w, flat_e, per_round = [], [], [], []
    for logs, evs in zip(log_degrees_by_round, evals_by_round):
        per_round.append(len(logs))
        for lg, e in zip(logs, evs):
            e = _u32(e)
            flat_logs.append(lg)
            flat_w.append(e.shape[1])
            flat_e.append(e)
    k = len(flat_e)
    return int(lib().ts_or_pcs_roundtrip(
        C.byref(cfg), len(per_round), (C.c_int * len(per_round))(*per_round),
        (C.c_uint * k)(*flat_logs), (C.c_size_t * k)(*flat_w),
        (u32p * k)(*[_p(e) for e in flat_e]), tamper))


def fri_roundtrip(cfg: FriConfig, inputs, sample_ext: bool = True, perm_kind: int = 1) -> int:
    inputs = [_u32(v) for v in inputs]
    k = len(inputs)
    logs = [(v.shape[0]).bit_length() - 1 for v in inputs]
    return int(lib().ts_or_fri_roundtrip(C.byref(cfg), k, (C.c_uint * k)(*logs),
                                         (u32p * k)(*[_p(v) for v in inputs]), int(sample_ext),
                                         perm_kind))


def pcs_commit_open(cfg: FriConfig, log_degrees_by_round, evals_by_round,
                    chal: OracleChallenger | None = None, multi: bool = False):
    """fri/tests/pcs.rs:62-90 flow on the oracle: returns (roots, zeta, opened (sum_w x 4), proof)."""
    flat_logs, flat_w, flat_e, per_round = [], [], [], []
    for logs, evs in zip(log_degrees_by_round, evals_by_round):
        per_round.append(len(logs))
        for lg, e in zip(logs, evs):
            e = _u32(e)
            flat_logs.append(lg)
            flat_w.append(e.shape[1])
            flat_e.append(e)
    k = len(flat_e)
    chal = chal or OracleChallenger()
    roots = np.zeros((len(per_round), 8), dtype=np.uint32)
    zeta = np.zeros(4, dtype=np.uint32)
    opened = np.zeros((sum(w * (1 + i % 3 if multi else 1) for i, w in enumerate(flat_w)), 4),
                      dtype=np.uint32)
    cap = 1 << 22
    proof = np.zeros(cap, dtype=np.uint32)
    fn = lib().ts_or_pcs_commit_open_multi if multi else lib().ts_or_pcs_commit_open
    fn.restype = C.c_int64
    n = fn(
        C.byref(cfg), C.byref(chal.c), len(per_round), (C.c_int * len(per_round))(*per_round),
        (C.c_uint * k)(*flat_logs), (C.c_size_t * k)(*flat_w), (u32p * k)(*[_p(e) for e in flat_e]),
        _p(roots), _p(zeta), _p(opened), _p(proof), C.c_size_t(cap))
    if n < 0:
        raise RuntimeError(f"oracle pcs_commit_open failed: {n}")
    return roots, zeta, opened, proof[:n].copy()


def fri_prove(cfg: FriConfig, inputs, chal: OracleChallenger) -> np.ndarray:
    """fri/tests/fri.rs: bf_prove over EF4 vectors (descending lengths), pass-through input proof."""
    inputs = [_u32(v) for v in inputs]
    k = len(inputs)
    logs = [(v.shape[0]).bit_length() - 1 for v in inputs]
    cap = 1 << 22
    out = np.zeros(cap, dtype=np.uint32)
    lib().ts_or_fri_prove.restype = C.c_int64
    n = lib().ts_or_fri_prove(C.byref(cfg), C.byref(chal.c), k, (C.c_uint * k)(*logs),
                              (u32p * k)(*[_p(v) for v in inputs]), _p(out), C.c_size_t(cap))
    if n < 0:
        raise RuntimeError(f"oracle fri_prove failed: {n}")
    return out[:n].copy()


def fri_verify(cfg: FriConfig, proof, chal: OracleChallenger) -> int:
    proof = _u32(proof)
    return int(lib().ts_or_fri_verify(C.byref(cfg), C.byref(chal.c), _p(proof), C.c_size_t(len(proof))))


# ---------------------------------------------------------------- taptree commitment (taptree.c)
def sha256(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().ts_or_sha256(data, C.c_size_t(len(data)), out)
    return bytes(out)


def tagged_hash(tag: str, msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().ts_or_tagged_hash(tag.encode(), msg, C.c_size_t(len(msg)), out)
    return bytes(out)


def tapleaf_hash(script: bytes, version: int = 0xC0) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().ts_or_tapleaf_hash(script, C.c_size_t(len(script)), version, out)
    return bytes(out)


def tapbranch(a: bytes, b: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().ts_or_tapbranch(a, b, out)
    return bytes(out)


def script_push_int(v: int) -> bytes:
    out = (C.c_uint8 * 16)()
    lib().ts_or_script_push_int.restype = C.c_size_t
    n = lib().ts_or_script_push_int(C.c_uint64(v), out)
    return bytes(out[:n])


def _locks_c(locks):
    arr = (C.c_char_p * len(locks))(*locks)
    lens = (C.c_size_t * len(locks))(*[len(x) for x in locks])
    return arr, lens


def tap_leaf_script(locks, index: int, values, u32_size: int = 1) -> bytes:
    vals = _u32(values).reshape(-1)
    arr, lens = _locks_c(locks)
    fn = lib().ts_or_tap_leaf_script
    fn.restype = C.c_size_t
    n_evals = len(locks) - 1
    n = fn(arr, lens, C.c_uint64(index), _p(vals), n_evals, u32_size, None, C.c_size_t(0))
    out = (C.c_uint8 * max(n, 1))()
    fn(arr, lens, C.c_uint64(index), _p(vals), n_evals, u32_size, out, C.c_size_t(n))
    return bytes(out[:n])


def padding_matrix(mats) -> np.ndarray:
    mats = [_u32(m) for m in mats]
    n = len(mats)
    ptrs = (u32p * n)(*[_p(m) for m in mats])
    hs = (C.c_size_t * n)(*[m.shape[0] for m in mats])
    ws = (C.c_size_t * n)(*[m.shape[1] for m in mats])
    out = np.zeros((max(m.shape[0] for m in mats), sum(m.shape[1] for m in mats)), dtype=np.uint32)
    lib().ts_or_padding_matrix(n, ptrs, hs, ws, _p(out))
    return out


class OracleTaptree:
    """build_tree (builder.rs:38-93) over leaf hashes: every level + the index dictionary."""

    def __init__(self, leaf_hashes):
        n = len(leaf_hashes)
        self.n = n
        self.nodes = (C.c_uint8 * (32 * (2 * n - 1)))()
        self.leaf_indices = (C.c_size_t * n)()
        lib().ts_or_taptree_build(C.c_size_t(n), b"".join(leaf_hashes), self.nodes, self.leaf_indices)
        nb = bytes(self.nodes)
        self.root = nb[-32:]
        self.levels, off, cnt = [], 0, n
        while cnt >= 1:
            self.levels.append([nb[32 * (off + i):32 * (off + i) + 32] for i in range(cnt)])
            off += cnt
            cnt //= 2

    @classmethod
    def from_scripts(cls, scripts):
        return cls([tapleaf_hash(s) for s in scripts])

    def path(self, index: int):
        depth = self.n.bit_length() - 1
        out = (C.c_uint8 * max(32 * depth, 1))()
        lib().ts_or_taptree_path(C.c_size_t(self.n), self.nodes, C.c_size_t(index), out)
        ob = bytes(out)
        return [ob[32 * k:32 * k + 32] for k in range(depth)]


def taptree_verify_inclusion(root: bytes, leaf: bytes, path) -> bool:
    return bool(lib().ts_or_taptree_verify_inclusion(root, leaf, b"".join(path) or b"\0", C.c_size_t(len(path))))


def tap_commit_polys(mats, locks, u32_size: int = 1) -> OracleTaptree:
    """commit_polys (tcs/mod.rs:238-282) for one tree."""
    mats = [_u32(m) for m in mats]
    ys = padding_matrix(mats)
    n_evals = ys.shape[1] // u32_size
    assert len(locks) == 1 + n_evals
    return OracleTaptree.from_scripts([tap_leaf_script(locks, i, ys[i], u32_size) for i in range(ys.shape[0])])


# ------------------------------------------------- whole proofs over the taptree MMCS (TSPF v2)
class _TapMode:
    """Context manager: the oracle's MMCS speaks taptree (mmcs.c) with `locks` = the flat table of
    lock scripts in commit order -- trace: Q (1 + w); quotient chunks: Q (1 + 4 qd); every FRI round:
    Q (1 + 2) -- while the block runs."""

    def __init__(self, num_queries: int, locks):
        self.q = num_queries
        self.blob = b"".join(locks)
        self.offs = np.zeros(len(locks) + 1, dtype=np.uint64)
        self.offs[1:] = np.cumsum([len(x) for x in locks])

    def __enter__(self):
        lib().ts_or_mmcs_set_taptree(self.q, self.blob, self.offs.ctypes.data_as(C.POINTER(C.c_uint64)))
        return self

    def __exit__(self, *a):
        lib().ts_or_mmcs_set_taptree(0, None, None)


def prove_tap(cfg: FriConfig, tape, trace, pis, locks, chal: OracleChallenger | None = None,
              cap_words: int = 1 << 24) -> np.ndarray:
    """prove() with `TapTreeMmcs` (basic/src/mmcs/taptree_mmcs.rs) as both MMCSs: TSPF v2 words."""
    with _TapMode(cfg.num_queries, locks):
        return prove(cfg, tape, trace, pis, chal, cap_words)


def verify_tap(cfg: FriConfig, tape, proof, pis, locks, chal: OracleChallenger | None = None) -> int:
    with _TapMode(cfg.num_queries, locks):
        return verify(cfg, tape, proof, pis, chal)
