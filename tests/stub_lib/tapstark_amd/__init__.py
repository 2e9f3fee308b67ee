"""STUB of the tapstark_amd package for bench.py's protocol dry-run on the CPU (TEST INFRASTRUCTURE:
tests/test_bench_dist_cpu.py puts this directory first on sys.path through TS_BENCH_STUB_LIB).

bench.py then runs its REAL code path -- lanes, start gate, priming probes, timed windows, record
building, the rank-0 legs and the sharded config-4 / config-5 blocks with their communicator branch --
against objects that only sleep and hand out fake proofs.  Nothing here computes a proof; the numbers in
the resulting line are meaningless except for their ARITHMETIC (value = cells of all ranks / max time),
which is what the test checks before the first real 8-GPU lease.  The real airs / benchutil modules are
loaded from the product package (pure Python)."""
import importlib.util
import os
import sys
import time

import numpy as np

_REAL = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))),
                     "tap-stark_amd")


def _load(name):
    spec = importlib.util.spec_from_file_location(f"tapstark_amd.{name}", os.path.join(_REAL, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"tapstark_amd.{name}"] = mod
    spec.loader.exec_module(mod)
    return mod


air = _load("air")
airs = _load("airs")
benchutil = _load("benchutil")
from .air import air_tape  # noqa: E402

STEP_S = float(os.environ.get("TS_STUB_STEP_S", "0.004"))


class _Lib:
    LIB_PATH = __file__  # "exists"


_lib = _Lib()


class Context:
    num_cus = 256

    def __init__(self, device=0):
        self.device = device
        self._timing = self._ktiming = False
        self._stages, self._kernels = [], {}

    def synchronize(self):
        pass

    def set_timing(self, on):
        self._timing = on

    def take_timings(self):
        s, self._stages = self._stages, []
        return s

    def set_kernel_timing(self, on):
        self._ktiming = on

    def take_kernel_timings(self):
        k, self._kernels = self._kernels, {}
        return k

    def alu_ceiling(self, kind):
        time.sleep(0.001)
        return 1e12

    def bench_stage(self, *a):
        return 1.0

    def stat(self, which):
        return 0

    def graph_stats(self):
        return {"replays": 0, "fallbacks": 0, "shapes": 0, "pool_bytes": 0, "reserve_failures": 0}


def default_context():
    return Context(0)


class FriConfig:
    def __init__(self, log_blowup, num_queries, proof_of_work_bits):
        self.log_blowup, self.num_queries, self.proof_of_work_bits = log_blowup, num_queries, proof_of_work_bits


class TwoAdicFriPcs:
    def __init__(self, fri, ctx=None, host_only=False):
        self.fri, self.ctx = fri, ctx


class StarkConfig:
    def __init__(self, pcs):
        self.pcs = pcs


class CompiledAir:
    is_jit = True

    def __init__(self, ctx, tape):
        self.ctx, self.tape = ctx, tape
        self.log_quotient_degree = 1
        self.width = int(tape[2])


class BfChallenger:
    pass


class DeviceMatrix:
    def __init__(self, ctx, h, w):
        self.ctx, self.h, self.w = ctx, h, w

    @classmethod
    def synth_mul(cls, ctx, n, width=64, seed=None):
        return cls(ctx, n, width)

    synth_ext = synth_mul

    @classmethod
    def fibonacci(cls, ctx, a, b, n):
        return cls(ctx, n, 2)

    @classmethod
    def upload(cls, ctx, values):
        return cls(ctx, values.shape[0], values.shape[1])

    @classmethod
    def upload_async(cls, ctx, pinned):
        return cls(ctx, *pinned.array.shape)

    def download(self):
        return np.zeros((min(self.h, 4), self.w), dtype=np.uint32)


class PinnedHostMatrix:
    def __init__(self, h, w):
        self.array = np.zeros((min(h, 4), w), dtype=np.uint32)


class Proof:
    def __init__(self, words):
        self.words = words
        self.degree_bits = 20
        self.pow_witness = 0


_STAGES = ["coset_lde", "merkle_commit", "compute quotient polynomial", "coset_lde", "merkle_commit",
           "compute opened values with Lagrange interpolation", "reduce rows", "FRI commit phase",
           "grind for proof-of-work witness", "query phase", "prove"]
_KERNELS = ["k_leaf_tree<2,strided>", "k_lde_fwd_contig<14, 4>", "k_lde_mid<1, 8192, 512>", "k_intt_contig<true>",
            "k_quotient_jit", "k_fri_round", "k_merkle_tree"]


def _fake_proof(ctx):
    time.sleep(STEP_S)
    if ctx._timing:
        ctx._stages += [(s, 0.1) for s in _STAGES]
    if ctx._ktiming:
        for k in _KERNELS:
            c, ms = ctx._kernels.get(k, (0, 0.0))
            ctx._kernels[k] = (c + 2, ms + 0.2)
    return Proof(np.arange(64, dtype=np.uint32))


def prove(config, air_, challenger, trace, public_values):
    return _fake_proof(config.pcs.ctx)


def prove_stream(lanes, traces, lane_of, public_values, gate_ms=0.0, want_times=True):
    """Stand-in for ts_prove_stream: one thread per lane, each 'proof' a sleep."""
    import threading

    n = len(traces)
    start, wall = np.zeros(n), np.zeros(n)
    t_begin = time.perf_counter()

    def lane(l):
        for i in range(n):
            if lane_of[i] == l:
                t0 = time.perf_counter()
                _fake_proof(lanes[l][0].pcs.ctx)
                start[i], wall[i] = 1e3 * (t0 - t_begin), 1e3 * (time.perf_counter() - t0)

    th = [threading.Thread(target=lane, args=(l,)) for l in range(len(lanes))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return Proof(np.arange(64, dtype=np.uint32) if n else np.zeros(0, dtype=np.uint32)), start, wall


def prove_sharded(config, air_, challenger, trace_rows, public_values, comm, min_local_log=0,
                  trace_replicated=False, local_quotient=False):
    ctx = config.pcs.ctx
    if ctx._timing:  # one collective of each kind on the record, like csrc/sharded.cpp leaves them
        ctx._stages += [(f"collective: all_gather 32 B/rank (commit sub-roots)", 0.01),
                        (f"collective: broadcast 1024 B (opened values)", 0.01)]
    comm.exchange()
    return _fake_proof(ctx)


def verify(config, air_, challenger, proof, public_values):
    return None
