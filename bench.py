#!/usr/bin/env python3
"""bench.py -- trace cells/sec (and proofs/sec) of the MI355X STARK prover hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete prove() (commit trace -> quotient -> commit chunks -> open -> FRI ->
queries -> proof on the host) of BASELINE.json configs[2]: the build-defined SynthMulAir-64 trace,
2^20 rows x 64 columns, log_blowup 2, 28 queries, 8 PoW bits, with the trace already resident in
HBM when the timed region starts.

N > 1: one rank per GPU.  Launched by `python -m torch.distributed.run` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) the process IS a rank; launched as a plain command with
WORLD_SIZE unset it spawns its N ranks itself as fresh child processes -- before it makes any GPU
call -- relays rank 0's JSON line and propagates a non-zero exit.  A WORLD_SIZE that differs from
--gpus, or fewer visible devices than ranks, is an error, never a silent one-GPU run.
By default every rank proves its own independent traces (proofs are independent objects: no
data-path collective, weak scaling, `value` is the aggregate over ranks).  After that measurement
every N > 1 run also proves BASELINE config 4 (2^22 x 64, log_blowup 4, 16 queries) as ONE proof
sharded over the ranks (tap-stark_amd/csrc/sharded.cpp over the library's native RCCL communicator)
and attaches it as `sharded_config4`.  `--mode sharded` makes the sharded proof the measured step
itself (strong scaling, the latency mode; `--workload config4`).

Prints ONE JSON line on rank 0 with the driver's contract fields plus:
  roofline     -- the dominant kernel's achieved algorithmic-bytes rate from HIP events recorded
                  around every launch on the library's own stream (a separate, untimed pass);
  cpu_baseline -- the CPU oracle (a port: the Rust reference cannot run here) on a bounded
                  sample of the same workload, on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

# ---- wall budget: the driver gives a run 600 s.  Phases of a rank-0 process, the
# seconds each is PLANNED to take on an 8-GPU node (from the one-GPU records: driver_run_s 32 for the
# default run of which 22 are the CPU leg; sharded blocks: 2 variants x (6 proofs of ~10 / ~6 ms + trace
# generation + one stage-timer proof), bounded by their watchdog), and what they took.
T_PROCESS_START = time.perf_counter()
WALL_LIMIT_S = 600.0
PLANNED_S = {"start-up (imports, rendezvous, context, AIR compile)": 40.0,
             "replicas: warm-up + timed windows": 10.0,
             "rank-0 legs (kernel timers, latency, h2d, clocks, cpu_baseline)": 100.0,
             "sharded_config4": 60.0, "sharded_config5": 60.0}
_PHASES = []


def phase_done(name: str):
    _PHASES.append((name, time.perf_counter()))


def wall_budget() -> dict:
    el, t = {}, T_PROCESS_START
    for name, t1 in _PHASES:
        el[name] = round(t1 - t, 2)
        t = t1
    return {"limit_s": WALL_LIMIT_S, "planned_s": PLANNED_S, "planned_total_s": sum(PLANNED_S.values()),
            "elapsed_s": el, "elapsed_total_s": round(time.perf_counter() - T_PROCESS_START, 2),
            "note": "sharded blocks: 2 variants each (replicated, localq), under a watchdog of "
                    "TS_BENCH_SHARD_TIMEOUT_S (default 200 s for both) that prints the record it has"}

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# Protocol dry-run on a CPU (tests/test_bench_dist_cpu.py): TS_BENCH_STUB_LIB names a directory holding a
# stand-in `tapstark_amd` package (tests/stub_lib: objects that sleep); everything else in this file runs
# as it will on the node.  The record says so in `data` and `stub_lib`; no number in it is a measurement.
STUB_LIB = os.environ.get("TS_BENCH_STUB_LIB")
if STUB_LIB:
    sys.path.insert(0, os.path.abspath(STUB_LIB))

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` as a plain command
def spawn_ranks(n_ranks: int) -> int:
    """Starts the N ranks as children of this (GPU-free) process and waits for them.  Rank 0's
    stdout is relayed (its last line is the JSON record); any rank's failure ends the job non-zero.
    Nothing here imports torch or touches the GPU: a process that has initialised the GPU must not
    be replaced or re-executed (and is not: the ranks are fresh processes)."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   TS_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = []

    def pump():
        for line in procs[0].stdout:
            out0.append(line)

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    rc = 0
    alive = set(range(n_ranks))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    t.join(timeout=10)
    sys.stdout.buffer.write(b"".join(out0))
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------
def workload(name: str, log_n: int, need_host_trace: bool):
    """(air, host trace or None, public values, description, fri config, device-side generator).
    The traces are generated ON THE DEVICE (ts_trace_*): inputs are born in HBM; a host copy is only
    made when a mode needs to slice it."""
    import numpy as np

    import tapstark_amd as ts
    from tapstark_amd.airs import (FibonacciAir, SynthMulAir, generate_fibonacci_trace,
                                   generate_synth_mul_trace)
    n = 1 << log_n
    if name in ("config3", "config4"):
        air = SynthMulAir(64)
        trace = generate_synth_mul_trace(n) if need_host_trace else None
        pis = np.zeros(0, dtype=np.uint32)
        cfg = (2, 28, 8) if name == "config3" else (4, 16, 8)
        desc = (f"SynthMulAir-64 (build-defined), trace 2^{log_n}x64, log_blowup={cfg[0]}, "
                f"{cfg[1]} queries, pow 8")
        return air, trace, pis, desc, cfg, (n, 64), lambda c: ts.DeviceMatrix.synth_mul(c, n, 64)
    if name == "config5":  # build-defined stand-in for the RISC0-recursion-style AIR
        from tapstark_amd.airs import SynthExtAir, generate_synth_ext_trace
        air = SynthExtAir(163)
        host = generate_synth_ext_trace(n, 163) if need_host_trace else None
        pis = np.zeros(0, dtype=np.uint32)
        desc = (f"SynthExt-163 (build-defined, EF4 multiplication constraints), trace 2^{log_n}x163, "
                "log_blowup=4, 16 queries, pow 8")
        return air, host, pis, desc, (4, 16, 8), (n, 163), lambda c: ts.DeviceMatrix.synth_ext(c, n, 163)
    if name == "config2":
        air = FibonacciAir()
        trace = generate_fibonacci_trace(0, 1, n) if need_host_trace else None

        def last_right(c):  # pis = [a, b, trace[n-1].right]  (fib_air.rs:133-139)
            return int(ts.DeviceMatrix.fibonacci(c, 0, 1, n).download()[-1, 1])

        desc = f"Fibonacci AIR, trace 2^{log_n}x2, log_blowup=2, 28 queries, pow 8"
        return (air, trace, last_right, desc, (2, 28, 8), (n, 2),
                lambda c: ts.DeviceMatrix.fibonacci(c, 0, 1, n))
    raise SystemExit(f"unknown workload {name}")


def algorithmic_bytes_per_proof(n: int, w: int, b: int, qd: int) -> dict:
    """Algorithmic HBM bytes per proof, per kernel (DESIGN.md "Kernels and their rooflines"):
    each array counted once per launch that must read or write it."""
    N = n << b
    wall = w + 4 * qd  # every committed column (trace + quotient chunks)
    # Merkle parents (64 B read + 32 B written each): two N-leaf trees + FRI trees (N/2 + N/4 + ..).
    # Levels with > 2^17 children take one k_merkle_level launch each; the last 17 levels of a tree
    # are one k_merkle_tree launch; a FRI round with <= 2^17 leaves is one k_fri_round launch (fold
    # of the previous vector + leaf hashes + the whole tree), the rounds above keep fold / levels.
    TREE = 17
    parents_all = 3 * N  # N - 1 per N-leaf tree, and N/2 + N/4 + ... over the FRI trees
    fri_leaves = [N >> r for r in range(1, 40) if (N >> r) > 1024 // 2]
    lvl2 = lvl1 = 0
    for leaves in [N, N] + [h for h in fri_leaves if h > (1 << TREE)]:
        p = leaves // 2
        while p >= (1 << TREE):
            lvl1 += p
            p //= 2
    fused = [h for h in fri_leaves if h <= (1 << TREE)]
    fused_bytes = sum((64 + 8 + 32 + 32 + 96) * h for h in fused)  # prev, twiddles, cur, leaf digests, parents
    small = max(parents_all - lvl1 - lvl2 - sum(fused), 0)
    big_fri_elems = sum(2 * h for h in fri_leaves if h > (1 << TREE))
    return {
        "k_transpose_bitrev": 8 * n * w,
        "k_transpose_bitrev_r16": 8 * n * w,
        "k_intt_contig": 8 * n * wall,
        "k_intt_contig_rm": 8 * n * w,
        "k_lde_mid<1>": 4 * n * wall + 4 * N * wall,
        "k_lde_mid<0>": 4 * n * wall + 4 * N * wall,
        "k_lde_fwd_contig": 8 * N * wall,
        # the trace (one matrix, width a multiple of 16) takes the strided kernel, the batch of
        # quotient chunks the pointer-table one; any other width sends both through the latter
        "k_leaf_hash_strided": 4 * N * w + 32 * N,
        "k_leaf_hash<2>": 4 * N * wall + 2 * 32 * N,
        "k_leaf_hash<1>": (4 * N * 4 * qd + 32 * N) if w % 16 == 0 else (4 * N * wall + 2 * 32 * N),
        "k_leaf_hash_ef_pairs": 16 * N + 16 * N,
        "k_merkle_level<2>": 96 * lvl2,
        "k_merkle_level<1>": 96 * lvl1,
        "k_merkle_tree": 96 * small,
        "k_fri_round<true>": fused_bytes,
        "k_selectors": 12 * n * qd,
        "k_quotient_jit": 4 * n * qd * w + 12 * n * qd + 16 * n * qd,
        "k_quotient<256>": 4 * n * qd * w + 12 * n * qd + 16 * n * qd,
        "k_bary_weights": 32 * n,
        "(k_bary_dots<2, 64>)": 4 * n * w + 32 * n,
        "(k_bary_dots<1, 8>)": qd * (16 * n + 16 * n),
        "k_reduce_fused": 4 * N * wall + 16 * N,
        "k_fri_fold_pairs": 16 * big_fri_elems + 8 * big_fri_elems + 8 * big_fri_elems,
        **_leaf_tree_bytes(N, w, qd, fri_leaves, TREE),
    }


def _leaf_tree_log_r(log_leaves: int) -> int:
    """csrc/merkle_tree.hpp leaf_tree_log_r: leaves per lane (log2) of the leaf-tree kernel."""
    return 2 if log_leaves >= 20 else (log_leaves - 18 if log_leaves >= 18 else 0)


def _leaf_tree_bytes(N: int, w: int, qd: int, fri_leaves, tree_log: int) -> dict:
    """The round-5 Merkle path (leaf_tree.hpp): leaves + every level in one launch, named
    k_leaf_tree<log2 leaves per lane, leaf kind>.  A tree of L leaves is 2L - 1 digests of 32 bytes,
    each written once; the leaf launch also reads what it hashes."""
    log_N = N.bit_length() - 1
    out = {}

    def add(name, b):
        out[name] = out.get(name, 0) + b
    if log_N >= 8:
        r = _leaf_tree_log_r(log_N)
        # the trace, and the batch of quotient chunks (commit() lays equal-height matrices back to back,
        # so both are "one matrix addressed by stride": the same kernel, two launches per proof)
        add(f"k_leaf_tree<{r},strided>", 4 * N * w + 64 * N)
        add(f"k_leaf_tree<{r},strided>", 4 * N * 4 * qd + 64 * N)
    first = True
    for h in fri_leaves:
        lg = h.bit_length() - 1
        if h > (1 << tree_log) and lg >= 8:
            r = _leaf_tree_log_r(lg)
            if first:
                add(f"k_leaf_tree<{r},fri_leaf>", 32 * h + 64 * h)            # vector of 2h EF4 in, tree out
            else:
                add(f"k_leaf_tree<{r},fri_fold>", 64 * h + 8 * h + 32 * h + 64 * h)  # prev, twiddles, cur, tree
        first = False
    return out


def _omp_set_threads(n: int):
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def cpu_baseline(gpu_proof_words=None) -> dict:
    """The oracle prover (oracle/, a C port of the reference's algorithm: the Rust reference cannot
    be built here) on this box's host cores, on the SAME workload at its full size: the median of 3
    runs of prove() on SynthMulAir-64 2^20 x 64 (log_blowup 2, 28 queries) after a warm-up, with every
    core of the GPU's share, plus one 1-thread sample on 2^16 rows -- the
    reference as configured is single-threaded (no manifest enables p3-maybe-rayon's `parallel`,
    SURVEY.md section 2).  A probe on 2^15 rows bounds the leg: a host too slow for three full-size
    runs in ~45 s gets the largest 2^k that fits, and says so in `sample`.

    The oracle's full-size proof is kept and compared word for word with `gpu_proof_words` (a proof
    of the same trace taken from the timed region): `matches_oracle`.  This leg is the only place
    bench.py touches oracle/ (with the fold leg's checker)."""
    import numpy as np

    import tapstark_amd as ts
    from oracle import oracle_py as orc
    from tapstark_amd.airs import SynthMulAir, generate_synth_mul_trace

    # the GPU box gives one GPU's share of the host (16 cores); never oversubscribe
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("OMP_NUM_THREADS", min(avail, 16)))
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle is first loaded
    air = SynthMulAir(64)
    tape = ts.air_tape(air, 0)
    cfg = orc.FriConfig(2, 28, 8)
    orc.prove(cfg, tape, generate_synth_mul_trace(1 << 8), [])  # thread-pool warm-up
    _omp_set_threads(cores)
    kept = {}

    def timed(trace):
        t0 = time.perf_counter()
        kept["proof"] = orc.prove(cfg, tape, trace, [], cap_words=1 << 22)
        return time.perf_counter() - t0

    probe_trace = generate_synth_mul_trace(1 << 15)
    est = min(timed(probe_trace), timed(probe_trace)) * 32  # ~linear in rows
    log_n = 20
    while log_n > 15 and est * 3.3 > float(os.environ.get("TS_BENCH_CPU_BUDGET_S", "45")):
        est /= 2
        log_n -= 1
    trace = generate_synth_mul_trace(1 << log_n)
    if log_n < 20:
        timed(trace)  # warm-up at size (at full size the three runs below are all there is time for)
    runs = sorted(timed(trace) for _ in range(3))
    dt = runs[1]
    full_proof = kept["proof"] if log_n == 20 else None
    # one thread, on 1/16 of the rows (about the same wall time per run); TS_BENCH_CPU_ONE_THREAD_FULL=1
    # runs it ONCE at the full size instead (~2 min: outside the default run's time budget; the record of
    # such a run is profiles/r06_cpu_one_thread_full.json)
    one_full = bool(os.environ.get("TS_BENCH_CPU_ONE_THREAD_FULL")) and log_n == 20
    log_n1 = log_n if one_full else max(log_n - 4, 10)
    trace1 = trace if one_full else generate_synth_mul_trace(1 << log_n1)
    _omp_set_threads(1)
    dt1 = timed(trace1) if one_full else min(timed(trace1), timed(trace1))
    _omp_set_threads(cores)
    rec = {"value": (64 << log_n) / dt, "unit": "trace cells/sec", "cores": cores, "kind": "port",
           "sample": f"oracle prove() of SynthMulAir-64 2^{log_n}x64, log_blowup=2, 28 queries"
                     + (" (the full BASELINE size)" if log_n == 20 else " (host too slow for the full size in the budget)")
                     + f": median of 3 runs ({runs[0]:.2f}..{runs[2]:.2f} s, median {dt:.2f} s), OpenMP over "
                       f"{cores} host threads",
           "proofs_per_sec": 1.0 / dt,
           "runs_s": [round(r, 3) for r in runs],
           "full_size": log_n == 20,
           "one_thread": {"value": (64 << log_n1) / dt1, "unit": "trace cells/sec", "cores": 1,
                          "full_size": log_n1 == 20,
                          "sample": f"the same prover on 2^{log_n1}x64, "
                                    + ("one run" if one_full else "best of 2") + f" ({dt1:.2f} s), 1 thread"
                                    + ("" if log_n1 == 20 else "; NOT the full size: a rate on 1/16 of the rows "
                                       "(the full-size single-thread run is profiles/r06_cpu_one_thread_full.json)")}}
    if full_proof is not None:
        rec["oracle_proof_blake3"] = orc.blake3(full_proof.tobytes()).hex()
        if gpu_proof_words is not None:
            g = np.ascontiguousarray(gpu_proof_words, dtype=np.uint32)
            rec["gpu_proof_blake3"] = orc.blake3(g.tobytes()).hex()
            rec["matches_oracle"] = bool(len(g) == len(full_proof) and (g == full_proof).all())
    return rec


# ------------------------------------------------------------------------------------------------
def fold_benchmark(args):
    """The reference's only benchmark, fri/benches/fold_even_odd.rs:14-46 (a fold over vectors of
    2^12 .. 2^22 elements), on this hardware: FriGenericConfig::fold_matrix (two_adic_pcs.rs:116-147, the
    fold the prover runs) on EF4 vectors of those sizes, device-resident (ts_fri_fold_device ->
    k_fri_fold_pairs), timed with HIP events around every launch on the library's stream.  Bytes per fold:
    2h EF4 read + h EF4 written + h twiddles = 52 h.  cpu_baseline leg: the oracle's fold_matrix on one
    host core beside it (and the check that the device result equals it)."""
    import ctypes as C

    import numpy as np
    import torch

    import tapstark_amd as ts
    from tapstark_amd import _lib

    P = 0x78000001
    ctx = ts.default_context()
    l = _lib.lib()
    rng = np.random.default_rng(3)
    beta = rng.integers(0, P, 4, dtype=np.uint32)
    bp = beta.ctypes.data_as(C.POINTER(C.c_uint32))
    rows, held = [], []
    for log_size in (12, 14, 16, 18, 20, 22):  # the reference's sweep: elements of the INPUT vector
        n = 1 << log_size
        h = n // 2
        vec = rng.integers(0, P, (n, 4), dtype=np.uint32)
        d_in = torch.from_numpy(vec.view(np.int32)).to("cuda:0")
        d_out = torch.zeros((h, 4), dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()

        def fold():
            ctx.check(l.ts_fri_fold_device(ctx.h, d_in.data_ptr(), h, bp, d_out.data_ptr()))
        fold()
        ctx.synchronize()
        reps = max(args.steps, 10)
        ctx.set_kernel_timing(True)
        for _ in range(reps):
            fold()
        kt = ctx.take_kernel_timings()
        ctx.set_kernel_timing(False)
        name = next(k for k in kt if "fold" in k)
        us = 1e3 * kt[name][1] / kt[name][0]
        nbytes = 52 * h
        rows.append({"log_size": log_size, "elements_in": n, "kernel": name, "kernel_us": round(us, 3),
                     "alg_bytes": nbytes, "GB_per_s": round(nbytes / (us * 1e-6) / 1e9, 1),
                     "frac_of_8TBps": round(nbytes / (us * 1e-6) / HBM_PEAK, 4),
                     "elements_per_s_gpu": round(n / (us * 1e-6))})
        held.append((vec, d_out.cpu().numpy().view(np.uint32)))
    # ---- cpu_baseline leg (the only place the oracle is touched): host timing + result check
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import oracle_py as orc
        crow = []
        for r, (vec, got) in zip(rows, held):
            best, want = 1e9, None
            for _ in range(5):
                t0 = time.perf_counter()
                want = orc.fold_matrix(vec, beta)
                best = min(best, time.perf_counter() - t0)
            assert (got == want).all(), f"device fold of 2^{r['log_size']} differs from the oracle"
            crow.append({"log_size": r["log_size"], "oracle_host_us": round(1e6 * best, 1),
                         "elements_per_s_host": round(r["elements_in"] / best)})
        cpu = {"value": crow[-1]["elements_per_s_host"], "unit": "input elements/sec", "cores": 1, "kind": "port",
               "sample": "oracle fold_matrix (C, one thread), best of 5 per size; value = the 2^22 row", "rows": crow}
    big = rows[-1]
    rec = {"metric": "fold_matrix on EF4 vectors (fri/benches/fold_even_odd.rs:14-46 sizes), input elements/sec at 2^22",
           "value": big["elements_per_s_gpu"], "unit": "input elements/sec", "n_gpus": 1, "steps": max(args.steps, 10),
           "warmup": 1, "ms_per_step": big["kernel_us"] * 1e-3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u32", "data": "synthetic (uniform field elements, resident in HBM)",
           "config": {"workload": "fold_matrix, EF4 vectors of 2^12 .. 2^22 elements, one launch each"},
           "roofline": {"bound": "hbm", "kernel": big["kernel"], "achieved": big["GB_per_s"], "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": big["frac_of_8TBps"], "traffic": None,
                        "alg_bytes_per_launch": big["alg_bytes"], "avg_launch_ms": big["kernel_us"] * 1e-3},
           "cpu_baseline": cpu, "rows": rows,
           "note": "below ~2^19 elements a launch is at the ~7 us floor HIP events see for any kernel; the prover "
                   "never launches such folds alone (rounds with <= 2^17 leaves fold inside k_fri_round / k_fri_tail)"}
    print(json.dumps(rec), flush=True)


# ------------------------------------------------------------------------------------------------
class Watchdog:
    """Bounds a block that contains collectives nobody has run on more than one rank yet: if it is
    still running at the deadline, `on_timeout` runs on the watchdog thread (rank 0 prints the record
    it has, every rank leaves with os._exit)."""

    def __init__(self, seconds: float, on_timeout):
        self._ev = threading.Event()
        self.phase = "start"

        def run():
            if not self._ev.wait(seconds):
                on_timeout(self.phase)

        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def done(self):
        self._ev.set()


SHARDED_BLOCKS = {
    # BASELINE.json configs[3] and configs[4] ("... sharded over 8xMI355X", "... on 8xMI355X")
    "sharded_config4": {"air": "mul64", "log_n": 22, "width": 64, "cfg": (4, 16, 8),
                        "desc": "SynthMulAir-64, trace 2^{log_n}x64, log_blowup=4, 16 queries, pow 8"},
    "sharded_config5": {"air": "ext163", "log_n": 20, "width": 163, "cfg": (4, 16, 8),
                        "desc": "SynthExt-163 (build-defined), trace 2^{log_n}x163, log_blowup=4, 16 queries, pow 8"},
}


def sharded_block(name: str, ts, env, ctx, dev, watchdog: Watchdog | None, state: dict) -> dict:
    """One of BASELINE's sharded configurations as ONE proof over the ranks: ts_prove_sharded over the
    library's native RCCL communicator (torch.distributed callbacks where the process group is not
    nccl: the gloo rehearsal).  Variants, A/B in one lease (TS_BENCH_SHARD_VARIANTS picks):
      replicated   every rank generates the trace on its device; FRI rounds stay sharded while a slab
                   holds >= 2^12 values (min_local_log 12)
      mll16/mll20  the same with min_local_log 16 / 20: fewer sub-root all-gathers, more replicated
                   tail rounds (csrc/sharded.cpp); not in the default list: on the per-rank solo times
                   they never paid (profiles/r04_config4_shard_stages.json)
      localq       every rank evaluates the quotient on its own cosets: no chunk broadcast, no rank
                   waiting for the owner of the quotient domain (ts_shard_options.local_quotient)
      sliced       row slices in (adds the trace all-gather)
    Every variant reports ms/step, per-rank stage times and the table of its collectives (kind, bytes,
    count, ms) from one extra proof with the stage timers on."""
    import hashlib

    import numpy as np
    import torch.distributed as dist

    from tapstark_amd.airs import SynthExtAir, SynthMulAir
    from tapstark_amd.benchutil import split_stage_timings

    def phase(p):
        if watchdog is not None:
            watchdog.phase = f"{name}: {p}"

    spec = SHARDED_BLOCKS[name]
    log_n = int(os.environ.get("TS_BENCH_SHARD_LOG_N", spec["log_n"]))
    n, w, cfg = 1 << log_n, spec["width"], spec["cfg"]
    steps = int(os.environ.get("TS_BENCH_SHARD_STEPS", "4"))
    world = env.world
    gsize = min(world, 1 << cfg[0])
    while world % gsize or (gsize & (gsize - 1)):
        gsize -= 1
    n_groups = world // gsize
    grank = env.rank % gsize
    out = {"workload": spec["desc"].format(log_n=log_n) + f": ONE proof sharded over {gsize} rank(s)"
                       + (f", {n_groups} groups" if n_groups > 1 else ""),
           "steps": steps, "group_size": gsize, "n_groups": n_groups}
    phase("communicator")
    if "comm" not in state:  # one communicator for every block of the run
        group = None
        if n_groups > 1:
            for gi in range(n_groups):  # every rank takes part in creating every group
                gr = dist.new_group(list(range(gi * gsize, (gi + 1) * gsize)))
                if env.rank // gsize == gi:
                    group = gr
        # (the stub library's dry-run takes the native branch too: its RcclComm checks the id hand-out)
        use_native = os.environ.get("TS_BENCH_COMM", "rccl") == "rccl" and (dist.get_backend() == "nccl" or bool(STUB_LIB))
        if use_native:
            from tapstark_amd import comm as tcomm
            ids = [None] * world
            dist.all_gather_object(ids, tcomm.rccl_unique_id() if grank == 0 else None)
            comm = tcomm.RcclComm(ctx, ids[(env.rank // gsize) * gsize], grank, gsize)
            info = comm.info()
            state["comm_desc"] = {"comm": "rccl (native ts_comm: ncclAllGather / ncclBroadcast on the context's stream)",
                                  "n_ranks_seen_by_rccl": info["comm_count"], "rccl_info_rank0": info}
        else:
            from tapstark_amd.dist import TorchComm
            comm = TorchComm(dev, group=group)
            state["comm_desc"] = {"comm": f"torch.distributed ({comm.backend}; host-staged unless nccl)",
                                  "n_ranks_seen_by_rccl": None}
        state["comm"], state["native"] = comm, use_native
    comm = state["comm"]
    out.update(state["comm_desc"])
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    air = SynthMulAir(64) if spec["air"] == "mul64" else SynthExtAir(163)
    cair = ts.CompiledAir(ctx, ts.air_tape(air, 0))
    pis = np.zeros(0, dtype=np.uint32)

    def gen():
        return ts.DeviceMatrix.synth_mul(ctx, n, w) if spec["air"] == "mul64" else ts.DeviceMatrix.synth_ext(ctx, n, w)

    variants = os.environ.get("TS_BENCH_SHARD_VARIANTS", "replicated,localq").split(",")
    out["variants"] = {}
    digests = set()
    for var in variants:
        var = var.strip()
        if var not in ("replicated", "localq", "mll16", "mll20", "sliced"):
            continue
        phase(f"{var}: inputs")
        sliced = var == "sliced"
        kw = dict(trace_replicated=not sliced,
                  local_quotient=(var == "localq"), min_local_log={"mll16": 16, "mll20": 20}.get(var, 12))
        if sliced:
            full = gen().download()
            rows = np.ascontiguousarray(full[grank * n // gsize:(grank + 1) * n // gsize])
            del full
            mats = [ts.DeviceMatrix.upload(ctx, rows) for _ in range(steps + 2)]
        else:
            mats = [gen() for _ in range(steps + 2)]

        def prove_one(i):
            return ts.prove_sharded(config, cair, ts.BfChallenger(), mats[i], pis, comm, **kw)

        phase(f"{var}: warm-up proof")
        first = prove_one(0)
        ctx.synchronize()
        dist.barrier()
        phase(f"{var}: timed proofs")
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            prove_one(i)
        ctx.synchronize()
        dist.barrier()
        dt = env.max_over_ranks(time.perf_counter() - t0)
        phase(f"{var}: stage timers")
        ctx.set_timing(True)
        prove_one(steps + 1)
        mine, colls = split_stage_timings(ctx.take_timings())
        ctx.set_timing(False)
        stages = [None] * world
        dist.all_gather_object(stages, mine)
        coll_all = [None] * world
        dist.all_gather_object(coll_all, colls)
        # every rank must hold the same proof: compare a digest of the words
        digs = [None] * world
        dist.all_gather_object(digs, hashlib.sha256(first.words.tobytes()).hexdigest())
        mygroup = digs[(env.rank // gsize) * gsize:(env.rank // gsize + 1) * gsize]
        digests.update(mygroup)
        out["variants"][var] = {
            "min_local_log": kw["min_local_log"],
            "ms_per_step": round(1e3 * dt / steps, 4),
            "proofs_per_sec": round(n_groups * steps / dt, 3),
            "trace_cells_per_sec": n_groups * steps * float(n * w) / dt,
            "all_ranks_same_proof": len(set(mygroup)) == 1,
            "proof_words": int(len(first.words)), "proof_sha256": mygroup[0],
            "shard_stages_ms_per_rank": stages,
            # rank 0's table + the slowest rank's total per row: where the collective time went
            "collectives_rank0": coll_all[0],
            "collectives_ms_total_per_rank": [round(sum(r["ms_total"] for r in c), 3) for c in coll_all],
            "collectives_count": sum(r["count"] for r in coll_all[0])}
        del mats
    out["all_variants_same_proof"] = len(digests) == 1
    if "replicated" in out["variants"]:
        out["ms_per_step"] = out["variants"]["replicated"]["ms_per_step"]
        out["shard_stages_ms_per_rank"] = out["variants"]["replicated"]["shard_stages_ms_per_rank"]
    phase("done")
    return out


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed proofs before the timed region, dealt to the lanes in turn; at least one "
                         "per lane is always run (tables, device pools and code objects are per context)")
    ap.add_argument("--workload", default="config3", choices=["config3", "config2", "config4", "config5", "fold"],
                    help="fold: the reference's only benchmark (fri/benches/fold_even_odd.rs:14-46) on this "
                         "hardware -- fold_matrix on EF4 vectors of 2^12 .. 2^22 elements, N = 1 only")
    ap.add_argument("--log-n", type=int, default=None, help="default 20 (22 for config4)")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"],
                    help="N > 1: independent proofs per GPU (default) or one proof over all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the roofline / latency / h2d / cpu legs (parameter sweeps)")
    ap.add_argument("--no-sharded-block", action="store_true",
                    help="N > 1: do not append the sharded config-4 / config-5 measurements")
    ap.add_argument("--host-traces", action="store_true",
                    help="hand every step its trace as a HOST buffer (ts_matrix_upload inside the "
                         "timed region): the PCIe-inclusive rate, reported in DESIGN.md, never `value`")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TS_BENCH_STREAMS", "4")),
                    help="independent proofs in flight per GPU (one context = one HIP stream and one "
                         "host thread each); the K timed steps are shared among them")
    ap.add_argument("--stagger-ms", type=float, default=float(os.environ.get("TS_BENCH_STAGGER_MS", "-1")),
                    help="minimum spacing between the starts of two proofs on one GPU, so that the lanes run in "
                         "complementary phases instead of lockstep (-1 = auto: a quarter of the time one "
                         "proof takes alone, measured before the run; 0 = off)")
    ap.add_argument("--windows", type=int, default=3,
                    help="timed windows of K steps each, back to back; `value` is the FIRST (the contract's "
                         "K steps), the others are reported as spread")
    args = ap.parse_args()

    # ---- launch protocol (before anything imports torch or touches the GPU)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to run "
                         "(launch N ranks, or run `python bench.py --gpus N` as a plain command)")

    if args.workload == "fold":
        if args.gpus != 1:
            raise SystemExit("bench.py --workload fold runs on one GPU")
        return fold_benchmark(args)

    # Native libraries print to stdout on their own (RCCL's version banner at communicator creation):
    # everything but the one JSON line goes to stderr, the line itself to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.log_n is None:
        args.log_n = 22 if args.workload == "config4" else 20
    sharded = args.mode == "sharded"
    stub = bool(os.environ.get("TS_BENCH_STUB"))  # protocol tests on CPU: no prover, no GPU
    share_gpu = bool(os.environ.get("TS_BENCH_SHARE_GPU"))
    if share_gpu and "TS_BENCH_BACKEND" not in os.environ:
        os.environ["TS_BENCH_BACKEND"] = "gloo"  # RCCL refuses two ranks on one device

    from tapstark_amd.benchutil import init_dist, run_timed

    env = init_dist()
    assert env.world == args.gpus

    def emit(record: dict):
        sys.stdout.flush()
        if isinstance(record, dict) and "wall_budget" not in record:
            record["wall_budget"] = wall_budget()
        if isinstance(record, dict) and "headline_windows" in record:
            # the driver's record keeps the END of this line verbatim: the per-window table goes last
            record["headline_windows"] = record.pop("headline_windows")
        os.write(real_stdout, (json.dumps(record) + "\n").encode())

    if stub:
        if os.environ.get("TS_BENCH_STUB_FAIL_RANK") == str(env.rank):
            raise SystemExit(f"rank {env.rank}: TS_BENCH_STUB_FAIL_RANK")
        res = run_timed(env, lambda i: time.sleep(0.002), args.steps, args.warmup, lambda: None,
                        units_per_step=1.0, extra_windows=max(args.windows - 1, 0))
        rec = {"metric": "stub", "stub": True, "value": res["value"], "unit": "steps/sec", "n_gpus": env.world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "windows_ms_per_step": res.get("windows_ms_per_step"),
               "data": "stub (TS_BENCH_STUB: launch / timing protocol only, no prover, no GPU)",
               "self_launched": bool(os.environ.get("TS_BENCH_SELF_LAUNCHED")),
               "sharded_config4": {"skipped": "stub"} if env.world > 1 else None,
               "sharded_config5": {"skipped": "stub"} if env.world > 1 else None}
        env.close()
        if env.rank == 0:
            emit(rec)
        return

    import numpy as np
    # Experiment knob (N = 1 only): TS_BENCH_NO_TORCH=1 keeps torch -- and with it the libamdhip64 its wheel bundles
    # (ROCm 7.0), which the library would otherwise be bound to -- out of the process: the library then runs on the
    # system's HIP runtime (ROCm 7.2), as a compiled host's process does.
    no_torch = os.environ.get("TS_BENCH_NO_TORCH") == "1" and env.world == 1 and not sharded and not STUB_LIB
    if no_torch:
        os.environ["TS_PRELOAD_TORCH"] = "0"
        torch = None
    else:
        import torch

    if not share_gpu and not STUB_LIB and not no_torch:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", env.world))
        have = torch.cuda.device_count()
        if have < local_world or env.local_rank >= have:
            raise SystemExit(f"bench.py: {local_world} rank(s) on this node but only {have} visible GPU(s) "
                             "(TS_BENCH_SHARE_GPU=1 puts every rank on GPU 0 for a rehearsal)")

    import tapstark_amd as ts

    if not os.path.exists(ts._lib.LIB_PATH):  # normally built by __graft_entry__.build() beforehand
        from tapstark_amd.build import build
        if env.rank == 0:
            build()
        if env.dist is not None:
            env.dist.barrier()
    # one GPU per rank (TS_BENCH_SHARE_GPU=1 lets a rehearsal put every rank on GPU 0)
    dev = 0 if share_gpu else env.local_rank
    ctx = ts.Context(dev)  # raises without a GPU: there is no fallback path

    air, trace, pis, desc, cfg, (n, w), make_trace = workload(args.workload, args.log_n, False)
    if callable(pis):
        pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    qd = 1 << cair.log_quotient_degree

    # Proofs are independent, so S of them are kept in flight per GPU: S contexts (one HIP stream,
    # one device pool and one host thread each).  The serial transcript of one proof leaves the
    # GPU idle at its host round trips; the next proof's kernels fill those gaps.
    # The K timed steps are dealt to the lanes in advance, so K should be a multiple of S: if the
    # requested lane count does not divide K, a neighbouring one that does is used (K = 10 -> 5 lanes).
    S = 1 if sharded else max(1, min(args.streams, args.steps))
    if S > 1 and args.steps % S:
        S = next((c for c in (S + 1, S - 1, S + 2, S - 2) if c >= 2 and args.steps % c == 0), S)
    lanes = [(ctx, config, cair)]
    for _ in range(1, S):
        c2 = ts.Context(dev)
        lanes.append((c2, ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c2)),
                      ts.CompiledAir(c2, ts.air_tape(air, len(pis)))))

    # inputs resident in HBM before the timed region (prove() consumes its trace, like the
    # reference's moved RowMajorMatrix, so one copy per step); step i runs on lane i % S.
    # Warm-up: at least one proof per lane (W = 5 on 4 lanes: 2, 1, 1, 1).
    warmup = max(args.warmup, S) if not sharded else args.warmup
    n_windows = max(1, args.windows)
    while n_windows > 1 and (warmup + n_windows * args.steps) * n * w * 4 > (100 << 30):
        n_windows -= 1
    total = warmup + n_windows * args.steps
    last = {}
    lane_log = [[] for _ in range(S)]  # per lane: (t ts_prove called, t returned, s waited at the start gate)
    gsize, n_groups, comm = 1, 1, None
    if sharded:
        # one proof per step over all ranks: rank g is handed natural rows [g n/G, (g+1) n/G)
        import torch.distributed as dist
        from tapstark_amd.dist import TorchComm

        if env.dist is None:  # a world of one still goes through the RCCL callbacks
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29512", rank=0, world_size=1,
                                    device_id=torch.device("cuda", dev))
        # a proof can be split over at most 2^log_blowup ranks (whole cosets per rank): with more
        # GPUs than that, groups of that size each prove their own proofs (TS_BENCH_GROUP overrides
        # the group size for rehearsals)
        gsize = min(env.world, 1 << cfg[0], int(os.environ.get("TS_BENCH_GROUP", env.world)))
        while env.world % gsize:
            gsize //= 2
        n_groups = env.world // gsize
        group = None
        if n_groups > 1:
            for gi in range(n_groups):  # every rank takes part in creating every group
                gr = dist.new_group(list(range(gi * gsize, (gi + 1) * gsize)))
                if env.rank // gsize == gi:
                    group = gr
        grank = env.rank % gsize
        # the collectives: the library's own RCCL communicator (csrc/comm.cpp: ncclAllGather /
        # ncclBroadcast on the context's stream; what a Rust host would link), bootstrapped with a
        # unique id made by each group's first rank and handed round over torch.distributed;
        # TS_BENCH_COMM=torch uses the torch.distributed callbacks (tap-stark_amd/dist.py) instead
        use_native = os.environ.get("TS_BENCH_COMM", "rccl") == "rccl" and dist.get_backend() == "nccl"
        if use_native:
            from tapstark_amd import comm as tcomm
            ids = [None] * dist.get_world_size()
            dist.all_gather_object(ids, tcomm.rccl_unique_id() if grank == 0 else None)
            comm = tcomm.RcclComm(ctx, ids[(env.rank // gsize) * gsize], grank, gsize)
            comm.backend = f"rccl (native ts_comm; ncclCommCount = {comm.info()['comm_count']})"
        else:
            comm = TorchComm(dev, group=group)
        # every rank generates the whole trace on its own device (ts_trace_*): nothing to exchange
        # for the input; TS_BENCH_SLICED=1 hands out row slices instead (adds the trace all-gather)
        sliced = bool(os.environ.get("TS_BENCH_SLICED"))
        localq = os.environ.get("TS_BENCH_LOCALQ", "1") != "0"  # every rank on its own cosets: no chunk broadcast
        if sliced:
            full = make_trace(ctx).download()
            rows = np.ascontiguousarray(full[grank * n // gsize:(grank + 1) * n // gsize])
            mats = [ts.DeviceMatrix.upload(ctx, rows) for _ in range(total + 1)]
        else:
            mats = [make_trace(ctx) for _ in range(total + 1)]

        def prove_one(i):
            last["proof"] = ts.prove_sharded(config, cair, ts.BfChallenger(), mats[i], pis, comm,
                                             trace_replicated=not sliced, local_quotient=localq)
    else:
        # one resident trace per step (prove() consumes it); beyond 100 GB of them (288 GB of HBM) the
        # trace of a step is generated on the device at the start of the step instead, inside the
        # timed region (measured: 3.4 instead of 3.0 ms/step)
        pregen = total * n * w * 4 <= 100 << 30
        mats = [make_trace(lanes[i % S][0]) for i in range(total)] if pregen else None

        if args.host_traces:
            pregen, mats = False, None
            host_trace = make_trace(ctx).download()

        def prove_one(i):
            c, conf, ca = lanes[i % S]
            if args.host_traces:
                m = ts.DeviceMatrix.upload(c, host_trace)
            else:
                m = mats[i] if pregen else make_trace(c)
            tg = time.perf_counter()
            start_gate()
            t0 = time.perf_counter()
            last["proof"] = ts.prove(conf, ca, ts.BfChallenger(), m, pis)
            lane_log[i % S].append((t0, time.perf_counter(), t0 - tg))  # ts_prove call, return, time held at the gate
            if pregen:
                mats[i] = None  # the matrix handle is spent

    # Start gate: two proofs never start within `stagger` ms of each other.  With every lane starting
    # at the same instant the S proofs run in lockstep -- all of them in their latency-bound phases
    # (Merkle tops, FRI rounds) at the same time -- and only drift into complementary phases after a
    # few proofs (and can drift back: whole windows 10 % slower were seen).  Spacing the starts puts
    # the lanes in complementary phases from the first proof on and keeps them there; the spacing is
    # well under the steady-state distance between starts (one step), so it does not throttle, and
    # the first lane starts at once (a proof alone fills the chip in its long kernels).
    stagger = {"ms": 0.0}
    gate_lock = threading.Lock()
    gate_last = [-1e9]
    primed = {"s": 0.0}

    def start_gate():
        d = stagger["ms"] * 1e-3
        if d <= 0:
            return
        with gate_lock:
            while True:
                wait = gate_last[0] + d - time.perf_counter()
                if wait <= 0:
                    break
                time.sleep(min(wait, 2e-4))
            gate_last[0] = time.perf_counter()

    # How the S lanes are driven.  Default: ONE library call per window (ts_prove_stream): the lane threads, the
    # start gate and the per-proof clock are C++ inside the library, as a compiled host (Rust, C++:
    # examples/prove_stream.cpp) would have them, and the timed region holds no Python.  TS_BENCH_PY_LANES=1 keeps
    # the lane loop of rounds 3-5 (Python threads, one ts_prove call per proof).  Same-box A/B: the same ms/step
    # (2.74-2.78) and the same sporadic slow window either way (profiles/r06_window_hunt.txt) -- the interpreter
    # lock is not what causes those.  The timed region is the same: EXACTLY K complete proofs between two
    # barrier + device-sync pairs.
    py_lanes = os.environ.get("TS_BENCH_PY_LANES") == "1" or sharded or args.host_traces
    stream_ok = (not py_lanes) and S > 1 and hasattr(ts, "prove_stream")
    pool = None
    stream_lanes = [(conf, ca) for _, conf, ca in lanes]
    if S == 1:
        step = prove_one
        run_steps = None
    elif stream_ok and pregen:
        step = prove_one

        def run_steps(first, count):
            idx = list(range(first, first + count))
            t_call = time.perf_counter()
            proof, st, wl = ts.prove_stream(stream_lanes, [mats[i] for i in idx], [i % S for i in idx], pis,
                                            gate_ms=stagger["ms"])
            last["proof"] = proof
            for k, i in enumerate(idx):  # the library's per-proof clock, on this process's time axis
                lane_log[i % S].append((t_call + 1e-3 * st[k], t_call + 1e-3 * (st[k] + wl[k]), 0.0))
                mats[i] = None
            for lg in lane_log:
                lg.sort()
    else:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=S)
        step = prove_one

        def run_steps(first, count):
            # lane l proves steps first+l, first+l+S, ... in its own thread (ctypes drops the GIL)
            def lane_job(l):
                for i in range(first + l, first + count, S):
                    prove_one(i)
            list(pool.map(lane_job, range(S)))
    if S > 1 and pool is None:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=S)  # per-lane priming and the h2d leg still use Python lanes

    # cgroup CPU-bandwidth throttling (cpu.max of the container: e.g. 16 CPUs' worth per 100 ms on a 256-core
    # host): when the container's processes together exceed the quota inside a period, EVERY thread in it is frozen
    # until the period ends -- tens of ms during which all lanes stand still.  Read before and after every call of
    # run_steps (one call = one timed window or warm-up block), so that a window can say whether it was throttled.
    throttle_log = []

    def read_throttle():
        try:
            with open("/sys/fs/cgroup/cpu.stat") as f:
                kv = dict(line.split() for line in f)
            return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
        except (OSError, ValueError):
            return None

    if run_steps is not None:
        _inner_run_steps = run_steps

        def run_steps(first, count):  # noqa: F811
            a = read_throttle()
            t_a = time.perf_counter()
            _inner_run_steps(first, count)
            b = read_throttle()
            if a is not None and b is not None:
                throttle_log.append((t_a, time.perf_counter(), b[0] - a[0], b[1] - a[1]))

    def local_sync():
        for c, _, _ in lanes:
            c.synchronize()
        if env.dist is not None and env.device is not None:
            torch.cuda.synchronize()

    # The interpreter's cyclic garbage collector is switched off for the priming and the timed windows (and what
    # exists is frozen out of its reach): a generation-2 collection in a process that has imported torch takes
    # milliseconds with the interpreter lock held.  A cheap source of jitter to exclude -- but not the source of the
    # sporadic slow window: about one window (or priming probe) in thirty reads 2.9-3.3 ms/step instead of 2.73-2.78
    # because one to four of its proofs take 15-50 ms instead of 11, often on all lanes at once.  Same-box A/Bs
    # (profiles/r06_window_hunt.txt) found it unchanged by: this switch, the amdsmi sampler child, sleeping or
    # polling stream waits (TS_SYNC_SPIN), the lane loop in Python threads or inside the library, per-proof or kept
    # output buffers, torch's bundled HIP runtime or the system's (TS_BENCH_NO_TORCH), transparent huge pages, BLAS
    # threads; the device pool makes no hipMalloc in a window (TS_POOL_DEBUG) and the container's CPU quota does not
    # throttle during one (cpu.stat, read around every window: `cgroup_cpu_throttled`).  A process WITHOUT an interpreter
    # (examples/prove_stream.cpp) showed none in 3600 proofs on the same boxes.  Unexplained; `value` (window 1)
    # draws such a window with that probability, and the line says when it did (`windows`, `headline_windows`).
    import gc
    keep_gc = os.environ.get("TS_BENCH_KEEP_GC") == "1"  # A/B knob
    if not keep_gc:
        gc.collect()
        gc.freeze()
        gc.disable()

    # Stagger: explicit, or auto = a quarter of the step time a single lane sustains (measured here
    # on one untimed proof per lane, which also primes every lane's tables / pools / code objects)
    single_ms = None
    if not sharded:
        prime = [make_trace(lanes[l][0]) for l in range(S)]

        def prime_lane(l):
            c, conf, ca = lanes[l]
            ts.prove(conf, ca, ts.BfChallenger(), prime[l], pis)
        prime_lane(0)
        ctx.synchronize()
        m0 = make_trace(ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        ts.prove(config, cair, ts.BfChallenger(), m0, pis)
        single_ms = 1e3 * (time.perf_counter() - t0)
        if S > 1:
            list(pool.map(prime_lane, range(1, S)))
        del prime
        if S > 1:
            stagger["ms"] = args.stagger_ms if args.stagger_ms >= 0 else 0.25 * single_ms
        # Sustained priming (setup, untimed, on the record as `priming`): before the W warm-up steps the
        # lanes prove freshly generated traces in probes of K steps -- the timed window's own shape: same
        # lanes, same start gate, a device sync on both sides -- until two consecutive probes agree within
        # 1.5 % and at least TS_BENCH_PRIME_MIN_S (0.5 s) of this load have run, for at most TS_BENCH_PRIME_S
        # (default 2 s) in all.  A fresh box needs a fraction of a
        # second of this load before clocks and power management settle (cold first windows of 3.18 ms per
        # step against 2.8 in every later one were seen, and the driver's r05 run: 2.908 against 2.690
        # sustained in the same process).  The metric is sustained proofs per second; TS_BENCH_PRIME_S=0
        # turns the priming off.
        prime_cap = float(os.environ.get("TS_BENCH_PRIME_S", "2.0"))
        prime_min = min(prime_cap, float(os.environ.get("TS_BENCH_PRIME_MIN_S", "0.5")))  # never less than this much load
        probes, probes_thr = [], []
        t_prime0 = time.perf_counter()
        while prime_cap > 0 and time.perf_counter() - t_prime0 < prime_cap:
            pm = [make_trace(lanes[i % S][0]) for i in range(args.steps)]
            for c, _, _ in lanes:
                c.synchronize()

            def probe_job(l):
                c, conf, ca = lanes[l]
                for i in range(l, args.steps, S):
                    start_gate()
                    ts.prove(conf, ca, ts.BfChallenger(), pm[i], pis)
            thr_a = read_throttle()
            tp = time.perf_counter()
            if stream_ok:
                ts.prove_stream(stream_lanes, pm, [i % S for i in range(args.steps)], pis, gate_ms=stagger["ms"],
                                want_times=False)
            elif S > 1:
                list(pool.map(probe_job, range(S)))
            else:
                probe_job(0)
            for c, _, _ in lanes:
                c.synchronize()
            probes.append(round(1e3 * (time.perf_counter() - tp) / args.steps, 4))
            thr_b = read_throttle()
            probes_thr.append(None if thr_a is None or thr_b is None else thr_b[0] - thr_a[0])
            if (len(probes) >= 2 and abs(probes[-1] - probes[-2]) <= 0.015 * min(probes[-1], probes[-2])
                    and time.perf_counter() - t_prime0 >= prime_min):
                break
        primed["s"] = round(time.perf_counter() - t_prime0, 3)
        primed["probes_ms_per_step"] = probes
        primed["probes_cgroup_cpu_throttled_times"] = probes_thr
        primed["settled"] = bool(len(probes) >= 2 and abs(probes[-1] - probes[-2]) <= 0.015 * min(probes[-1], probes[-2]))

    phase_done("start-up (imports, rendezvous, context, AIR compile)")
    # clock / power sampler over the timed windows: a CHILD process (amdsmi only), so that it takes
    # nothing from this process's interpreter lock; rank 0 only; TS_BENCH_SAMPLER=0 turns it off
    sampler = None
    if env.rank == 0 and os.environ.get("TS_BENCH_SAMPLER", "1") != "0":
        from tapstark_amd.benchutil import GpuSamplerProcess
        sampler = GpuSamplerProcess(dev, 0.01)
    for lg in lane_log:
        lg.clear()
    # sharded: the ranks of a group share each step's n*w cells
    res = run_timed(env, step, args.steps, warmup, local_sync,
                    units_per_step=float(n * w) / (gsize if sharded else 1), run_steps=run_steps,
                    extra_windows=n_windows - 1)
    if sampler is not None:
        sampler.stop()
    gc.enable()
    phase_done("replicas: warm-up + timed windows")
    # per window: ms/step, the clock and power the chip held, and what the host side did on each lane
    # (the longest pause between one ts_prove returning and the next being called, and how much of the
    # pauses was the start gate) -- so that a slow window names its cause
    per_window = []
    for k, (w0, w1) in enumerate(res.get("windows_t") or []):
        row = {"window": k + 1, "ms_per_step": round(res["windows_ms_per_step"][k], 4)}
        if sampler is not None:
            row.update(sampler.window(w0, w1) if not sampler.error or sampler.samples else {"sampler_error": sampler.error})
        thr = [e for e in throttle_log if w0 <= e[0] <= w1]
        if thr:
            row["cgroup_cpu_throttled"] = {"times": sum(e[2] for e in thr), "thread_usec": sum(e[3] for e in thr)}
        gaps, gate, first_call, last_ret, lat, per_lane = [], 0.0, [], [], [], []
        for li, lg in enumerate(lane_log):
            evs = [e for e in lg if w0 <= e[0] <= w1]
            if not evs:
                continue
            per_lane.append([round(1e3 * (e[1] - e[0]), 2) for e in evs])
            first_call.append(evs[0][0] - w0)
            last_ret.append(w1 - evs[-1][1])
            gate += sum(e[2] for e in evs)
            lat += [e[1] - e[0] for e in evs]
            gaps += [b[0] - a[1] for a, b in zip(evs, evs[1:])]
        if lat:
            row.update({"longest_host_gap_ms": round(1e3 * max(gaps), 3) if gaps else 0.0,
                        "host_gaps_total_ms": round(1e3 * sum(gaps), 3), "of_which_start_gate_ms": round(1e3 * gate, 3),
                        "first_call_after_window_start_ms": [round(1e3 * x, 3) for x in first_call],
                        "window_end_after_last_return_ms": [round(1e3 * x, 3) for x in last_ret],
                        "proof_latency_in_window_ms_median": round(1e3 * sorted(lat)[len(lat) // 2], 3),
                        "proof_latency_in_window_ms_max": round(1e3 * max(lat), 3),
                        # every proof's wall time, lane by lane in call order: a stall of the GPU shows in
                        # all lanes at once, a stall of one host thread / stream in one
                        "proof_latencies_ms_by_lane": per_lane})
        per_window.append(row)
    shard_stages = None
    if sharded:
        import torch.distributed as dist
        res["steps_per_sec"] = n_groups * args.steps / res["elapsed_s"]
        # one more proof with the stage timers on, on every rank: where a rank's time goes
        ctx.set_timing(True)
        prove_one(total)
        mine = {}
        for k, v in ctx.take_timings():
            mine[k] = round(mine.get(k, 0.0) + v, 3)
        ctx.set_timing(False)
        shard_stages = [None] * dist.get_world_size()
        dist.all_gather_object(shard_stages, mine)

    out = None
    if env.rank == 0:
        proof = last["proof"]
        wins = res.get("windows_ms_per_step") or [res["ms_per_step"]]
        # the last proof of the timed region, checked after it (outside every timed window): the
        # product's native verifier (csrc/verifier.cpp, uni-stark/src/verifier.rs:19-161) must accept it
        import hashlib
        verified = None
        try:
            ts.verify(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx)), air, ts.BfChallenger(), proof, pis)
            verified = True
        except Exception as e:  # noqa: BLE001
            verified = f"REJECTED: {e!r}"
        out = {
            "metric": f"trace cells/sec (proofs/sec alongside), 2^{args.log_n}x{w} BabyBear trace",
            "value": res["value"], "unit": "trace cells/sec", "n_gpus": env.world,
            "steps": args.steps, "warmup": warmup, "ms_per_step": res["ms_per_step"],
            # every timed window of K steps (value / ms_per_step are the first), with the clock, power and
            # host-side gaps of each; `priming`: the untimed probes run before the warm-up steps
            "windows_ms_per_step": [round(x, 4) for x in wins],
            "windows": per_window, "priming": dict(primed),
            "headline_windows": {
                "ms_per_step": [round(x, 4) for x in wins],
                "gfxclk_mhz_median": [r_.get("gfxclk_mhz_median") for r_ in per_window],
                "socket_power_w_median": [r_.get("socket_power_w_median") for r_ in per_window],
                "longest_host_gap_ms": [r_.get("longest_host_gap_ms") for r_ in per_window],
                "cgroup_cpu_throttled_times": [(r_.get("cgroup_cpu_throttled") or {}).get("times") for r_ in per_window],
                "priming_probes_ms_per_step": primed.get("probes_ms_per_step"), "priming_s": primed.get("s"),
                "python_gc": "on (TS_BENCH_KEEP_GC=1)" if keep_gc else "disabled and frozen over priming + timed windows",
                "note": "every timed window of K steps (value = the first); clock / power from an amdsmi child "
                        "process during the window; host gap = longest pause between a ts_prove returning and "
                        "the next call on one lane; the sustained 160-step leg is clocks.prover_sustained"},
            "higher_is_better": True,
            "scaling": ("strong" if n_groups == 1 else "strong within a group, weak across groups")
                       if sharded else "weak", "vs_baseline": None,
            "dtype": "u32",  # BabyBear arithmetic on u32 lanes (64-bit intermediates)
            "data": ("STUB LIBRARY (TS_BENCH_STUB_LIB): protocol dry-run on a CPU, no prover ran, no number is a measurement"
                     if STUB_LIB else
                     "synthetic (trace uploaded from host memory INSIDE the timed region: PCIe-inclusive)"
                     if args.host_traces else
                     "synthetic (trace generated on the device, resident in HBM before the timed region)"),
            "stub_lib": bool(STUB_LIB),
            "config": {"workload": desc, "rows": n, "width": w, "log_blowup": cfg[0],
                       "num_queries": cfg[1], "proof_of_work_bits": cfg[2], "quotient_degree": qd,
                       "parallelism": (f"{n_groups} group(s) of {gsize} GPU(s), one proof sharded over each group, "
                                        f"collectives over {comm.backend}, quotient {'local' if localq else 'broadcast'}" if sharded else
                                       ("1 rank per GPU (replicas)" if env.world > 1 else "1 GPU")
                                       + f", {S} proofs in flight per GPU, starts spaced >= {stagger['ms']:.2f} ms"),
                       "hip_runtime": "system (torch not imported: TS_BENCH_NO_TORCH=1)" if no_torch else "the one torch's wheel bundles",
                       "lane_driver": ("ts_prove_stream (lane threads, gate and per-proof clock inside the library)"
                                       if (not sharded and S > 1 and stream_ok and pregen) else
                                       "python threads (one ts_prove call per proof per lane)" if S > 1 else "one lane"),
                       "quotient_kernel": "hiprtc-specialised" if cair.is_jit else "interpreter",
                       "proof_words": int(len(proof.words))},
            "proofs_per_sec": res["steps_per_sec"],
            # --mode sharded: which quotient path was timed.  Both return ts_prove's proof for every trace
            # (the local path falls back to the broadcast one when FRI's final polynomial is not constant)
            "local_quotient": bool(localq) if sharded else None,
            "timed_proof_verified": verified,
            "timed_proof_sha256": hashlib.sha256(proof.words.tobytes()).hexdigest(),
            "extra": {"windows_ms_per_step": [round(x, 4) for x in wins],
                      "windows_min_ms_per_step": round(min(wins), 4),
                      "windows_median_ms_per_step": round(sorted(wins)[len(wins) // 2], 4),
                      "note": f"{len(wins)} back-to-back timed windows of {args.steps} steps each, every one "
                              "bracketed by barrier + device sync; `value` / `ms_per_step` are window 1",
                      "lanes": S, "stagger_ms": round(stagger["ms"], 3), "sustained_prime_s": primed["s"],
                      "sampler_error": getattr(sampler, "error", None),
                      "one_proof_alone_ms_before_the_run": None if single_ms is None else round(single_ms, 3),
                      "self_launched": bool(os.environ.get("TS_BENCH_SELF_LAUNCHED"))},
            "shard_stages_ms_per_rank": shard_stages,
        }
    if env.rank == 0 and not args.headline_only:
        out.update(rank0_legs(ts, args, env, ctx, config, cair, lanes, S, pool if S > 1 else None, local_sync,
                              make_trace, pis, cfg, n, w, qd, res, sharded, start_gate, proof,
                              gate_ms=stagger["ms"], stream_ok=(not sharded and S > 1 and stream_ok)))
        cb = out.get("cpu_baseline") or {}
        # the two fields VERDICT r3 asked for, at the top level of the line
        out["proof_blake3"] = cb.get("gpu_proof_blake3")
        out["matches_oracle"] = cb.get("matches_oracle")
        try:
            ps_ = out["clocks"]["prover_sustained"]
            out["headline_windows"]["sustained_leg"] = {"steps": ps_["steps"], "ms_per_step": ps_["ms_per_step"],
                                                        "gfxclk_mhz_median": ps_["gfxclk_mhz_median"],
                                                        "socket_power_w_median": ps_["socket_power_w_median"]}
        except Exception:  # noqa: BLE001
            pass
    phase_done("rank-0 legs (kernel timers, latency, h2d, clocks, cpu_baseline)")

    # ---- N > 1: BASELINE configs 4 and 5 as one sharded proof each, in the same lease
    wd = None
    force_block = bool(os.environ.get("TS_BENCH_FORCE_SHARD_BLOCK")) and env.world == 1 and not sharded
    if force_block:
        # Rehearsal of the N > 1 blocks' NATIVE-RCCL branch on a one-GPU box: a process group of one
        # nccl rank, so that the unique-id exchange, ts_comm_rccl_create, every variant and the gathers of
        # stage / collective tables run exactly as they will on a node -- everything but a second rank.
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29513", rank=0, world_size=1,
                                device_id=torch.device("cuda", dev))
        env.dist, env.device = dist, f"cuda:{dev}"
    if (env.world > 1 or force_block) and not sharded and not args.no_sharded_block:
        for c, _, _ in lanes[1:]:
            c.synchronize()
        mats = None
        limit = float(os.environ.get("TS_BENCH_SHARD_TIMEOUT_S", "200"))
        names = [x.strip() for x in os.environ.get("TS_BENCH_SHARD_BLOCKS", "sharded_config4,sharded_config5").split(",")
                 if x.strip() in SHARDED_BLOCKS]
        state = {}

        def on_timeout(phase):
            # A rank stuck in a collective (these have never run on > 1 GPU before the first lease):
            # rank 0 prints the record it has, with the error in it, then EVERY rank leaves non-zero
            # so that the launcher and the driver see the hang (never exit 0 from here).
            if env.rank == 0 and out is not None:
                for nm in names:
                    out.setdefault(nm, {"error": f"no result after {limit:.0f} s (stuck in: {phase}); the "
                                                 "replicas measurement above is unaffected"})
                emit(out)
            os._exit(3)

        env.dist.barrier()  # rank 0 comes here later than the others (its roofline legs): start the clocks together
        wd = Watchdog(limit, on_timeout)  # stays armed until the process group is closed
        for nm in names:
            try:
                blk = sharded_block(nm, ts, env, ctx, dev, wd, state)
            except Exception as e:  # noqa: BLE001 -- the headline must survive a failure here
                import traceback
                blk = {"error": repr(e), "traceback": traceback.format_exc()[-1500:]}
                # tell the peers: a rank that failed must not leave them waiting in a collective
                try:
                    if state.get("native") and state.get("comm") is not None:
                        cc = state["comm"].c
                        if cc.abort:
                            cc.abort(cc.user)  # ncclCommAbort: pending collectives on the peers fail
                        state["comm"].close()
                        state.pop("comm")
                except Exception:  # noqa: BLE001
                    pass
                if out is not None:
                    out[nm] = blk
                break
            if out is not None:
                out[nm] = blk
            phase_done(nm)
        if state.get("native") and state.get("comm") is not None:
            state["comm"].close()

    if sharded and env.dist is None:
        import torch.distributed as dist
        dist.destroy_process_group()
    env.close()
    if wd is not None:
        wd.done()
    if out is not None:
        emit(out)


# Live counter passes run `rocprofv3 -- python tools/prof_prove.py` as children.  On a timeout the whole
# process GROUP is killed (the profiled python is a grandchild that would otherwise keep the pipes and
# its GPU context), the wait after the kill is bounded, and the passes share one time budget.
_PMC_BUDGET = {"left_s": float(os.environ.get("TS_BENCH_PMC_BUDGET_S", "90"))}


def run_child_bounded(cmd, cwd, env, timeout_s: float):
    """(returncode, stdout, stderr) or raises TimeoutError; never blocks past timeout_s + 5 s."""
    import signal
    import subprocess

    timeout_s = min(timeout_s, _PMC_BUDGET["left_s"])
    if timeout_s <= 1:
        raise TimeoutError("live counter passes: time budget (TS_BENCH_PMC_BUDGET_S) used up")
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout_s)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        try:
            p.communicate(timeout=5)
        except Exception:  # noqa: BLE001
            pass
        raise TimeoutError(f"child still running after {timeout_s:.0f} s: process group killed")
    finally:
        _PMC_BUDGET["left_s"] -= time.perf_counter() - t0


def live_pmc_traffic(workload: str, kernel: str, timeout_s: float = 150.0):
    """HBM-side traffic of `kernel` measured NOW: two child processes, `rocprofv3 --pmc FETCH_SIZE` and
    `rocprofv3 --pmc WRITE_SIZE` (separate passes, counters only, the program itself after `--`) around
    tools/prof_prove.py (2 proofs of the workload); KiB units; FETCH_SIZE doubled for gfx950
    (/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).  Returns (bytes per launch, note)
    or (None, reason).  The parent keeps its contexts; the child is an ordinary second process on the GPU."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    # the names rocprofv3 reports are the demangled C++ ones; the library's timer names drop "ts::" and
    # shorten the leaf kinds (tools/pmc_summary.py: short)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    tmp = tempfile.mkdtemp(prefix="ts_pmc_", dir="/tmp")
    tot, launches = {}, 0
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable or "/usr/bin/python3", os.path.join(ROOT, "tools", "prof_prove.py"), "2", workload]
            env = dict(os.environ, TMPDIR="/tmp")
            env.pop("TS_BENCH_SELF_LAUNCHED", None)
            rc, r_out, r_err = run_child_bounded(cmd, "/tmp", env, timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {rc}): {(r_err or r_out)[-200:]}"
            v, c = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != counter:
                    continue
                name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"])).replace("ts::", "")
                m = re.match(r"k_leaf_tree<(\d), (\w+)(<(true|false)> ?)?>", name)
                if m:
                    kind = {"StridedLeaf": "strided", "TableLeaf": "table", "EfPairLeaf": "ef_pairs",
                            "FriLeaf": "fri_fold" if m.group(4) == "true" else "fri_leaf"}.get(m.group(2), m.group(2))
                    name = f"k_leaf_tree<{m.group(1)},{kind}>"
                want = kernel.strip("()")
                if name == want or name.startswith(want.rstrip(">") + ",") or name.startswith(want.rstrip(">") + ">"):
                    v += float(row["Counter_Value"])
                    c += 1
            if c == 0:
                return None, f"no {counter} rows for {kernel}"
            tot[counter], launches = v, c
        per_launch = (2 * 1024 * tot["FETCH_SIZE"] + 1024 * tot["WRITE_SIZE"]) / launches
        return round(per_launch), (f"live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two child processes, "
                                   f"{launches} launches of {kernel} over 2 proofs; KiB units, FETCH_SIZE x2 for gfx950)")
    except Exception as e:  # noqa: BLE001 -- the leg is optional
        return None, repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_valu_instructions(workload: str, timeout_s: float = 150.0):
    """SQ_INSTS_VALU summed over ONE proof of the workload, measured now: a child `rocprofv3 --pmc
    SQ_INSTS_VALU` around tools/prof_prove.py (counters only).  The table builders, the selector kernel
    and the synthetic trace generator are not part of a proof.  Returns (wave-instructions, note) or
    (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="ts_sq_", dir="/tmp")
    try:
        n_proofs = 2  # the first builds the per-context tables; both are counted and halved
        cmd = [exe, "--pmc", "SQ_INSTS_VALU", "--output-format", "csv", "-d", tmp, "-o", "sq", "--",
               sys.executable or "/usr/bin/python3", os.path.join(ROOT, "tools", "prof_prove.py"), str(n_proofs), workload]
        env = dict(os.environ, TMPDIR="/tmp")
        rc, r_out, r_err = run_child_bounded(cmd, "/tmp", env, timeout_s)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if rc != 0 or not files:
            return None, f"rocprofv3 --pmc SQ_INSTS_VALU failed (rc {rc}): {(r_err or r_out)[-200:]}"
        tot = 0.0
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] != "SQ_INSTS_VALU":
                continue
            if any(x in row["Kernel_Name"] for x in ("k_build_", "k_trace_", "k_selectors")):
                continue
            tot += float(row["Counter_Value"])
        if tot <= 0:
            return None, "no SQ_INSTS_VALU rows"
        return tot / n_proofs, (f"live: rocprofv3 --pmc SQ_INSTS_VALU (a child process, {n_proofs} proofs, per proof; table "
                                "builders, selectors and the trace generator excluded)")
    except Exception as e:  # noqa: BLE001 -- the leg is optional
        return None, repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def latest_profile(pattern: str):
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return fs[-1] if fs else None


def rank0_legs(ts, args, env, ctx, config, cair, lanes, S, pool, local_sync, make_trace, pis, cfg, n, w, qd,
               res, sharded, start_gate=lambda: None, timed_proof=None, gate_ms=0.0, stream_ok=False) -> dict:
    """Everything on the record beside the headline: measured on rank 0 after the timed region."""
    import hashlib

    import numpy as np

    # ---- roofline leg: per-kernel HIP-event timings over 10 extra (untimed) proofs.  The GPU has been idle for a
    # few hundred ms by now (sampler child joined, window statistics, the verifier on the host) and starts from
    # its idle clock: timed cold, EVERY kernel of the leg read 5-9 % long (k_leaf_tree 0.371 ms here against
    # 0.348 by rocprofv3 on the same box).  So the lane first proves for ~0.25 s, untimed, as rocprofv3's
    # ten-proof trace does before the launches it averages.
    t_warm = time.perf_counter() + 0.25
    while time.perf_counter() < t_warm:
        ts.prove(config, cair, ts.BfChallenger(), make_trace(ctx), pis)
    reps = 10  # as many proofs as the committed rocprofv3 trace averages over (tools/prof_prove.py 10)
    extra = [make_trace(ctx) for _ in range(reps)]
    ctx.synchronize()
    ctx.set_kernel_timing(True)
    for m in extra:
        ts.prove(config, cair, ts.BfChallenger(), m, pis)
    kt = ctx.take_kernel_timings()
    ctx.set_kernel_timing(False)
    # stage timers: five proofs, the MEDIAN of every stage (one sample showed a host hiccup as a 0.67-ms
    # query phase in the driver's r05 record)
    ctx.set_timing(True)
    stage_runs = []
    for _ in range(5):
        ts.prove(config, cair, ts.BfChallenger(), make_trace(ctx), pis)
        one = {}
        for k, v in ctx.take_timings():  # a stage name can occur twice (trace commit, quotient commit)
            one[k] = one.get(k, 0.0) + v
        stage_runs.append(one)
    ctx.set_timing(False)
    # one proof alone on the GPU, no timers inside (the stage timers synchronise at every stage
    # boundary): host wall clock around ts_prove, which returns with the proof on the host
    lat = []
    for _ in range(5):
        m_ = make_trace(ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        ts.prove(config, cair, ts.BfChallenger(), m_, pis)
        lat.append(1e3 * (time.perf_counter() - t0))
    single_latency = sorted(lat)[len(lat) // 2]
    stage_sum = {k: round(sorted(r.get(k, 0.0) for r in stage_runs)[len(stage_runs) // 2], 3) for k in stage_runs[0]}
    alg = algorithmic_bytes_per_proof(n, w, cfg[0], qd)

    leaf_tree_path = any("k_leaf_tree" in k for k in kt)

    def alg_bytes(name):  # every instantiation of the strided NTT pass moves the same bytes
        if leaf_tree_path and name in ("k_fri_fold_pairs", "k_merkle_tree"):
            # round-5 Merkle path: what is left to these two is the fold into the tail's 1024 elements and
            # the top of a tree above a few thousand sub-roots (latency, kilobytes): no bandwidth figure
            return None
        if "k_lde_mid" in name:
            return alg["k_lde_mid<1>"]
        if name == "k_merkle_level<1>" and "k_merkle_level<2>" not in kt:
            return alg["k_merkle_level<1>"] + alg["k_merkle_level<2>"]
        if "k_intt_contig" in name:  # with the fused transpose the trace and the chunk columns are separate launches
            fused = any("k_intt_contig" in k and "true" in k for k in kt)
            if "true" in name:
                return 8 * n * w
            return 8 * n * 4 * qd if fused else alg["k_intt_contig"]
        if name in alg:
            return alg[name]
        stem = name.strip("()").split("<")[0]  # "k_lde_fwd_contig<14, 4>" -> "k_lde_fwd_contig"
        return alg.get(stem)

    per_kernel = {}
    for name, (cnt, ms) in kt.items():
        ms_pp = ms / reps
        b = alg_bytes(name)
        per_kernel[name] = {
            "launches_per_proof": cnt / reps, "ms_per_proof": round(ms_pp, 4),
            "avg_launch_ms": round(ms / cnt, 5),
            "alg_gbps": round(b / (ms_pp * 1e-3) / 1e9, 1) if b and ms_pp > 0 else None}
    dom = max(kt.items(), key=lambda kv: kv[1][1])[0]
    dom_ms_pp = kt[dom][1] / reps
    achieved = (alg_bytes(dom) or 0) / (dom_ms_pp * 1e-3)
    # HBM traffic of that kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled for gfx950:
    # profiles/*_pmc_traffic.json, made from tools/prof_prove.py); bytes per launch, or null
    # NOT measured in this run (PMC collection needs rocprofv3 around the process): the file it
    # comes from and that file's hash are reported beside it
    traffic, traffic_source = None, None
    try:
        import glob
        tag = {"config3": "", "config2": "config2_", "config4": "config4_"}.get(args.workload)
        pmc_files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))
                           if tag is not None and (("config" in os.path.basename(f)) == bool(tag))
                           and (not tag or tag in os.path.basename(f)))
        if pmc_files and args.log_n == (22 if args.workload == "config4" else 20):
            ks = json.load(open(pmc_files[-1]))["kernels"]
            stem = dom.strip("()").rstrip(">")  # "k_lde_mid<1" matches "k_lde_mid<1, 8192, 512>"
            pk = ks.get(dom) or next((v for k, v in ks.items() if k.startswith(stem)), None)
            if pk:
                traffic = round((pk["fetch_bytes_per_proof_corrected"] + pk["write_bytes_per_proof"])
                                / pk["launches_per_proof"])
                traffic_source = {
                    "file": os.path.relpath(pmc_files[-1], ROOT),
                    "sha256": hashlib.sha256(open(pmc_files[-1], "rb").read()).hexdigest(),
                    "note": "static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run "
                            "(tools/prof_prove.py), not collected in this run"}
    except Exception:
        traffic, traffic_source = None, None
    # ... unless it can be measured right here (default; TS_BENCH_LIVE_PMC=0 keeps the committed figure):
    # the committed file then stays on the record as the figure it is compared with
    if os.environ.get("TS_BENCH_LIVE_PMC", "1") != "0" and args.log_n == (22 if args.workload == "config4" else 20):
        t_live0 = time.perf_counter()
        live, note = live_pmc_traffic(args.workload, dom)
        if live is not None:
            traffic_source = {"note": note, "seconds": round(time.perf_counter() - t_live0, 1),
                              "committed_figure": {"bytes_per_launch": traffic, **(traffic_source or {})}}
            traffic = live
        elif traffic_source is not None:
            traffic_source["live_attempt"] = note
    lpp = kt[dom][0] / reps
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved / 1e9, 2),
                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                "traffic": traffic, "traffic_source": traffic_source,
                "alg_bytes_per_launch": round((alg_bytes(dom) or 0) / lpp) if lpp else None,
                "avg_launch_ms": round(kt[dom][1] / kt[dom][0], 5),
                "launches_per_proof": kt[dom][0] / reps,
                "alg_bytes_per_proof": alg_bytes(dom),
                "kernel_ms_total_per_proof": round(sum(v[1] for v in kt.values()) / reps, 3)}
    # The kernel rounds 1-4 reported here was the forward NTT pass; since round 5 the trace tree and the
    # chunk tree run under ONE kernel name (k_leaf_tree<2,strided>, two launches) whose sum is a little
    # larger.  The NTT pass stays on the record beside it: same definition, its own figures.
    roofline_ntt = None
    ntt_name = next((k for k in kt if "k_lde_fwd_contig" in k), None)
    if ntt_name and ntt_name != dom:
        ms_pp_ = kt[ntt_name][1] / reps
        ab_ = alg_bytes(ntt_name) or 0
        roofline_ntt = {"bound": "hbm", "kernel": ntt_name, "achieved": round(ab_ / (ms_pp_ * 1e-3) / 1e9, 2),
                        "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(ab_ / (ms_pp_ * 1e-3) / HBM_PEAK, 4),
                        "avg_launch_ms": round(kt[ntt_name][1] / kt[ntt_name][0], 5),
                        "launches_per_proof": kt[ntt_name][0] / reps, "alg_bytes_per_proof": ab_}
    # ---- the stage and the whole job against the same roofline (SURVEY.md section 8(d) terms)
    N_ = n << cfg[0]
    wall_ = w + 4 * qd
    A1 = 4 * n * w + 4 * N_ * w            # trace LDE: read n x w, write N x w
    C1 = 16 * n * qd + 16 * N_ * qd        # chunk LDE
    whole = (A1 + 32 * (2 * N_ - 1) + (4 * n * qd * w + 16 * n * qd) + C1 + 32 * (2 * N_ - 1)
             + (4 * N_ * w + 16 * N_ * qd + 16 * N_) + (32 * N_ + 16 * N_ + 64 * N_))
    lde_ms = stage_sum.get("coset_lde")
    roofline_stage = None
    if lde_ms:
        roofline_stage = {"stage": "coset_lde (trace + quotient chunks)", "alg_bytes": A1 + C1,
                          "ms": lde_ms, "achieved": round((A1 + C1) / (lde_ms * 1e-3) / 1e9, 1),
                          "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                          "frac": round((A1 + C1) / (lde_ms * 1e-3) / HBM_PEAK, 4),
                          "note": "single proof alone on the GPU; the stage is three passes (60 n W bytes "
                                  "moved for 20 n W algorithmic) and VALU-bound, see alu_ceiling"}
    # every term of SURVEY.md section 8(d) against the stage timers of ONE proof alone on the GPU
    # (spans named as the reference's tracing spans; leaf hashing is not fused into the forward
    # pass, so the two Merkle terms carry the re-read of what they hash: + 4 N w and + 16 N qd)
    A2 = C2 = 32 * (2 * N_ - 1)
    B_ = 4 * n * qd * w + 16 * n * qd
    D_ = 4 * N_ * w + 16 * N_ * qd + 16 * N_
    E_ = 32 * N_ + 16 * N_ + 64 * N_
    terms = [("A1 + C1: coset_lde (trace + quotient chunks)", A1 + C1, ("coset_lde",), None),
             ("A2 + C2: merkle_commit (both trees, leaves re-read)", A2 + C2 + 4 * N_ * w + 16 * N_ * qd,
              ("merkle_commit",), "SURVEY's fused figure: %d" % (A2 + C2)),
             ("B: compute quotient polynomial", B_, ("compute quotient polynomial",), None),
             ("D: open (opened values + reduce rows)", D_,
              ("compute opened values with Lagrange interpolation", "reduce rows"),
              "the barycentric pass reads the low coset once more: + %d" % (4 * n * wall_)),
             ("E: FRI commit phase + query phase", E_, ("FRI commit phase", "query phase"), None)]
    roofline_stages = []
    for label, nbytes, spans, note in terms:
        ms_ = sum(stage_sum.get(sp, 0.0) for sp in spans)
        if ms_ > 0:
            roofline_stages.append({"stage": label, "alg_bytes": nbytes, "ms": round(ms_, 4),
                                    "achieved": round(nbytes / (ms_ * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9,
                                    "unit": "GB/s", "frac": round(nbytes / (ms_ * 1e-3) / HBM_PEAK, 4),
                                    **({"note": note} if note else {})})
    roofline_whole = {"alg_bytes_per_proof": whole, "ms_per_step": round(res["ms_per_step"], 4),
                      "achieved": round(whole / (res["ms_per_step"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9,
                      "unit": "GB/s", "frac": round(whole / (res["ms_per_step"] * 1e-3) / HBM_PEAK, 4)}
    # ---- the integer-ALU ceiling, measured in this run with the library's own arithmetic
    log_n_ = n.bit_length() - 1
    butterflies = wall_ * (n // 2) * log_n_ * (1 + (1 << cfg[0]))
    compressions = N_ * ((4 * w + 63) // 64) + N_ * ((16 * qd + 63) // 64) + N_ + 3 * N_

    def ms_of(*names):
        return sum(v["ms_per_proof"] for k, v in per_kernel.items() if any(x in k for x in names))
    bf_peak, b3_peak = ctx.alu_ceiling(0), ctx.alu_ceiling(1)
    ntt_ms = ms_of("k_intt_contig", "k_lde_mid", "k_lde_fwd_contig")
    b3_ms = ms_of("k_leaf_hash", "k_merkle_level", "k_merkle_tree", "k_fri_round", "k_leaf_tree")
    alu_ceiling = {
        "butterflies_per_s_peak": round(bf_peak), "butterflies_per_proof": butterflies,
        "ntt_kernels_ms": round(ntt_ms, 4), "ntt_ms_at_peak": round(butterflies / bf_peak * 1e3, 4),
        "ntt_frac_of_alu_peak": round(butterflies / bf_peak * 1e3 / ntt_ms, 4) if ntt_ms else None,
        "blake3_compressions_per_s_peak": round(b3_peak), "blake3_compressions_per_proof": compressions,
        "merkle_kernels_ms": round(b3_ms, 4), "merkle_ms_at_peak": round(compressions / b3_peak * 1e3, 4),
        "merkle_frac_of_alu_peak": round(compressions / b3_peak * 1e3 / b3_ms, 4) if b3_ms else None,
        "note": "peaks from ts_bench_alu (register-resident loops of the same butterfly / compression code, "
                "no memory traffic); merkle_kernels_ms also holds the folds that the FRI round launches carry "
                "(k_fri_round, k_leaf_tree<*,fri_fold>), whose arithmetic is not in the compression count"}
    # ---- the ceiling that actually binds: VALU instruction issue.  A wave64 VALU instruction
    # holds its SIMD for four cycles (tools/pmc_alu.sh: the register-resident loops issue exactly
    # one per 4 cycles per SIMD at 2.30-2.34 GHz); the prover's kernels run at ~2.0 GHz
    # (SQ_BUSY_CYCLES over their durations).  SQ_INSTS_VALU summed over one proof comes from a
    # committed rocprofv3 --pmc pass (static, like roofline.traffic): tools/pmc_sq.sh.
    valu_issue = None
    try:
        import re as _re
        sqf = latest_profile(f"r*_{args.workload}_sq_counters.txt")
        if sqf and args.log_n == (22 if args.workload == "config4" else 20):
            m_ = _re.search(r"whole proof: SQ_INSTS_VALU ([0-9.e+]+)", open(sqf).read())
            insts = float(m_.group(1))
            simds, clock = 4 * ctx.num_cus if hasattr(ctx, "num_cus") else 1024, 2.0e9
            floor_ms = insts * 4 / simds / clock * 1e3
            valu_issue = {
                "wave_instructions_per_proof": insts, "simds": simds, "cycles_per_instruction": 4,
                "clock_hz_under_load": clock, "ms_per_proof_at_ceiling": round(floor_ms, 4),
                "frac_of_ceiling": round(floor_ms / res["ms_per_step"], 4),
                "source": {"file": os.path.relpath(sqf, ROOT),
                           "sha256": hashlib.sha256(open(sqf, "rb").read()).hexdigest(),
                           "note": "static: SQ_INSTS_VALU of an earlier rocprofv3 --pmc pass, not collected "
                                   "in this run; clock = SQ_BUSY_CYCLES / kernel durations of the same pass"}}
    except Exception:
        valu_issue = None
    # ... measured right here where rocprofv3 can run as a child (TS_BENCH_LIVE_PMC=0: the committed figure)
    if os.environ.get("TS_BENCH_LIVE_PMC", "1") != "0" and args.log_n == (22 if args.workload == "config4" else 20):
        t_sq0 = time.perf_counter()
        insts_live, note = live_valu_instructions(args.workload)
        if insts_live is not None:
            simds, clock = 1024, 2.0e9
            floor_ms = insts_live * 4 / simds / clock * 1e3
            valu_issue = {
                "wave_instructions_per_proof": round(insts_live), "simds": simds, "cycles_per_instruction": 4,
                "clock_hz_under_load": clock, "ms_per_proof_at_ceiling": round(floor_ms, 4),
                "frac_of_ceiling": round(floor_ms / res["ms_per_step"], 4),
                "source": {"note": note, "seconds": round(time.perf_counter() - t_sq0, 1),
                           "committed_figure": None if valu_issue is None else
                           {"wave_instructions_per_proof": valu_issue["wave_instructions_per_proof"],
                            **valu_issue["source"]}}}
        elif valu_issue is not None:
            valu_issue["source"]["live_attempt"] = note

    # ---- the rate a caller sees who hands over HOST traces (never `value`): pinned buffer,
    # asynchronous upload on each lane's stream, the upload of one lane overlapping the proofs of the others
    def h2d_leg(lanes_, make_trace_, pis_, n_, w_):
        try:
            host = make_trace_(lanes_[0][0]).download()
            pins = []
            for _ in lanes_:  # one page-locked buffer per lane, as a host with S trace generators would have
                pins.append(ts.PinnedHostMatrix(n_, w_))
                pins[-1].array[:] = host
            del host
            k2 = 6 * len(lanes_)

            def h2d_job(l):
                c, conf, ca = lanes_[l]
                pin = pins[l]
                for _ in range(k2 // len(lanes_)):
                    start_gate()  # (gate before the upload: 5.4-5.5 ms/step; after it: 6.3-6.5, same box)
                    m = ts.DeviceMatrix.upload_async(c, pin)
                    ts.prove(conf, ca, ts.BfChallenger(), m, pis_)
            list(pool.map(h2d_job, range(len(lanes_))))  # warm-up
            local_sync()
            t0 = time.perf_counter()
            list(pool.map(h2d_job, range(len(lanes_))))
            local_sync()
            dt_h = time.perf_counter() - t0
            return {"ms_per_step": round(1e3 * dt_h / k2, 4), "steps": k2,
                    "h2d_GB_per_s": round(n_ * w_ * 4 * k2 / dt_h / 1e9, 1),
                    "note": "every step uploads its trace from page-locked host memory inside the timed "
                            "region (hipMemcpyAsync on the lane's stream); PCIe Gen5 x16 bounds it"}
        except Exception as e:  # never let the extra leg take the headline down
            return {"error": repr(e)}

    h2d = h2d_c2 = None
    if env.world == 1 and not sharded and not args.host_traces and pool is not None:
        if args.workload in ("config3", "config2"):
            h2d = h2d_leg(lanes, make_trace, pis, n, w)
        if args.workload == "config3":
            # the same leg for BASELINE config 2 (Fibonacci 2^20 x 2): the other shape a caller of the
            # reference's fib_air test would hand over; with its device-resident rate beside it
            try:
                air2, _, pis2, desc2, cfg2, (n2, w2), make2 = workload("config2", 20, False)
                pis2 = np.array([0, 1, pis2(ctx)], dtype=np.uint32)
                lanes2 = [(c, ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg2), c)),
                           ts.CompiledAir(c, ts.air_tape(air2, len(pis2)))) for c, _, _ in lanes]
                k2 = 6 * len(lanes2)
                mats2 = [make2(lanes2[i % len(lanes2)][0]) for i in range(2 * k2)]

                def dev_job(first):
                    def job(l):
                        c, conf, ca = lanes2[l]
                        for i in range(first + l, first + k2, len(lanes2)):
                            start_gate()
                            ts.prove(conf, ca, ts.BfChallenger(), mats2[i], pis2)
                    return job
                list(pool.map(dev_job(0), range(len(lanes2))))
                local_sync()
                t0 = time.perf_counter()
                list(pool.map(dev_job(k2), range(len(lanes2))))
                local_sync()
                dev_ms = 1e3 * (time.perf_counter() - t0) / k2
                h2d_c2 = h2d_leg(lanes2, make2, pis2, n2, w2)
                h2d_c2["workload"] = desc2
                h2d_c2["device_resident_ms_per_step"] = round(dev_ms, 4)
            except Exception as e:
                h2d_c2 = {"error": repr(e)}
    # ---- clocks and power under this load: a sustained run of the same workload (all lanes, start
    # gate on) sampled with amdsmi every 10 ms, beside an idle sample and the pure-ALU loop -- so that
    # "the chip runs this path at ~2.0 GHz, the register-resident loops at ~2.3" is a measurement
    clocks = None
    if env.world == 1 and not sharded and pool is not None:
        try:
            from tapstark_amd.benchutil import GpuSampler
            smp = GpuSampler(0, 0.01)
            if smp.error:
                clocks = {"error": smp.error}
            else:
                with smp:
                    time.sleep(0.3)
                idle = smp.summary()
                k3 = 40 * len(lanes)  # ~0.45 s at the headline shape
                mats3 = [make_trace(lanes[i % len(lanes)][0]) for i in range(k3)]

                def clk_job(l):
                    c, conf, ca = lanes[l]
                    for i in range(l, k3, len(lanes)):
                        start_gate()
                        ts.prove(conf, ca, ts.BfChallenger(), mats3[i], pis)
                local_sync()
                with smp:
                    t0 = time.perf_counter()
                    if stream_ok:  # the same driver as the timed windows
                        ts.prove_stream([(conf, ca) for _, conf, ca in lanes], mats3,
                                        [i % len(lanes) for i in range(k3)], pis, gate_ms=gate_ms, want_times=False)
                    else:
                        list(pool.map(clk_job, range(len(lanes))))
                    local_sync()
                    dt_c = time.perf_counter() - t0
                load = smp.summary()
                with smp:
                    t_end = time.perf_counter() + 0.4
                    while time.perf_counter() < t_end:
                        ctx.alu_ceiling(0)
                alu = smp.summary()
                clocks = {"idle": idle, "prover_sustained": dict(load, ms_per_step=round(1e3 * dt_c / k3, 4), steps=k3),
                          "alu_loop_butterflies": alu,
                          "note": "amdsmi gpu_metrics (current_gfxclks averaged over the 8 XCDs, current_socket_power), "
                                  "sampled from a host thread every 10 ms, outside the timed region"}
        except Exception as e:  # noqa: BLE001
            clocks = {"error": repr(e)}
    # the VALU-issue ceiling at the clock MEASURED in this run (the static 2.0 GHz above came from
    # SQ_BUSY_CYCLES of an earlier counter pass)
    try:
        mhz = clocks["prover_sustained"]["gfxclk_mhz_median"]
        if valu_issue and mhz:
            fl = valu_issue["wave_instructions_per_proof"] * 4 / valu_issue["simds"] / (mhz * 1e6) * 1e3
            valu_issue["measured_clock"] = {
                "gfxclk_mhz_median_under_this_load": mhz, "ms_per_proof_at_ceiling": round(fl, 4),
                "frac_of_ceiling": round(fl / res["ms_per_step"], 4),
                "socket_power_w_median": clocks["prover_sustained"]["socket_power_w_median"],
                "note": "amdsmi sample of a sustained run of the same workload in this process (`clocks`); the "
                        "register-resident ALU loops hold the clock in `clocks.alu_loop_butterflies`"}
    except Exception:  # noqa: BLE001
        pass
    # energy: under the power cap this, not the issue fill, is what a step costs
    try:
        ps = clocks["prover_sustained"]
        idle_w = clocks["idle"]["socket_power_w_median"]
        clocks["energy_per_proof_j"] = round(ps["socket_power_w_median"] * ps["ms_per_step"] * 1e-3, 3)
        clocks["energy_per_proof_above_idle_j"] = round((ps["socket_power_w_median"] - idle_w) * ps["ms_per_step"] * 1e-3, 3)
        clocks["nj_per_trace_cell"] = round(clocks["energy_per_proof_j"] / float(n * w) * 1e9, 2)
    except Exception:  # noqa: BLE001
        pass
    # (the contract: timed on rank 0 at N = 1 only)
    same = args.workload == "config3" and args.log_n == 20 and timed_proof is not None
    cpu = None if (args.no_cpu_baseline or env.world > 1) else cpu_baseline(timed_proof.words if same else None)
    return {
        # one proof alone on the GPU, HIP events around ts_prove (the `value` above keeps
        # several in flight; this is the latency a single caller sees)
        "single_proof_latency_ms": round(single_latency, 4),
        "single_proof_latency_with_stage_timers_ms": stage_sum.get("prove"),
        "roofline": roofline, "roofline_ntt": roofline_ntt, "roofline_stage": roofline_stage,
        "roofline_stages": roofline_stages,
        "roofline_whole": roofline_whole,
        "alu_ceiling": alu_ceiling, "valu_issue": valu_issue, "clocks": clocks, "h2d_inclusive": h2d,
        "h2d_inclusive_config2": h2d_c2,
        "h2d_inclusive_ms_per_step": (h2d or {}).get("ms_per_step"), "cpu_baseline": cpu,
        "stages_ms": stage_sum,
        "kernels": per_kernel,
    }


if __name__ == "__main__":
    main()
