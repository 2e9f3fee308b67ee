"""One proof sharded over several ranks: bit-identical to prove().

The GPU box has ONE card: the ranks of a multi-rank case are separate processes that share it and
talk over gloo (host-staged collectives, the CPU rehearsal of the RCCL path); the nccl (= RCCL)
callback path itself -- zero-copy wrapping of the library's device buffers, collectives enqueued
behind the library's HIP stream -- runs with a world of one rank.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import tapstark_amd as ts

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _sharded_worker import make_case  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    from tapstark_amd.build import build

    build()
    return ts.default_context()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(tmp_path, spec):
    spec = dict(spec, port=free_port(), out=str(tmp_path / "proof"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(spec["world"]):
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_worker.py"),
                                       json.dumps(spec)], env=dict(env, RANK=str(r)),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    proofs = [np.load(f"{spec['out']}.rank{r}.npy") for r in range(spec["world"])]
    metas = [json.load(open(f"{spec['out']}.rank{r}.json")) for r in range(spec["world"])]
    return proofs, metas


def single_gpu_proof(ctx, spec):
    air, trace, pis = make_case(spec["air"], spec["log_n"])
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*spec["cfg"]), ctx))
    ch = ts.BfChallenger()
    proof = ts.prove(config, air, ch, trace, pis)
    ts.verify(config, air, ts.BfChallenger(), proof, pis)
    return proof, ch.sample_bits(20), (air, pis)


CASES = [
    # air, log_n, (log_blowup, queries, pow), world, min_local_log
    ("fib", 8, (1, 6, 8), 2, 1),
    ("fib", 10, (2, 9, 8), 2, 3),
    ("fib", 10, (2, 9, 8), 4, 1),
    ("mul7", 9, (2, 7, 8), 4, 2),
    ("mul64", 10, (2, 28, 8), 4, 4),
    ("mul64", 11, (3, 16, 8), 4, 12),   # two cosets per rank; every FRI round replicated
    ("mul64", 13, (2, 28, 8), 2, 10),   # two-pass NTT (n > 4096), cosets split 2 + 2
    ("ext25", 8, (2, 5, 8), 4, 1),
]


@pytest.mark.parametrize("air,log_n,cfg,world,mll", CASES,
                         ids=[f"{c[0]}-2p{c[1]}-b{c[2][0]}-G{c[3]}-m{c[4]}" for c in CASES])
def test_sharded_proof_bit_identical(ctx, orc, tmp_path, air, log_n, cfg, world, mll):
    spec = {"air": air, "log_n": log_n, "cfg": list(cfg), "world": world, "backend": "gloo",
            "min_local_log": mll}
    want, want_bits, (air_obj, pis) = single_gpu_proof(ctx, spec)
    proofs, metas = run_ranks(tmp_path, spec)
    for r, p in enumerate(proofs):
        assert len(p) == len(want.words), f"rank {r}: proof length"
        assert (p == want.words).all(), f"rank {r}: {int((p != want.words).sum())} words differ"
        # the transcript ends in the same state on every rank (prover.rs takes &mut challenger)
        assert metas[r]["chal_bits"] == want_bits
    tape = ts.air_tape(air_obj, len(pis))
    assert orc.verify(orc.FriConfig(*cfg), tape, proofs[0], pis) == 0
    # exchange steps: trace all-gather + 2 commits + sharded FRI rounds + FRI vector + answers
    assert metas[0]["calls"]["all_gather"] >= 5
    assert metas[0]["calls"]["broadcast"] == 1 << ts.CompiledAir(ctx, tape).log_quotient_degree


def test_sharded_with_replicated_trace(ctx, orc, tmp_path):
    # every rank already holds the whole trace (device-generated in production): one collective
    # fewer -- the bulk one -- and the same proof
    spec = {"air": "mul64", "log_n": 10, "cfg": [2, 28, 8], "world": 4, "backend": "gloo",
            "min_local_log": 4, "replicated": True}
    want, want_bits, _ = single_gpu_proof(ctx, spec)
    proofs, metas = run_ranks(tmp_path, spec)
    sliced, sliced_metas = run_ranks(tmp_path, dict(spec, replicated=False))
    for r in range(4):
        assert (proofs[r] == want.words).all() and (sliced[r] == want.words).all()
    assert metas[0]["calls"]["all_gather"] == sliced_metas[0]["calls"]["all_gather"] - 1
    assert metas[0]["calls"]["bytes"] < sliced_metas[0]["calls"]["bytes"]


def test_sharded_world_of_one_over_rccl(ctx, tmp_path):
    # nccl backend (RCCL): device buffers wrapped in place, collectives ordered on the HIP stream
    spec = {"air": "mul64", "log_n": 12, "cfg": [2, 28, 8], "world": 1, "backend": "nccl",
            "min_local_log": 6}
    want, want_bits, _ = single_gpu_proof(ctx, spec)
    proofs, metas = run_ranks(tmp_path, spec)
    assert (proofs[0] == want.words).all()
    assert metas[0]["chal_bits"] == want_bits


def test_sharded_rejects_partial_cosets(ctx, tmp_path):
    # G > 2^log_blowup would split cosets: TS_ERR_UNSUPPORTED, as documented
    spec = {"air": "fib", "log_n": 8, "cfg": [1, 4, 8], "world": 4, "backend": "gloo",
            "min_local_log": 1, "port": free_port(), "out": str(tmp_path / "x")}
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RANK="0")
    spec1 = dict(spec, world=1)
    # a world of one with a communicator that claims four ranks
    code = (
        "import sys, json, numpy as np, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {os.path.dirname(HERE)!r}); sys.path.insert(0, {HERE!r})\n"
        "import tapstark_amd as ts\n"
        "from tapstark_amd.dist import TorchComm\n"
        "from _sharded_worker import make_case\n"
        f"spec = json.loads({json.dumps(json.dumps(spec1))})\n"
        "dist.init_process_group('gloo', init_method=f\"tcp://127.0.0.1:{spec['port']}\", rank=0, world_size=1)\n"
        "ctx = ts.default_context()\n"
        "air, trace, pis = make_case('fib', 8)\n"
        "comm = TorchComm(0); comm.c.world = 4; comm.world = 4\n"
        "cfg = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(1, 4, 8), ctx))\n"
        "try:\n"
        "    ts.prove_sharded(cfg, air, ts.BfChallenger(), trace[:64], pis, comm)\n"
        "except Exception as e:\n"
        "    print('ERR', e); sys.exit(0 if 'TS_ERR_UNSUPPORTED' in str(e) else 3)\n"
        "sys.exit(4)\n"
    )
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


# ------------------------------------------------------------------ native communicators (csrc/comm.cpp)
def run_local_ranks(spec):
    """G ranks = G threads of THIS process, each with its own context on the one GPU, talking
    through the library's in-process communicator (ts_comm_local_*): no torch, no subprocesses, so
    eight ranks fit a box that allows six processes on its card."""
    import threading

    from tapstark_amd.comm import LocalCommGroup

    G = spec["world"]
    air, trace, pis = make_case(spec["air"], spec["log_n"])
    n = trace.shape[0]
    group = LocalCommGroup(G)
    proofs, bits, errors = [None] * G, [None] * G, [None] * G

    def rank_main(r):
        try:
            ctx = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*spec["cfg"]), ctx))
            rows = trace if spec.get("replicated") else np.ascontiguousarray(trace[r * n // G:(r + 1) * n // G])
            ch = ts.BfChallenger()
            if spec.get("timing"):
                ctx.set_timing(True)
            p = ts.prove_sharded(config, air, ch, rows, pis, group.comm(r), spec["min_local_log"],
                                 trace_replicated=bool(spec.get("replicated")))
            proofs[r], bits[r] = p.words, ch.sample_bits(20)
            if spec.get("timing"):
                spec.setdefault("stages", {})[r] = ctx.take_timings()
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    return proofs, bits, errors


LOCAL_CASES = [
    # air, log_n, (log_blowup, queries, pow), world, min_local_log, replicated
    ("mul64", 10, (3, 16, 8), 8, 4, False),    # one coset per rank
    ("mul64", 12, (4, 16, 8), 8, 6, False),    # config 4's split: 16 cosets, two per rank, row-sliced trace
    ("mul64", 13, (4, 16, 8), 8, 12, True),    # the same with a replicated trace and the two-pass NTT (n > 4096)
    ("ext25", 9, (4, 5, 8), 8, 2, False),
    ("fib", 11, (3, 9, 8), 8, 1, False),
    ("mul64", 11, (2, 28, 8), 4, 4, False),    # the headline config's FRI parameters over 4 ranks
    ("mul7", 22, (2, 7, 8), 4, 12, True),      # 2^22 rows: 16384-element NTT chunks, one coset per rank
    ("mul7", 21, (3, 7, 8), 4, 12, False),     # 2^21 rows: 8192-element chunks, two cosets per rank, row-sliced
]


@pytest.mark.parametrize("air,log_n,cfg,world,mll,repl", LOCAL_CASES,
                         ids=[f"{c[0]}-2p{c[1]}-b{c[2][0]}-G{c[3]}-m{c[4]}{'-repl' if c[5] else ''}"
                              for c in LOCAL_CASES])
def test_sharded_eight_ranks_native_comm(ctx, orc, air, log_n, cfg, world, mll, repl):
    spec = {"air": air, "log_n": log_n, "cfg": list(cfg), "world": world, "min_local_log": mll,
            "replicated": repl}
    want, want_bits, (air_obj, pis) = single_gpu_proof(ctx, spec)
    proofs, bits, errors = run_local_ranks(spec)
    for r in range(world):
        assert errors[r] is None, f"rank {r}: {errors[r]!r}"
        assert len(proofs[r]) == len(want.words), f"rank {r}: proof length"
        assert (proofs[r] == want.words).all(), f"rank {r}: {int((proofs[r] != want.words).sum())} words differ"
        assert bits[r] == want_bits
    tape = ts.air_tape(air_obj, len(pis))
    assert orc.verify(orc.FriConfig(*cfg), tape, proofs[0], pis) == 0


def test_sharded_failure_on_one_rank_does_not_hang_the_others(ctx):
    # a rank that throws between collectives must not leave its peers waiting.  Rank 2
    # is handed a trace slice of the wrong width: it fails its argument checks and aborts the
    # communicator; every other rank returns TS_ERR_COMM instead of blocking in the all-gather.
    import threading

    from tapstark_amd._lib import TsError
    from tapstark_amd.comm import LocalCommGroup

    G = 4
    air, trace, pis = make_case("mul7", 8)
    n = trace.shape[0]
    group = LocalCommGroup(G)
    errors = [None] * G

    def rank_main(r):
        try:
            c = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), c))
            rows = np.ascontiguousarray(trace[r * n // G:(r + 1) * n // G])
            if r == 2:
                rows = np.ascontiguousarray(rows[:, :5])
            ts.prove_sharded(config, ts.CompiledAir(c, ts.air_tape(air, 0)), ts.BfChallenger(), rows, pis,
                             group.comm(r), 2)
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    assert all(isinstance(e, TsError) for e in errors), errors
    assert errors[2].code == 1 and {errors[r].code for r in (0, 1, 3)} == {7}

    # The group is one-shot with respect to failure (include/tapstark.h): every later collective
    # fails until it is reset -- and after ts_comm_local_group_reset the SAME group proves again
    # (one bad proof must not kill the group for the rest of the process).
    errors2 = [None] * G

    def rank_poisoned(r):
        try:
            c = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), c))
            rows = np.ascontiguousarray(trace[r * n // G:(r + 1) * n // G])
            ts.prove_sharded(config, ts.CompiledAir(c, ts.air_tape(air, 0)), ts.BfChallenger(), rows, pis,
                             group.comm(r), 2)
        except BaseException as e:  # noqa: BLE001
            errors2[r] = e

    threads = [threading.Thread(target=rank_poisoned, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert all(isinstance(e, TsError) and e.code == 7 for e in errors2), errors2
    group.reset()
    config0 = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), ctx))
    want = ts.prove(config0, air, ts.BfChallenger(), trace, pis)
    proofs, errors3 = [None] * G, [None] * G

    def rank_good(r):
        try:
            c = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), c))
            rows = np.ascontiguousarray(trace[r * n // G:(r + 1) * n // G])
            proofs[r] = ts.prove_sharded(config, ts.CompiledAir(c, ts.air_tape(air, 0)), ts.BfChallenger(),
                                         rows, pis, group.comm(r), 2).words
        except BaseException as e:  # noqa: BLE001
            errors3[r] = e

    threads = [threading.Thread(target=rank_good, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads)
    for r in range(G):
        assert errors3[r] is None, f"rank {r} after reset: {errors3[r]!r}"
        assert (proofs[r] == want.words).all()


def test_local_group_timeout_is_configurable(ctx):
    # a peer that never arrives: the waiting rank gives up after the configured time (not 10 minutes)
    import time

    import torch

    from tapstark_amd.comm import LocalCommGroup

    group = LocalCommGroup(2)
    group.set_timeout(1)
    c = group.comm(0).c
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    out = torch.zeros(128, dtype=torch.uint8, device="cuda:0")
    t0 = time.time()
    rc = c.all_gather(c.user, buf.data_ptr(), out.data_ptr(), 64, ctx.stream)
    assert rc != 0 and time.time() - t0 < 10
    group.reset()


def test_rccl_native_comm_world_of_one(ctx, orc):
    # the C++ RCCL communicator (ncclAllGather / ncclBroadcast on the context's stream) with a
    # world of one: the only size a one-GPU box can run; G > 1 over xGMI is unverified
    from tapstark_amd import comm as tc

    if not tc.rccl_available():
        pytest.skip("librccl not loadable")
    spec = {"air": "mul64", "log_n": 12, "cfg": [2, 28, 8], "world": 1, "min_local_log": 6}
    want, want_bits, _ = single_gpu_proof(ctx, spec)
    air, trace, pis = make_case("mul64", 12)
    c = ts.Context(0)
    rc = tc.RcclComm(c, tc.rccl_unique_id(), 0, 1)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 28, 8), c))
    ch = ts.BfChallenger()
    p = ts.prove_sharded(config, air, ch, trace, pis, rc, 6)
    assert (p.words == want.words).all()
    assert ch.sample_bits(20) == want_bits
    info = rc.info()
    assert info["comm_count"] == 1 and info["comm_user_rank"] == 0 and info["world"] == 1, info
    assert info["rccl_version"] > 0 and info["aborted"] == 0
    rc.close()


def test_rccl_native_comm_collective_shapes_of_config4(ctx):
    """Every collective shape BASELINE config 4 puts on the native RCCL communicator at G = 8, run
    with the one world size a one-GPU box allows (1): the 128 MiB-per-rank trace all-gather, the
    16 n-byte quotient-chunk broadcast (64 MiB at n = 2^22), the 32-byte sub-root all-gathers, the
    FRI-tail and answered-query gathers (odd sizes, not multiples of 16).  What this can catch before
    a node exists: buffer-size / alignment / datatype errors, stream ordering with the context's own
    non-blocking stream.  What it cannot: anything about a second rank."""
    import torch

    from tapstark_amd import comm as tc

    if not tc.rccl_available():
        pytest.skip("librccl not loadable")
    c = ts.Context(0)
    rc = tc.RcclComm(c, tc.rccl_unique_id(), 0, 1)
    cc = rc.c
    stream = c.stream
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    for nbytes in (128 << 20, 64 << 20, 32, 8 * 32, 16 * 4096, 4 * 21877, 12345, 1):
        src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
        dst = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        assert cc.all_gather(cc.user, src.data_ptr(), dst.data_ptr(), nbytes, stream) == 0
        c.synchronize()
        assert torch.equal(src, dst), f"all_gather of {nbytes} bytes"
        keep = src.clone()
        torch.cuda.synchronize()
        assert cc.broadcast(cc.user, src.data_ptr(), nbytes, 0, stream) == 0
        c.synchronize()
        assert torch.equal(src, keep), f"broadcast of {nbytes} bytes"
    # unaligned device pointers (a slab that starts in the middle of a buffer)
    big = torch.randint(0, 256, (1 << 16,), dtype=torch.uint8, device="cuda:0", generator=g)
    out = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    assert cc.all_gather(cc.user, big.data_ptr() + 36, out.data_ptr() + 4, 1000, stream) == 0
    c.synchronize()
    assert torch.equal(big[36:1036], out[4:1004])
    rc.close()


def test_config4_as_specified_eight_ranks_full_shape(ctx, orc):
    """BASELINE config 4 as written: SynthMulAir-64, trace 2^22 x 64, log_blowup 4, 16 queries, ONE
    proof sharded over 8 ranks (two cosets each).  The ranks are threads on the box's one GPU (native
    in-process communicator); every rank generates the trace on the device (replicated input, as
    bench.py --mode sharded does).  All eight proofs equal the single-GPU proof byte for byte and the
    oracle's verifier accepts it."""
    import threading

    from tapstark_amd.airs import SynthMulAir
    from tapstark_amd.comm import LocalCommGroup

    n, G, cfg = 1 << 22, 8, (4, 16, 8)
    air = SynthMulAir(64)
    tape = ts.air_tape(air, 0)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    want = ts.prove(config, ts.CompiledAir(ctx, tape), ts.BfChallenger(), ts.DeviceMatrix.synth_mul(ctx, n, 64), [])
    assert orc.verify(orc.FriConfig(*cfg), tape, want.words, []) == 0
    group = LocalCommGroup(G)
    proofs, errors = [None] * G, [None] * G

    def rank_main(r):
        try:
            c = ts.Context(0)
            conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
            p = ts.prove_sharded(conf, ts.CompiledAir(c, tape), ts.BfChallenger(), ts.DeviceMatrix.synth_mul(c, n, 64),
                                 [], group.comm(r), trace_replicated=True)
            proofs[r] = p.words
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    for r in range(G):
        assert errors[r] is None, f"rank {r}: {errors[r]!r}"
        assert len(proofs[r]) == len(want.words) and (proofs[r] == want.words).all(), f"rank {r} differs"


# ------------------------------------------------------------------ BASELINE config 5 over 8 ranks
def _thread_ranks(G, rank_fn):
    import threading

    out, errors = [None] * G, [None] * G

    def main(r):
        try:
            out[r] = rank_fn(r)
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=main, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    for r in range(G):
        assert errors[r] is None, f"rank {r}: {errors[r]!r}"
    return out


@pytest.mark.parametrize("log_n,mode", [(10, "replicated"), (10, "sliced"), (12, "replicated"), (12, "sliced"),
                                        (13, "replicated"), (13, "sliced"), (14, "replicated")])
def test_config5_air_sharded_over_eight_ranks(ctx, orc, log_n, mode):
    """BASELINE config 5 ("... on 8 x MI355X", shape README.md:91,101): SynthExt-163 at log_blowup 4 /
    16 queries as ONE proof over 8 ranks (two cosets each; threads on the box's GPU over the native
    in-process communicator).  163 columns: the strided leaf hash runs on a slab with a ragged last
    block; 2^13 / 2^14 rows take the two-pass LDE.  Byte-identical to ts_prove AND to the oracle's proof."""
    from tapstark_amd.airs import SynthExtAir, generate_synth_ext_trace
    from tapstark_amd.comm import LocalCommGroup

    G, cfg, n = 8, (4, 16, 8), 1 << log_n
    air = SynthExtAir(163)
    tape = ts.air_tape(air, 0)
    trace = generate_synth_ext_trace(n, 163)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    ch = ts.BfChallenger()
    want = ts.prove(config, ts.CompiledAir(ctx, tape), ch, trace.copy(), [])
    oracle = orc.prove(orc.FriConfig(*cfg), tape, trace, [])
    assert len(oracle) == len(want.words) and (oracle == want.words).all()
    group = LocalCommGroup(G)
    sliced = mode.endswith("sliced")

    def rank(r):
        c = ts.Context(0)
        conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
        rows = trace[r * n // G:(r + 1) * n // G] if sliced else trace
        chal = ts.BfChallenger()
        p = ts.prove_sharded(conf, ts.CompiledAir(c, tape), chal, np.ascontiguousarray(rows), [], group.comm(r),
                             min_local_log=3, trace_replicated=not sliced)
        return p.words, chal.sample_bits(20)

    res = _thread_ranks(G, rank)
    bits = ch.sample_bits(20)
    for r, (words, b) in enumerate(res):
        assert len(words) == len(oracle) and (words == oracle).all(), f"rank {r}: proof differs from the oracle's"
        assert b == bits, f"rank {r}: transcript state differs"


# ------------------------------------------------------------------ local quotient
class PowAir(ts.BaseAir):
    """Test AIR: columns (a, c) with c = a^k (one constraint of degree k, a quotient that really has
    degree ~ (k - 1) n: every block Q_k of its coefficients is non-zero) and a' = a + 1."""

    def __init__(self, k):
        self.k = k

    def width(self):
        return 2

    def eval(self, builder):
        main = builder.main()
        local, nxt = main.row_slice(0), main.row_slice(1)
        acc = local[0]
        for _ in range(self.k - 1):
            acc = acc * local[0]
        builder.assert_zero(acc - local[1])
        builder.when_transition().assert_eq(local[0] + 1, nxt[0])


def pow_trace(n, k):
    a = [(7 + i) % 0x78000001 for i in range(n)]
    return np.array([[x, pow(x, k, 0x78000001)] for x in a], dtype=np.uint32)


LOCALQ_CASES = [
    # name, air/trace factory, log_n, (log_blowup, queries, pow), G, expect the local path (cosets/rank >= qd)
    ("mul64-qd2-b4-G8", lambda n: (SynthMulAir64(), None), 10, (4, 16, 8), 8, True),
    ("mul64-qd2-b2-G2", lambda n: (SynthMulAir64(), None), 13, (2, 28, 8), 2, True),
    ("mul64-qd2-b2-G4-fallback", lambda n: (SynthMulAir64(), None), 10, (2, 28, 8), 4, False),
    ("ext163-qd1-b4-G8", lambda n: ("ext163", None), 10, (4, 16, 8), 8, True),
    ("pow5-qd4-b3-G2", lambda n: (PowAir(5), pow_trace(n, 5)), 9, (3, 12, 8), 2, True),
    ("pow5-qd4-b3-G4-fallback", lambda n: (PowAir(5), pow_trace(n, 5)), 9, (3, 12, 8), 4, False),
    ("pow9-qd8-b4-G2", lambda n: (PowAir(9), pow_trace(n, 9)), 8, (4, 10, 8), 2, True),
    ("fib-qd1-b2-G4", lambda n: ("fib", None), 11, (2, 9, 8), 4, True),
    ("mul64-qd2-b3-G1", lambda n: (SynthMulAir64(), None), 9, (3, 9, 8), 1, True),
    # two-pass LDE (n > 4096) of the per-rank chunk matrices of the local quotient
    ("mul64-qd2-b3-G4-2p14", lambda n: (SynthMulAir64(), None), 14, (3, 9, 8), 4, True),
]


def SynthMulAir64():
    from tapstark_amd.airs import SynthMulAir
    return SynthMulAir(64)


@pytest.mark.parametrize("name,make,log_n,cfg,G,local", LOCALQ_CASES, ids=[c[0] for c in LOCALQ_CASES])
def test_local_quotient_same_proof_no_broadcast(ctx, orc, name, make, log_n, cfg, G, local):
    """ts_shard_options.local_quotient: every rank evaluates the quotient on its own cosets, extends
    the per-coset interpolants to its slab and mixes them with the Vandermonde change-of-basis matrix
    (csrc/sharded.cpp).  The proof must be the single-GPU proof (= the oracle's) word for word, and no
    broadcast may happen -- except where a rank holds fewer cosets than the quotient degree, where the
    option falls back to the broadcast path."""
    from tapstark_amd import _lib
    from tapstark_amd.airs import (FibonacciAir, SynthExtAir, fibonacci_public_values,
                                   generate_fibonacci_trace, generate_synth_ext_trace, generate_synth_mul_trace)
    from tapstark_amd.comm import LocalCommGroup

    n = 1 << log_n
    air, trace = make(n)
    pis = np.zeros(0, dtype=np.uint32)
    if air == "ext163":
        air, trace = SynthExtAir(163), generate_synth_ext_trace(n, 163)
    elif air == "fib":
        air, trace = FibonacciAir(), generate_fibonacci_trace(0, 1, n)
        pis = fibonacci_public_values(trace)
    elif trace is None:
        trace = generate_synth_mul_trace(n)
    tape = ts.air_tape(air, len(pis))
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    want = ts.prove(config, ts.CompiledAir(ctx, tape), ts.BfChallenger(), trace.copy(), pis).words
    oracle = orc.prove(orc.FriConfig(*cfg), tape, trace, pis)
    assert len(oracle) == len(want) and (oracle == want).all()
    group = LocalCommGroup(G)
    n_bcast = [0] * G

    def rank(r):
        c = ts.Context(0)
        conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
        inner = group.comm(r)
        ic = inner.c

        def bc(_u, buf, nbytes, root, stream):  # count the broadcasts, then do them
            n_bcast[r] += 1
            return ic.broadcast(ic.user, buf, nbytes, root, stream)

        class Counting:
            pass
        cc = Counting()
        cc._bc = _lib.BROADCAST_FN(bc)
        cc.c = _lib.CommC(ic.rank, ic.world, ic.user, ic.all_gather, cc._bc, ic.abort)
        cc.rank, cc.world, cc.error, cc._keep = r, G, None, inner
        rows = trace[r * n // G:(r + 1) * n // G]
        p = ts.prove_sharded(conf, ts.CompiledAir(c, tape), ts.BfChallenger(), np.ascontiguousarray(rows), pis, cc,
                             min_local_log=3, local_quotient=True)
        return p.words

    res = _thread_ranks(G, rank)
    for r, words in enumerate(res):
        assert len(words) == len(want) and (words == want).all(), \
            f"rank {r}: {int((words != want).sum())} words differ from ts_prove"
    qd = 1 << ts.CompiledAir(ctx, tape).log_quotient_degree
    assert n_bcast == [0 if local else qd] * G, n_bcast


@pytest.mark.parametrize("G,cfg", [(2, (2, 9, 8)), (8, (4, 9, 8))])
def test_local_quotient_on_an_invalid_trace_is_still_the_single_gpu_proof(ctx, orc, G, cfg):
    """For a trace that violates its constraints, constraints / Z_H is not a polynomial of degree
    < n qd.  The reference (release build, prover.rs:40-41) and ts_prove commit to the interpolants of
    its values on the quotient domain and hand out a proof the verifier rejects (OodEvaluationMismatch);
    the broadcast path does exactly the same, word for word.  The local-quotient ranks interpolate the
    values on their OWN cosets: the mixed chunk LDEs are then not low-degree and FRI's final polynomial
    is not constant (fri/src/prover.rs:129-134) -- which every rank sees alike, and which sends all of
    them back through the broadcast path (csrc/sharded.cpp): the call returns ts_prove's proof for
    EVERY trace, and the context counts the fall-back."""
    from tapstark_amd.airs import SynthMulAir, generate_synth_mul_trace
    from tapstark_amd.comm import LocalCommGroup

    n = 1 << 9
    air = SynthMulAir(64)
    tape = ts.air_tape(air, 0)
    bad = generate_synth_mul_trace(n)
    bad[9, 2] = (int(bad[9, 2]) + 1) % 0x78000001
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    single = ts.prove(config, ts.CompiledAir(ctx, tape), ts.BfChallenger(), bad.copy(), []).words
    assert orc.verify(orc.FriConfig(*cfg), tape, single, []) == 7  # OodEvaluationMismatch
    for localq in (False, True):
        group = LocalCommGroup(G)

        def rank(r):
            c = ts.Context(0)
            conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
            ch = ts.BfChallenger()
            words = ts.prove_sharded(conf, ts.CompiledAir(c, tape), ch, bad.copy(), [], group.comm(r),
                                     min_local_log=3, trace_replicated=True, local_quotient=localq).words
            return words, c.stat(5), ch.state()

        res = _thread_ranks(G, rank)
        ch1 = ts.BfChallenger()
        ts.prove(config, ts.CompiledAir(ctx, tape), ch1, bad.copy(), [])
        for words, fallbacks, state in res:
            assert len(words) == len(single) and (words == single).all()
            assert fallbacks == (1 if localq else 0)
            assert (np.asarray(state) == np.asarray(ch1.state())).all()  # the transcript ends where ts_prove's does


def test_shard_options_of_another_layout_are_refused(ctx):
    """ABI 5: ts_shard_options starts with struct_size.  A caller built against ABI 4 (whose first field
    was min_local_log, and which had a field that did nothing) is refused instead of having its fields
    read as something else."""
    from tapstark_amd._lib import TsError
    from tapstark_amd.airs import SynthMulAir, generate_synth_mul_trace
    from tapstark_amd.comm import LocalCommGroup

    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 4), ctx))
    group = LocalCommGroup(1)
    for bad in (0, 12, 20):
        with pytest.raises(TsError, match="struct_size"):
            ts.prove_sharded(config, SynthMulAir(7), ts.BfChallenger(), generate_synth_mul_trace(16, 7), [],
                             group.comm(0), _options_struct_size=bad)
    ts.prove_sharded(config, SynthMulAir(7), ts.BfChallenger(), generate_synth_mul_trace(16, 7), [], group.comm(0))
