"""GPU tests of the MMCS layer on its own (BFMmcs, basic/src/mmcs/bf_mmcs.rs:17-68): the device
Blake3 leaf hash against the official BLAKE3 vectors, rows wider than one chunk, batches of more
than 16 matrices, mixed heights -- every digest level against the oracle."""
import json
import os

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd.airs import splitmix64_stream

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from tapstark_amd.build import build

    build()
    return ts.default_context()


def rand_mat(seed, h, w):
    return splitmix64_stream(seed, h * w).reshape(h, w)


def test_device_leaf_hash_official_blake3_vectors(ctx):
    # a one-matrix commit whose row IS the official test input (bytes i % 251, as little-endian
    # words): the leaf digest must be the official digest.  Covers the strided single-chunk kernel
    # (multiples of 64 B), the pointer-table kernel (partial last block) and the multi-chunk path
    # (> 1024 B: chunk counters, parent nodes, ROOT on the last parent only).
    vec = json.load(open(os.path.join(GOLDEN, "blake3_official.json")))
    cases = {int(k): v for g in ("official_lengths", "word_lengths") for k, v in vec[g].items()
             if int(k) % 4 == 0 and 0 < int(k) <= 16384}
    assert {64, 1024, 1028, 2048, 2052, 4096, 8192, 16384} <= set(cases)
    mmcs = ts.Blake3Mmcs(ctx)
    for ln, hexd in sorted(cases.items()):
        row = np.frombuffer(bytes(i % 251 for i in range(ln)), dtype="<u4")
        m = np.tile(row, (4, 1))  # four identical rows: every leaf must agree
        root, data = mmcs.commit([m])
        leaves = data.digests(0)
        for r in range(4):
            assert leaves[r].astype("<u4").tobytes().hex() == hexd, f"len {ln} row {r}"


@pytest.mark.parametrize("shape", [[(6, 300)], [(5, 257), (5, 3)], [(4, 1030), (3, 700), (4, 2)],
                                   [(9, 513)], [(3, 4100)]], ids=str)
def test_commit_rows_wider_than_one_chunk(ctx, orc, shape):
    mats = [rand_mat(700 + i, 1 << lh, w) for i, (lh, w) in enumerate(shape)]
    mmcs = ts.Blake3Mmcs(ctx)
    root, data = mmcs.commit([m.copy() for m in mats])
    om = orc.OracleMmcs(mats)
    for lvl in range(data.log_height + 1):
        assert (data.digests(lvl) == om.layer(lvl)).all(), f"level {lvl}"
    assert (root == om.root).all()
    for idx in {0, (1 << data.log_height) - 1, 5 % (1 << data.log_height)}:
        rows, path = data.open_batch(idx)
        orows, opath = om.open(idx)
        assert (rows == orows).all() and (path == opath).all()
        assert om.verify(idx, rows, path, root)


def test_commit_more_than_16_matrices(ctx, orc):
    # taptree_mmcs.rs:101-114 takes any Vec of matrices; 40 here, three different heights
    mats = [rand_mat(900 + i, 1 << (6 - i % 3), 1 + i % 5) for i in range(40)]
    mmcs = ts.Blake3Mmcs(ctx)
    root, data = mmcs.commit([m.copy() for m in mats])
    om = orc.OracleMmcs(mats)
    assert (root == om.root).all()
    for idx in (0, 63, 37):
        rows, path = data.open_batch(idx)
        orows, opath = om.open(idx)
        assert (rows == orows).all() and (path == opath).all()
    # the same batch through Pcs::commit (with the LDE in front)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(1, 2, 0), ctx)
    root2, data2 = pcs.commit([((6 - i % 3, 1), m.copy()) for i, m in enumerate(mats)])
    ldes = [orc.commit_lde(m, 1, 1) for m in mats]
    assert (root2 == orc.OracleMmcs(ldes).root).all()


def test_commit_longest_trace_2pow26(ctx, orc):
    # the longest trace the field allows at log_blowup 1: n = 2^26, LDE of 2^27 rows = the whole
    # two-adic subgroup (two_adic_pcs.rs:233-241).  The oracle is too slow to diff against at this
    # size: the column is the evaluation of c1 X + c2 X^2 on H_n (built by doubling), so every LDE
    # row is known in closed form; a random sample of rows, the leaf digests under them and the
    # Merkle paths above them are checked.
    P = 0x78000001
    log_n, b = 26, 1
    n, N = 1 << log_n, 1 << (log_n + b)
    w_n = pow(0x1A427A41, 1 << (27 - log_n), P)
    x = np.ones(n, dtype=np.uint64)
    m, step = 1, w_n
    while m < n:  # x[i] = w_n^i
        x[m:2 * m] = x[:m] * np.uint64(step) % np.uint64(P)
        step = step * step % P
        m *= 2
    c1, c2 = 123456789, 987654321
    col = (np.uint64(c1) * x % np.uint64(P) + np.uint64(c2) * (x * x % np.uint64(P)) % np.uint64(P)) % np.uint64(P)
    del x
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 2, 0), ctx)
    root, data = pcs.commit([((log_n, 1), col.astype(np.uint32).reshape(n, 1))])
    del col
    assert data.log_height == log_n + b
    W_N = pow(0x1A427A41, 1 << (27 - (log_n + b)), P)
    rng = np.random.default_rng(5)
    leaves = data.digests(0)
    for r in [0, 1, N - 1, N // 2] + [int(v) for v in rng.integers(0, N, 40)]:
        j = int(format(r, f"0{log_n + b}b")[::-1], 2)
        y = 31 * pow(W_N, j, P) % P
        want = (c1 * y + c2 * y * y) % P
        rows, path = data.open_batch(r)
        assert int(rows[0]) == want, f"row {r}"
        assert leaves[r].astype("<u4").tobytes() == orc.blake3(int(want).to_bytes(4, "little"))
        om_ok = orc.OracleMmcs.verify  # the oracle's verify_batch needs only the shape
        hs = (ts.stark.C.c_size_t * 1)(N)
        ws = (ts.stark.C.c_size_t * 1)(1)
        assert orc.lib().ts_or_mmcs_verify(1, hs, ws, ts.stark.C.c_size_t(r), ts.stark._p(rows),
                                           ts.stark._p(path), ts.stark.C.c_size_t(path.shape[0]),
                                           ts.stark._p(root))


def test_alu_ceiling_probe(ctx):
    # bench.py's alu_ceiling block: whole-chip rates of the library's own butterfly / Blake3 /
    # SHA-256 loops; on an MI355X they are of the order of 10^12, 10^10 and 10^10 per second
    bf, b3, sha = ctx.alu_ceiling(0), ctx.alu_ceiling(1), ctx.alu_ceiling(2)
    assert 1e11 < bf < 2e13 and 5e9 < b3 < 5e11 and 2e9 < sha < 2e11


@pytest.mark.parametrize("knobs", [
    {"TS_TREE_MAX_LOG": "16", "TS_FRI_ROUND_LOG": "0"}, {"TS_TREE_MAX_LOG": "22", "TS_FRI_ROUND_LOG": "22"},
    {"TS_TREE_MAX_LOG": "19", "TS_FRI_ROUND_LOG": "20"},
    # round 4's Merkle path (leaf launch, level launches, tree launch), with two of its shapes
    {"TS_LEAF_TREE": "0"}, {"TS_LEAF_TREE": "0", "TS_TREE_MAX_LOG": "16", "TS_FRI_ROUND_LOG": "0"},
    # the leaf-tree kernel: every leaves-per-lane variant on every height that takes it, the sub-roots to
    # a second launch, every FRI round through it
    {"TS_LEAF_TREE_R": "0"}, {"TS_LEAF_TREE_R": "1"}, {"TS_LEAF_TREE_R": "3"},
    {"TS_LEAF_TREE_FINISH": "0", "TS_FRI_ROUND_LOG": "0"},
    {"TS_HOST_GRIND": "1"},  # the proof-of-work search on the host instead of in the last FRI kernel
    {"TS_LDE_PAIR": "0"},    # a set of LDE launches per quotient chunk instead of one for both
], ids=lambda k: ",".join(f"{a[3:]}={b}" for a, b in k.items()))
def test_tree_launch_shapes_agree(ctx, knobs):
    # how a Merkle commitment is cut into launches (leaves and tree in one launch or apart, leaves per
    # lane, how many levels a whole-tree launch takes, FRI commit rounds with the fold fused in) is a
    # set of tuning knobs read once per process: every setting must give the same roots, paths and
    # proof bytes as the default one in this process
    import subprocess
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import _tree_shapes_probe as probe

    here = probe.probe()
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, probe.__file__], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("PROBE ")][-1]
    assert json.loads(line[6:]) == here


def test_stage_bench_and_context_counters(ctx):
    # ts_bench_stage: sustained loops of one stage on resident data (tools/power_vs_working_set.py);
    # ts_ctx_stat: the counters by index, refusing an unknown one
    from tapstark_amd._lib import TsError

    lde = ctx.bench_stage(0, 14, 8, 2, 3)
    tree = ctx.bench_stage(1, 14, 8, 2, 3)
    assert 0 < lde < 50 and 0 < tree < 50
    for single_pass in (2, 3, 4):  # one LDE pass alone (log_n > 12 only)
        assert 0 < ctx.bench_stage(single_pass, 14, 8, 2, 3) < lde + 1
    with pytest.raises(TsError):
        ctx.bench_stage(5, 14, 8, 2, 3)
    with pytest.raises(TsError):
        ctx.bench_stage(3, 10, 8, 2, 3)
    lde2 = ctx.bench_stage(0, 14, 8, 2, 3)  # the pass mask is back at "all three"
    assert 0.3 * lde < lde2 < 3 * lde
    with pytest.raises(TsError):
        ctx.bench_stage(0, 26, 8, 2, 1)  # LDE larger than the two-adic subgroup
    st = ctx.graph_stats()
    assert [ctx.stat(i) for i in range(4)] == [st["replays"], st["fallbacks"], st["shapes"], st["pool_bytes"]]
    assert ctx.stat(4) == st["reserve_failures"] and ctx.stat(5) >= 0
    assert ctx.stat(7) == 0  # proof-of-work candidates of the device search refused by the host: never
    with pytest.raises(TsError):
        ctx.stat(9)
