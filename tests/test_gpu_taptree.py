"""GPU tests of the taptree-compatible commitment: the SHA-256
kernels (TapLeaf over host-supplied scripts and over leaf scripts assembled on the device, TapBranch
levels) against the oracle, on the reference's own test shapes:
  basic/src/tcs/complete_taptree.rs:163-369  test_build_tree, test_combine_tree, test_combine_with_different_depth
  basic/src/tcs/mod.rs:520-718               test_taptree_mmcs, test_taptree_mmcs_with_multi_query
  basic/src/mmcs/taptree_mmcs.rs:133-231     the same three matrices through the BFMmcs surface
"""
import json
import os

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd import taptree as tt
from tapstark_amd._lib import TsError
from tapstark_amd.airs import splitmix64_stream

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "kats.json")))
P = 0x78000001


@pytest.fixture(scope="module")
def ctx():
    from tapstark_amd.build import build

    build()
    return ts.default_context()


def num_script(orc, i):  # script! { {i} OP_ADD }
    return orc.script_push_int(i) + b"\x93"


def locks_for(q, n_evals, u32=1):
    return [tt.winternitz_lock_script(bytes([q, s & 0xFF, s >> 8, 9]), 1 if s == 0 else u32)
            for s in range(1 + n_evals)]


def test_build_tree(ctx, orc):
    # complete_taptree.rs:163-209
    scripts = [num_script(orc, i) for i in range(16)]
    tree = tt.CompleteTaptree.new_with_scripts(scripts, ctx)
    want = orc.OracleTaptree.from_scripts(scripts)
    assert tree.leaf_count == 16 and tree.root == want.root
    for q in range(16):
        leaf, path = tree.get_leaf_proof(q)
        assert leaf == orc.tapleaf_hash(scripts[q]) == tt.tapleaf_hash(scripts[q])
        assert path == want.path(q)
        assert tt.verify_inclusion(tree.root, leaf, path)  # verify_inclusion_by_index
        assert orc.taptree_verify_inclusion(tree.root, leaf, path)
    one = tt.CompleteTaptree.new_with_scripts([scripts[3]], ctx)  # a single leaf is its own root
    assert one.root == orc.tapleaf_hash(scripts[3]) and one.get_leaf_proof(0) == (one.root, [])
    with pytest.raises(TsError):  # builder.rs:40 assert!(is_power_of_two(leaf_count))
        tt.CompleteTaptree.new_with_scripts(scripts[:15], ctx)


@pytest.mark.parametrize("nb", [8, 4], ids=["test_combine_tree", "test_combine_with_different_depth"])
def test_combine(ctx, orc, nb):
    # complete_taptree.rs:211-369: merkle indices follow [self leaves..., other leaves...]
    a_s = [num_script(orc, i) for i in range(8)]
    b_s = [num_script(orc, 8 + i) for i in range(nb)]
    a, b = tt.CompleteTaptree.new_with_scripts(a_s, ctx), tt.CompleteTaptree.new_with_scripts(b_s, ctx)
    oa, ob = orc.OracleTaptree.from_scripts(a_s), orc.OracleTaptree.from_scripts(b_s)
    for first, second, fs, ss, of, os_ in ((a, b, a_s, b_s, oa, ob), (b, a, b_s, a_s, ob, oa)):
        c = first.combine(second)
        assert c.leaf_count == len(fs) + len(ss)
        assert c.root == orc.tapbranch(of.root, os_.root)
        for q, script in enumerate(fs + ss):
            leaf, path = c.get_leaf_proof(q)
            assert leaf == orc.tapleaf_hash(script)
            inner = of.path(q) + [os_.root] if q < len(fs) else os_.path(q - len(fs)) + [of.root]
            assert path == inner
            assert tt.verify_inclusion(c.root, leaf, path)


def test_tapleaf_kernel_on_scripts_of_every_length(ctx, orc):
    # host-supplied scripts of 0..1100 bytes: every alignment of the tail, the 0xfd compact-size
    # switch at 253 bytes, padding boundaries of SHA-256 (55/56/64 mod 64)
    rng = np.random.default_rng(7)
    lens = list(range(0, 140)) + [247, 250, 251, 252, 253, 254, 255, 256, 300, 511, 512, 1000, 1100]
    lens += [int(x) for x in rng.integers(0, 1100, 256 - len(lens))]
    scripts = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in lens]
    tree = tt.CompleteTaptree.new_with_scripts(scripts, ctx)
    want = orc.OracleTaptree.from_scripts(scripts)
    assert tree.root == want.root
    for q in range(len(scripts)):
        leaf, path = tree.get_leaf_proof(q)
        assert leaf == want.levels[0][q], f"leaf {q} (len {lens[q]})"
    assert tree.get_leaf_proof(200)[1] == want.path(200)


def three_matrices():
    k = KATS["padding_matrix"]
    return [np.array(k[n], dtype=np.uint32) for n in ("mat_3", "mat_1", "mat_2")], k["leaf_ys"]


@pytest.mark.parametrize("query_times", [1, 8], ids=["test_taptree_mmcs", "test_taptree_mmcs_with_multi_query"])
def test_taptree_mmcs_reference_shape(ctx, orc, query_times):
    # tcs/mod.rs:520-718 and taptree_mmcs.rs:133-231: mat_1 4x2, mat_2 4x4, mat_3 8x1
    mats, leaf_ys = three_matrices()
    mm = tt.TapTreeMmcs(query_times, locks_for, ctx=ctx)
    roots, data = mm.commit([m.copy() for m in mats])
    assert len(roots) == query_times == data.num_queries and data.n_evals == 7 and data.log_max_height == 3
    assert len(set(roots)) == query_times  # a tree per query, each with its own bit commitments
    for q in range(query_times):
        want = orc.tap_commit_polys(mats, locks_for(q, 7))
        assert roots[q] == want.root, f"tree {q}"
        for index in range(8):
            rows, path, script = mm.open_batch(q, index, data, 7)
            assert rows.tolist() == leaf_ys[index]  # the layout written at tcs/mod.rs:594-602
            assert path == want.path(index)
            assert script == orc.tap_leaf_script(locks_for(q, 7), index, rows)
            assert orc.tapleaf_hash(script) == want.levels[0][index]
            assert mm.verify_batch(data.tree_locks(q), index, rows, path, roots[q])
            bad = rows.copy()
            bad[2] = (int(bad[2]) + 1) % P
            assert not mm.verify_batch(data.tree_locks(q), index, bad, path, roots[q])
            assert not mm.verify_batch(data.tree_locks(q), index, rows, path, roots[(q + 1) % query_times]) \
                or query_times == 1


def test_taptree_mmcs_extension_field_rows(ctx, orc):
    # F::U32_SIZE = 4 (tcs/mod.rs:239-246 CommitType::U128): the FRI commit-phase matrices have rows of
    # two EF4 elements (fri/src/prover.rs:112); limbs are pushed last first (tcs/mod.rs:214-217)
    m = splitmix64_stream(77, 64 * 8).reshape(64, 8)
    mm = tt.TapTreeMmcs(3, lambda q, n: locks_for(q, n, 4), u32_size=4, ctx=ctx)
    roots, data = mm.commit([m.copy()])
    assert data.n_evals == 2
    for q in range(3):
        want = orc.tap_commit_polys([m], locks_for(q, 2, 4), 4)
        assert roots[q] == want.root
        for index in (0, 17, 63):
            rows, path, script = mm.open_batch(q, index, data, 8)
            assert (rows == m[index]).all() and path == want.path(index)
            assert script == orc.tap_leaf_script(locks_for(q, 2, 4), index, rows, 4)
            assert mm.verify_batch(data.tree_locks(q), index, rows, path, roots[q])


@pytest.mark.parametrize("small", [False, True], ids=["field-elements", "small-values"])
def test_taptree_mmcs_larger_mixed_heights(ctx, orc, small):
    # random field elements (4-byte pushes almost everywhere) and small values (pushes of 1-2 bytes:
    # neighbouring leaves drift apart in the byte stream, the divergent case for the leaf kernel)
    shapes = [(10, 5), (10, 1), (8, 3), (5, 2)]
    mats = [splitmix64_stream(300 + i, (1 << lh) * w).reshape(1 << lh, w) for i, (lh, w) in enumerate(shapes)]
    if small:
        mats = [m % np.uint32(300) for m in mats]
    mm = tt.TapTreeMmcs(2, locks_for, ctx=ctx)
    roots, data = mm.commit([m.copy() for m in mats])
    ys = orc.padding_matrix(mats)
    for q in range(2):
        want = orc.tap_commit_polys(mats, locks_for(q, 11))
        assert roots[q] == want.root
        for index in (0, 1, 511, 1023, 700):
            rows, path, script = mm.open_batch(q, index, data, 11)
            assert (rows == ys[index]).all() and path == want.path(index)
            assert mm.verify_batch(data.tree_locks(q), index, rows, path, roots[q])


def test_taptree_mmcs_refuses_what_the_reference_panics_on(ctx):
    m_small, m_tall = np.zeros((4, 2), dtype=np.uint32), np.zeros((8, 1), dtype=np.uint32)
    mm = tt.TapTreeMmcs(1, locks_for, ctx=ctx)
    with pytest.raises(TsError) as e:  # taptree_mmcs.rs:68-72 assert_eq!(openings_flatten, openings_evals)
        mm.commit([m_small, m_tall])
    assert e.value.code == 5
    with pytest.raises(TsError):  # width not a multiple of U32_SIZE
        tt.TapTreeMmcs(1, lambda q, n: locks_for(q, n, 4), u32_size=4, ctx=ctx).commit([np.zeros((4, 6), dtype=np.uint32)])


# ------------------------------------------------------------------ prove() over taptrees
def _lock_for(ci, q, s, u32):
    return tt.winternitz_lock_script(bytes([ci, q, s & 0xFF, s >> 8]), u32)


TAP_PROOFS = [
    # uni-stark/tests/fib_air.rs:117-149 (test_public_value): Fibonacci 2^3, log_blowup 2, 28 queries, 8 PoW bits
    ("fib8_q28", "fib", 3, (2, 28, 8)),
    ("fib8_q6", "fib", 3, (2, 6, 8)),     # fib_air.rs:151-192 uses 6 queries
    ("fib2p7_b1", "fib", 7, (1, 5, 8)),
    ("mul7_2p5_b3", "mul7", 5, (3, 4, 4)),   # quotient degree 2: two chunk matrices in one commitment
    ("mul64_2p6", "mul64", 6, (2, 3, 8)),    # 65 lock scripts per trace leaf
]


@pytest.mark.parametrize("name,air_name,log_n,cfg", TAP_PROOFS, ids=[c[0] for c in TAP_PROOFS])
def test_prove_over_taptrees_bit_identical_to_oracle(ctx, orc, name, air_name, log_n, cfg):
    from tapstark_amd.airs import (FibonacciAir, SynthMulAir, fibonacci_public_values,
                                   generate_fibonacci_trace, generate_synth_mul_trace)

    n = 1 << log_n
    if air_name == "fib":
        air, trace = FibonacciAir(), generate_fibonacci_trace(0, 1, n)
        pis = fibonacci_public_values(trace)
    else:
        w = int(air_name[3:])
        air, trace, pis = SynthMulAir(w), generate_synth_mul_trace(n, w), np.zeros(0, dtype=np.uint32)
    tape = ts.air_tape(air, len(pis))
    qd = 1 << ts.get_log_quotient_degree(air, len(pis))
    locks = tt.make_lock_table(cfg[1], trace.shape[1], qd, log_n, _lock_for)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    ch = ts.BfChallenger()
    proof = tt.prove_tap(config, air, ch, trace, pis, locks)
    ocfg = orc.FriConfig(*cfg)
    och = orc.OracleChallenger()
    want = orc.prove_tap(ocfg, tape, trace, pis, locks, och)
    assert orc.verify_tap(ocfg, tape, proof, pis, locks) == 0, "oracle verifier rejects the GPU proof"
    assert tt.verify_tap(config, air, ts.BfChallenger(), proof, pis, locks) == 0  # prove, then verify afresh
    assert len(proof) == len(want)
    assert (proof == want).all(), f"{int((proof != want).sum())} proof words differ"
    assert ch.sample_bits(20) == och.sample_bits(20)  # the caller's challenger ends in the same state
    bad = proof.copy()
    bad[len(bad) // 2] ^= 1
    assert tt.verify_tap(config, air, ts.BfChallenger(), bad, pis, locks) != 0


def test_prove_over_taptrees_refuses_a_short_lock_table(ctx):
    from tapstark_amd.airs import FibonacciAir, fibonacci_public_values, generate_fibonacci_trace

    trace = generate_fibonacci_trace(0, 1, 8)
    locks = tt.make_lock_table(4, 2, 1, 3, _lock_for)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), ctx))
    with pytest.raises(TsError):
        tt.prove_tap(config, FibonacciAir(), ts.BfChallenger(), trace, fibonacci_public_values(trace), locks[:-1])


# ------------------------------------------------------------------ one proof over G GPUs, split by tree
@pytest.mark.parametrize("G,air_name,log_n,cfg", [
    (2, "fib", 5, (2, 7, 8)),    # 4 + 3 trees
    (4, "fib", 3, (2, 5, 8)),    # 2, 2, 1, 0 trees: a rank with nothing to build
    (8, "fib", 6, (2, 28, 8)),   # the reference's 28 queries over 8 ranks: 4, 4, ..., 0
    (3, "mul7", 5, (3, 4, 4)),   # a world that is not a power of two; two chunk matrices
    (8, "mul64", 4, (2, 3, 4)),  # more ranks than trees
], ids=["G2", "G4-idle-rank", "G8-q28", "G3-mul7", "G8-three-trees"])
def test_prove_over_taptrees_split_by_tree(ctx, G, air_name, log_n, cfg):
    # every rank (a host thread with its own context, in-process communicator) passes the whole trace
    # and must return the proof ts_prove_tap makes on one GPU, with its challenger in the same state
    import threading

    from tapstark_amd.airs import (FibonacciAir, SynthMulAir, fibonacci_public_values,
                                   generate_fibonacci_trace, generate_synth_mul_trace)
    from tapstark_amd.comm import LocalCommGroup

    n = 1 << log_n
    if air_name == "fib":
        air, trace = FibonacciAir(), generate_fibonacci_trace(0, 1, n)
        pis = fibonacci_public_values(trace)
    else:
        w = int(air_name[3:])
        air, trace, pis = SynthMulAir(w), generate_synth_mul_trace(n, w), np.zeros(0, dtype=np.uint32)
    qd = 1 << ts.get_log_quotient_degree(air, len(pis))
    locks = tt.make_lock_table(cfg[1], trace.shape[1], qd, log_n, _lock_for)
    ch0 = ts.BfChallenger()
    want = tt.prove_tap(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx)), air, ch0, trace.copy(), pis, locks)
    state = ch0.sample_bits(20)
    group = LocalCommGroup(G)
    got, errs = [None] * G, [None] * G

    def rank_main(r):
        try:
            c = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
            ch = ts.BfChallenger()
            p = tt.prove_tap(config, air, ch, trace.copy(), pis, locks, comm=group.comm(r))
            got[r] = (p, ch.sample_bits(20))
        except BaseException as e:  # noqa: BLE001
            errs[r] = repr(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(errs), errs
    for r in range(G):
        assert got[r] is not None, f"rank {r} did not finish"
        assert len(got[r][0]) == len(want) and (got[r][0] == want).all(), f"rank {r}: proof differs"
        assert got[r][1] == state
    assert tt.verify_tap(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx)), air, ts.BfChallenger(),
                         got[G - 1][0], pis, locks) == 0


def test_split_by_tree_rank_failure_does_not_hang(ctx):
    # one rank is handed a lock table that is too short: it fails before its first collective and
    # aborts the communicator; its peers return a communicator error instead of waiting for ever
    import threading

    from tapstark_amd.airs import FibonacciAir, fibonacci_public_values, generate_fibonacci_trace
    from tapstark_amd.comm import LocalCommGroup

    G, cfg = 4, (2, 6, 8)
    trace = generate_fibonacci_trace(0, 1, 16)
    pis = fibonacci_public_values(trace)
    locks = tt.make_lock_table(cfg[1], 2, 1, 4, _lock_for)
    group = LocalCommGroup(G)
    out = [None] * G

    def rank_main(r):
        try:
            c = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
            tt.prove_tap(config, FibonacciAir(), ts.BfChallenger(), trace.copy(), pis,
                         locks[:-1] if r == 1 else locks, comm=group.comm(r))
            out[r] = "ok"
        except BaseException as e:  # noqa: BLE001
            out[r] = repr(e)

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(G)]
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    assert all(not t.is_alive() for t in th), f"ranks still waiting: {out}"
    assert "TS_ERR_INVALID" in out[1]
    assert all(o != "ok" for o in out), out
