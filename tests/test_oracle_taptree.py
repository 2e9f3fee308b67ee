"""CPU tests of the taptree commitment: the oracle (oracle/taptree.c) against published vectors
(NIST SHA-256, BIP-341) and the reference's own known answers and test properties
(basic/src/tcs/mod.rs:594-602, basic/src/tcs/complete_taptree.rs:163-369), then the product's host
side (csrc/taptree.cpp: script assembly, TapLeaf/TapBranch, verify_batch) against the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from tapstark_amd import taptree as tt
from tapstark_amd.airs import splitmix64_stream

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "kats.json")))
P = 0x78000001


@pytest.fixture(scope="module")
def lib():
    from tapstark_amd.build import build

    build()
    from tapstark_amd import _lib

    return _lib.lib()


# ------------------------------------------------------------------ published vectors
def test_sha256_nist_vectors(orc):
    for v in KATS["sha256_nist"]["vectors"]:
        msg = v["msg_ascii"].encode() if "msg_ascii" in v else v["msg_repeat"][0].encode() * v["msg_repeat"][1]
        assert orc.sha256(msg).hex() == v["digest"] == hashlib.sha256(msg).hexdigest()
    for ln in (55, 56, 63, 64, 65, 119, 120, 1000):  # padding boundaries, against hashlib
        msg = bytes((7 * i + 1) & 0xFF for i in range(ln))
        assert orc.sha256(msg) == hashlib.sha256(msg).digest()


def test_bip341_tapleaf_and_tapbranch_vectors(orc, lib):
    b = KATS["bip341"]
    for leaf in b["leaves"]:
        script = bytes.fromhex(leaf["script"])
        assert orc.tapleaf_hash(script, leaf["version"]).hex() == leaf["leaf_hash"]
        assert tt.tapleaf_hash(script).hex() == leaf["leaf_hash"]  # the product's host TapLeaf (0xc0)
    t = b["two_leaf_tree"]
    hs = [orc.tapleaf_hash(bytes.fromhex(s), v) for s, v in zip(t["scripts"], t["versions"])]
    assert orc.tapbranch(hs[0], hs[1]).hex() == t["merkle_root"]
    assert orc.tapbranch(hs[1], hs[0]).hex() == t["merkle_root"]  # children are sorted
    assert tt.tapbranch_hash(hs[0], hs[1]).hex() == t["merkle_root"]
    assert tt.tapbranch_hash(hs[1], hs[0]).hex() == t["merkle_root"]
    # tagged hash by the book
    th = hashlib.sha256(b"TapLeaf").digest()
    s0 = bytes.fromhex(b["leaves"][0]["script"])
    assert hashlib.sha256(th + th + b"\xc0" + bytes([len(s0)]) + s0).hexdigest() == b["leaves"][0]["leaf_hash"]


def test_script_number_pushes(orc):
    # rust-bitcoin Builder::push_int: OP_0, OP_1..OP_16, else a minimal little-endian sign-magnitude push
    want = {0: "00", 1: "51", 16: "60", 17: "0111", 127: "017f", 128: "028000", 255: "02ff00", 256: "020001",
            32767: "02ff7f", 32768: "03008000", 8388607: "03ffff7f", 8388608: "0400008000",
            0x78000000: "0400000078", 0x7FFFFFFF: "04ffffff7f"}
    for v, hexs in want.items():
        assert orc.script_push_int(v).hex() == hexs, v


def test_padding_matrix_known_answer(orc):
    k = KATS["padding_matrix"]
    mats = [np.array(k[n], dtype=np.uint32) for n in ("mat_1", "mat_2", "mat_3")]  # the test's input order
    assert orc.padding_matrix(mats).tolist() == k["leaf_ys"]
    # already sorted input gives the same leaves
    assert orc.padding_matrix([mats[2], mats[0], mats[1]]).tolist() == k["leaf_ys"]


# ------------------------------------------------------------------ tree shape (complete_taptree.rs tests)
def _num_script(i):  # script! { {i} OP_ADD }
    from oracle import oracle_py

    return oracle_py.script_push_int(i) + b"\x93"


def _dfs_order(hashes):
    """Leaves in the depth-first order of the NodeInfo the reference builds: (root hash, [merkle idx])."""
    nodes = [(h, [i]) for i, h in enumerate(hashes)]
    from oracle import oracle_py

    while len(nodes) > 1:
        nxt = []
        for a, b in zip(nodes[0::2], nodes[1::2]):
            h = oracle_py.tapbranch(a[0], b[0])
            nxt.append((h, a[1] + b[1] if a[0] <= b[0] else b[1] + a[1]))
        nodes = nxt
    return nodes[0]


@pytest.mark.parametrize("n", [1, 2, 16, 64])
def test_build_tree_properties(orc, n):
    # complete_taptree.rs:163-209 test_build_tree: every query index finds its own script and its
    # merkle path verifies against the root
    scripts = [_num_script(i) for i in range(n)]
    t = orc.OracleTaptree.from_scripts(scripts)
    root, order = _dfs_order([orc.tapleaf_hash(s) for s in scripts])
    assert t.root == root
    li = list(t.leaf_indices)
    assert sorted(li) == list(range(n))
    for m in range(n):
        assert order[li[m]] == m  # leaves().nth(leaf_indices[m]) is merkle leaf m
        assert orc.taptree_verify_inclusion(t.root, orc.tapleaf_hash(scripts[m]), t.path(m))
        assert tt.verify_inclusion(t.root, tt.tapleaf_hash(scripts[m]), t.path(m))  # the product's host check
    if n > 1:
        assert not orc.taptree_verify_inclusion(t.root, orc.tapleaf_hash(scripts[0]), t.path(1))
        assert not tt.verify_inclusion(t.root, tt.tapleaf_hash(scripts[0]), t.path(1))


# ------------------------------------------------------------------ leaf scripts and the MMCS, host side
def _locks(q, n_evals, u32=1):
    return [tt.winternitz_lock_script(bytes([q, s, 7]), 1 if s == 0 else u32) for s in range(1 + n_evals)]


def test_leaf_script_product_equals_oracle(orc, lib):
    vals = splitmix64_stream(3, 12)
    vals[0], vals[1], vals[2], vals[3] = 0, 16, 17, 200  # every push length
    for u32, n_evals in ((1, 12), (4, 3)):
        locks = _locks(5, n_evals, u32)
        for index in (0, 1, 16, 17, 300, 70000, (1 << 26) + 5):
            want = orc.tap_leaf_script(locks, index, vals, u32)
            assert tt.leaf_script(locks, index, vals, u32) == want
            # skeleton of tcs/mod.rs:197-225: index lock, push(index), OP_EQUALVERIFY, ..., OP_1
            assert want.startswith(locks[0] + orc.script_push_int(index) + b"\x88") and want[-1] == 0x51
    # U32_SIZE 4: the limbs of an evaluation are pushed last limb first (:214-217)
    locks = _locks(1, 1, 4)
    s = orc.tap_leaf_script(locks, 2, [10, 11, 12, 13], 4)
    assert s.endswith(locks[1] + b"\x5d\x88\x5c\x88\x5b\x88\x5a\x88\x51")


def test_mmcs_verify_batch_host_against_oracle_tree(orc, lib):
    # TCS::verify (tcs/mod.rs:425-436) minus script execution: prover = oracle, verifier = product
    k = KATS["padding_matrix"]
    mats = [np.array(k[n], dtype=np.uint32) for n in ("mat_3", "mat_1", "mat_2")]
    ys = orc.padding_matrix(mats)
    locks = _locks(0, 7)
    tree = orc.tap_commit_polys(mats, locks)
    mm = tt.TapTreeMmcs(1, None, host_only=True)
    for index in range(8):
        assert mm.verify_batch(locks, index, ys[index], tree.path(index), tree.root)
        bad = ys[index].copy()
        bad[3] = (int(bad[3]) + 1) % P
        assert not mm.verify_batch(locks, index, bad, tree.path(index), tree.root)
        assert not mm.verify_batch(locks, index ^ 1, ys[index], tree.path(index), tree.root)
    other = _locks(1, 7)
    assert not mm.verify_batch(other, 0, ys[0], tree.path(0), tree.root)  # another tree's bit commitments


# ------------------------------------------------------------------ the lock-script stand-in
def _ripemd160(msg: bytes) -> bytes:
    """RIPEMD-160 by the book (independent of csrc/taptree.cpp), for the stand-in's public keys."""
    rl = [list(range(16)), [7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5, 2, 14, 11, 8],
          [3, 10, 14, 4, 9, 15, 8, 1, 2, 7, 0, 6, 13, 11, 5, 12], [1, 9, 11, 10, 0, 8, 12, 4, 13, 3, 7, 15, 14, 5, 6, 2],
          [4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13]]
    rr = [[5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12], [6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12, 4, 9, 1, 2],
          [15, 5, 1, 3, 7, 14, 6, 9, 11, 8, 12, 2, 10, 0, 4, 13], [8, 6, 4, 1, 3, 11, 15, 0, 5, 12, 2, 13, 9, 7, 10, 14],
          [12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11]]
    sl = [[11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8], [7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9, 11, 7, 13, 12],
          [11, 13, 6, 7, 14, 9, 13, 15, 14, 8, 13, 6, 5, 12, 7, 5], [11, 12, 14, 15, 14, 15, 9, 8, 9, 14, 5, 6, 8, 6, 5, 12],
          [9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8, 5, 6]]
    sr = [[8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6], [9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7, 6, 15, 13, 11],
          [9, 7, 15, 11, 8, 6, 6, 14, 12, 13, 5, 14, 13, 13, 7, 5], [15, 5, 8, 11, 14, 14, 6, 14, 6, 9, 12, 9, 12, 5, 15, 8],
          [8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11, 11]]
    kl = [0, 0x5A827999, 0x6ED9EBA1, 0x8F1BBCDC, 0xA953FD4E]
    kr = [0x50A28BE6, 0x5C4DD124, 0x6D703EF3, 0x7A6D76E9, 0]
    M = 0xFFFFFFFF

    def rol(x, n):
        return ((x << n) | (x >> (32 - n))) & M

    def f(j, x, y, z):
        return [x ^ y ^ z, (x & y) | (~x & M & z), ((x | (~y & M)) ^ z), (x & z) | (y & ~z & M),
                x ^ (y | (~z & M))][j]

    h = [0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0]
    m = msg + b"\x80" + b"\x00" * ((55 - len(msg)) % 64) + (8 * len(msg)).to_bytes(8, "little")
    for off in range(0, len(m), 64):
        X = [int.from_bytes(m[off + 4 * i:off + 4 * i + 4], "little") for i in range(16)]
        al, bl, cl, dl, el = h
        ar, br, cr, dr, er = h
        for rnd in range(5):
            for i in range(16):
                t = (rol((al + f(rnd, bl, cl, dl) + X[rl[rnd][i]] + kl[rnd]) & M, sl[rnd][i]) + el) & M
                al, el, dl, cl, bl = el, dl, rol(cl, 10), bl, t
                t = (rol((ar + f(4 - rnd, br, cr, dr) + X[rr[rnd][i]] + kr[rnd]) & M, sr[rnd][i]) + er) & M
                ar, er, dr, cr, br = er, dr, rol(cr, 10), br, t
        t = (h[1] + cl + dr) & M
        h = [t, (h[2] + dl + er) & M, (h[3] + el + ar) & M, (h[4] + al + br) & M, (h[0] + bl + cr) & M][:]
        h = [h[0], h[1], h[2], h[3], h[4]]
    return b"".join(x.to_bytes(4, "little") for x in h)


def test_ripemd160_vectors():
    for v in KATS["ripemd160"]["vectors"]:
        assert _ripemd160(v["msg_ascii"].encode()).hex() == v["digest"]


def test_winternitz_lock_script_stand_in(lib):
    # the stand-in follows the LOCAL copy of the construction: scripts/src/bit_comm/winternitz.rs
    # :171-274 checksig_verify, :282-297 generate_public_key, bit_comm_u32.rs:80-85, u32_std.rs:122-173
    def hash160(b):
        return _ripemd160(hashlib.sha256(b).digest())

    def pubkey(secret, digit):
        h = hash160(secret + bytes([digit]))
        for _ in range(15):
            h = hash160(h)
        return h

    secret = bytes.fromhex("b138982ce17ac813d505b5b40b665d404e9528e7")  # winternitz.rs:314 MY_SECKEY
    s = tt.winternitz_lock_script(secret, 1)
    per_digit = 2 + 3 + 30 + 2 + 21 + 1 + 8  # OP_15 OP_MIN, dup/toalt x2, 15x(dup hash160), fromalt pick, push20, equalverify, 8x 2drop
    assert per_digit == 67
    for d in range(10):
        blk = s[per_digit * d:per_digit * (d + 1)]
        assert blk[:5] == bytes([0x5F, 0xA3, 0x76, 0x6B, 0x6B]) and blk[5:35] == bytes([0x76, 0xA9]) * 15
        assert blk[35:38] == bytes([0x6C, 0x79, 0x14]) and blk[38:58] == pubkey(secret, 9 - d)
        assert blk[58:] == bytes([0x88]) + bytes([0x6D]) * 8
    tail = s[670:]
    assert tail.startswith(bytes([0x6C, 0x76, 0x8F]) + bytes([0x6C, 0x7D, 0x94]) * 7 + bytes([0x01, 0x78, 0x93]))
    assert s.endswith(bytes([0x6C, 0x63, 0x8F, 0x68]))  # u32_compress: OP_FROMALTSTACK OP_IF OP_NEGATE OP_ENDIF
    s4 = tt.winternitz_lock_script(secret, 4)
    assert len(s4) == 4 * (len(s) + 1) + 4 and s4.endswith(bytes([0x6C]) * 4)


# ------------------------------------------------------------------ prove/verify over taptrees (host side)
def _lock_for(ci, q, s, u32):
    return tt.winternitz_lock_script(bytes([ci, q, s & 0xFF, s >> 8]), u32)


def _tap_case(name, log_n, cfg):
    import tapstark_amd as ts
    from tapstark_amd.airs import (FibonacciAir, SynthMulAir, fibonacci_public_values,
                                   generate_fibonacci_trace, generate_synth_mul_trace)

    n = 1 << log_n
    if name == "fib":
        air, trace = FibonacciAir(), generate_fibonacci_trace(0, 1, n)
        pis = fibonacci_public_values(trace)
    else:
        air, trace, pis = SynthMulAir(7), generate_synth_mul_trace(n, 7), np.zeros(0, dtype=np.uint32)
    tape = ts.air_tape(air, len(pis))
    qd = 1 << ts.get_log_quotient_degree(air, len(pis))
    locks = tt.make_lock_table(cfg[1], trace.shape[1], qd, log_n, _lock_for)
    return air, trace, pis, tape, locks


@pytest.mark.parametrize("name,log_n,cfg", [("fib", 3, (2, 4, 8)), ("fib", 5, (1, 3, 4)), ("mul7", 4, (2, 3, 8))],
                         ids=["fib8-q4", "fib32-b1", "mul7-16"])
def test_native_verify_tap_on_oracle_proofs(orc, lib, name, log_n, cfg):
    # the reference's own configuration (uni-stark/tests/fib_air.rs:117-149: TapTreeMmcs as both
    # MMCSs): prover = oracle in taptree mode, verifier = the product's host code; verdicts on
    # tampered proofs must agree with the oracle's verifier
    import tapstark_amd as ts

    air, trace, pis, tape, locks = _tap_case(name, log_n, cfg)
    ocfg = orc.FriConfig(*cfg)
    proof = orc.prove_tap(ocfg, tape, trace, pis, locks)
    assert proof[1] == 2 and proof[5] == cfg[1]  # TSPF v2, num_queries roots per commitment
    assert orc.verify_tap(ocfg, tape, proof, pis, locks) == 0
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True))
    assert tt.verify_tap(config, air, ts.BfChallenger(), proof, pis, locks) == 0
    # a v2 proof is not a v1 proof and vice versa
    with pytest.raises(ts.VerificationError):
        ts.verify(config, air, ts.BfChallenger(), proof, pis)
    rng = np.random.default_rng(3)
    rejected = 0
    for pos in [6, 6 + 8 * cfg[1] + 3, len(proof) - 1, len(proof) - 3] + [int(x) for x in rng.integers(6, len(proof), 12)]:
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P if bad[pos] < P else int(bad[pos]) ^ 1
        want = orc.verify_tap(ocfg, tape, bad, pis, locks)
        got = tt.verify_tap(config, air, ts.BfChallenger(), bad, pis, locks)
        assert (got == 0) == (want == 0) and (got == want or {got, want} <= {4, 5, 8, 9, 1}), (pos, got, want)
        rejected += want != 0
    assert rejected >= 12
    # another prover's bit commitments: the leaves do not rebuild
    other = list(locks)
    other[3] = other[3][:-2] + b"\x51\x51"
    assert tt.verify_tap(config, air, ts.BfChallenger(), proof, pis, other) != 0
    assert orc.verify_tap(ocfg, tape, proof, pis, other) != 0


def test_postcard_round_trip_of_a_proof_over_taptrees(orc, lib):
    # TSPF v2 <-> postcard: Commitment = Vec<[[u8; 4]; 8]> with num_queries roots
    # (basic/src/mmcs/taptree_mmcs.rs:43); everything else as for v1
    import tapstark_amd as ts

    cfg = (2, 4, 8)
    air, trace, pis, tape, locks = _tap_case("fib", 3, cfg)
    proof = orc.prove_tap(orc.FriConfig(*cfg), tape, trace, pis, locks)
    data = ts.Proof(words=proof).to_postcard()
    assert data[0] == 4 and data[1:33] == proof[6:14].astype("<u4").tobytes()  # varint(Q) then the first root
    back = ts.Proof.from_postcard(data)
    assert (back.words == proof).all()
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True))
    assert tt.verify_tap(config, air, ts.BfChallenger(), back.words, pis, locks) == 0
