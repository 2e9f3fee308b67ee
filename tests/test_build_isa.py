"""Build-time check of the cross-workgroup hand-off in the whole-tree kernels (merkle_tree.hpp
tree_body): the sub-roots must be stored and loaded with `global_* ... sc1` (written
through / bypassing the CU's L1), never through `flat_` instructions, and the finishing workgroup
must run an agent-scope acquire (`buffer_inv sc1`) before it reads them.  Needs hipcc only (no GPU):
the gfx950 code object of the built library is disassembled."""
import re

import pytest

from tapstark_amd.build import device_disassembly


@pytest.fixture(scope="module")
def merkle_isa():
    return device_disassembly("merkle.hip")


@pytest.fixture(scope="module")
def fri_isa():
    return device_disassembly("fri.hip")


def _kernel(isa, needle):
    names = [n for n in isa if needle in n]
    assert names, f"{needle} not found among {sorted(isa)[:8]}..."
    return [(n, isa[n]) for n in names]


def _check_handoff(name, lines):
    ops = [l.split()[0] for l in lines if l]
    assert not any(o.startswith("flat_") for o in ops), f"{name}: flat_ memory access in a hand-off kernel"
    sc1_stores = [l for l in lines if l.startswith("global_store_dword") and l.rstrip().endswith("sc1")]
    sc1_loads = [l for l in lines if l.startswith("global_load_dword") and l.rstrip().endswith("sc1")]
    # the published sub-root (two words per lane of a quad, twice: unrolled) and the ticket reset
    assert len(sc1_stores) >= 3, f"{name}: sub-root stores are not sc1: {sc1_stores}"
    assert len(sc1_loads) >= 1, f"{name}: sub-root loads are not sc1"
    assert any(re.match(r"buffer_inv\s+sc1", l) for l in lines), f"{name}: no agent-scope acquire (buffer_inv sc1)"
    assert any(l.startswith("global_atomic_add") for l in lines), f"{name}: no ticket add"
    # the storing waves drain their stores before the ticket: an s_waitcnt vmcnt(0) precedes the atomic
    i_atomic = next(i for i, l in enumerate(lines) if l.startswith("global_atomic_add"))
    assert any(re.match(r"s_waitcnt\s+vmcnt\(0\)", l) for l in lines[:i_atomic]), f"{name}: stores not drained"
    # acquire comes after the ticket add, the sc1 loads after the acquire
    i_inv = next(i for i, l in enumerate(lines) if re.match(r"buffer_inv\s+sc1", l))
    i_load = next(i for i, l in enumerate(lines) if l.startswith("global_load_dword") and l.rstrip().endswith("sc1"))
    assert i_atomic < i_inv < i_load, f"{name}: order ticket -> acquire -> loads violated ({i_atomic}, {i_inv}, {i_load})"


def test_merkle_tree_handoff_isa(merkle_isa):
    for name, lines in _kernel(merkle_isa, "k_merkle_tree"):
        _check_handoff(name, lines)


def test_fri_round_handoff_isa(fri_isa):
    for name, lines in _kernel(fri_isa, "k_fri_round"):
        _check_handoff(name, lines)


def test_leaf_tree_handoff_isa(merkle_isa, fri_isa):
    # leaves + tree in one launch (leaf_tree.hpp): the same finisher, so the same hand-off rules, for
    # every leaf kind and every leaves-per-lane variant
    n = 0
    for isa in (merkle_isa, fri_isa):
        for name, lines in _kernel(isa, "k_leaf_tree"):
            _check_handoff(name, lines)
            n += 1
    assert n == 20  # (strided, table, ef_pairs, fri_fold, fri_leaf) x (1, 2, 4, 8 leaves per lane)
