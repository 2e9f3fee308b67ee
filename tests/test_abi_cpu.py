"""CPU-side checks of the product library: it loads, exports every symbol include/tapstark.h
declares, and its host-only parts (Fiat-Shamir challenger, AIR tape compiler) agree with the
oracle.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd import _lib
from tapstark_amd.airs import FibonacciAir, SynthMulAir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from tapstark_amd.build import build

    build()
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "tapstark.h")).read()
    declared = set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", hdr)) - {"ts_status"}
    assert declared == set(_lib.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ts_abi_version() == 1


def test_no_cpu_fallback_without_device(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.TsError):
        ts.Context(0)


def test_challenger_matches_oracle(lib, orc):
    rng = np.random.default_rng(1)
    for perm in (0, 1):
        for ext in (True, False):
            a = ts.BfChallenger(perm, ext)
            b = orc.OracleChallenger(perm, ext)
            for step in range(200):
                k = rng.integers(0, 4)
                if k == 0:
                    w = int(rng.integers(0, 2**32))
                    a.observe(w)
                    b.observe(w)
                elif k == 1:
                    d = rng.integers(0, 2**32, size=8, dtype=np.uint64).astype(np.uint32)
                    a.observe_commitment(d)
                    b.observe_digest(d)
                elif k == 2:
                    assert a.sample().tolist() == b.sample().tolist()
                else:
                    bits = int(rng.integers(1, 28))
                    assert a.sample_bits(bits) == b.sample_bits(bits)
            if perm == 0:
                assert a.grind(8) == b.grind(8)
            st = a.state()
            assert st[:16].tolist() == list(b.c.state)
            assert int(st[16]) == b.c.n_in and int(st[25]) == b.c.n_out


def test_challenger_reference_kat(lib):
    # reference script_expr/src/challenger_expr.rs:279-296
    word = int.from_bytes(bytes([1, 1, 1, 1]), "little")
    c = ts.BfChallenger(0, sample_ext=False)
    c.observe(word)
    c.sample()
    c.observe(word)
    assert int(c.sample()[0]) == 1103171332


def test_clone_is_independent(lib):
    a = ts.BfChallenger()
    a.observe(7)
    b = a.clone()
    assert a.sample().tolist() == b.sample().tolist()
    a.observe(1)
    assert a.state().tolist() != b.state().tolist()
