"""Soak of the cross-workgroup hand-off in the whole-tree kernels (merkle_tree.hpp tree_body; every
proof builds ~14 trees through it).  The sc1 / ticket form is used with two or three workgroups per
CU and several proofs in flight -- outside the one-workgroup-per-CU envelope the microarchitecture
guide measured it in -- so beside the acquire the finisher now runs the suite keeps this
test: six lanes sharing the CUs, every proof of a trace byte-identical to the first proof of that
trace.  Short by default (240 proofs, a few seconds); TS_SOAK_PROOFS=1500 is the 9000-proof form of
tools/soak.py that DESIGN.md quotes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.slow
def test_handoff_soak_six_lanes():
    per_lane = int(os.environ.get("TS_SOAK_PROOFS", "40"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), str(per_lane), "6"],
                       capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "mismatches: 0" in r.stdout
