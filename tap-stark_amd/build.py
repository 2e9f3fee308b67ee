"""Builds libtapstark_hip.so (HIP kernels + C ABI + host driver) in-tree with hipcc for gfx950.

    python -m tapstark_amd.build            # or: from tapstark_amd.build import build; build()

hipcc cross-compiles without a GPU.  Objects land in tap-stark_amd/_build/, the library in
tap-stark_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_build")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libtapstark_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_mtime() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(os.path.dirname(PKG), "include", "tapstark.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _headers_mtime()
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", CSRC]
    jobs = []
    objs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if not force and os.path.exists(op) and os.path.getmtime(op) > max(os.path.getmtime(sp), hdr_m):
            continue
        if src.endswith(".hip"):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-x", "hip", *common, "-c", sp, "-o", op]
        else:  # host sources include HIP headers: same single target, no stray default-arch device code
            cmd = [hipcc, f"--offload-arch={ARCH}", *common, "-c", sp, "-o", op]
        jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
