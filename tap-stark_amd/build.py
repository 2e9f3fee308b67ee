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


def build(force: bool = False, verbose: bool = False, asan: bool = False, defines: tuple = ()) -> str:
    """defines: diagnostic builds (-D...; e.g. ("TS_TAIL_STAMPS",) = in-kernel time stamps in k_fri_tail,
    tools/tail_stamps.py) go to lib_diag/ and are used through TS_LIB_PATH; the product build has none.

    asan=True: the HOST sources (.cpp: C ABI, wire formats, verifier, challenger, host prover) are
    built with AddressSanitizer + UBSan into lib_asan/ (device code is not instrumented: GPU
    sanitizers are not available on the pool).  Run the CPU suite on it with

        TS_LIB_PATH=tap-stark_amd/lib_asan/libtapstark_hip.so ASAN_OPTIONS=detect_leaks=0 \
        LD_PRELOAD="$(python -m tapstark_amd.build --asan-runtime)" python -m pytest tests -m "not gpu"
    """
    obj_dir = OBJ + ("_asan" if asan else "_diag" if defines else "")
    lib_dir = LIBDIR + ("_asan" if asan else "_diag" if defines else "")
    lib = os.path.join(lib_dir, "libtapstark_hip.so")
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(lib_dir, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _headers_mtime()
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", CSRC, *("-D" + d for d in defines)]
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-shared-libsan"]
    jobs = []
    objs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, src + ".o")
        objs.append(op)
        if not force and os.path.exists(op) and os.path.getmtime(op) > max(os.path.getmtime(sp), hdr_m):
            continue
        if src.endswith(".hip"):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-x", "hip", *common, "-c", sp, "-o", op]
        elif asan:  # plain host C++ (no offload), instrumented
            cmd = [_clangxx(), "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(_rocm(), "include"),
                   "-O1", "-g", *common[1:], *san, "-c", sp, "-o", op]
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
    if jobs or not os.path.exists(lib):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *(san if asan else []), "-o", lib, *objs])
    # the out-of-process hiprtc helper of the background AIR specialisation (plain host C++, beside the library)
    helper_src = os.path.join(PKG, "jitc", "ts_jitc.cpp")
    helper = os.path.join(lib_dir, "ts_jitc")
    if force or not os.path.exists(helper) or os.path.getmtime(helper) < os.path.getmtime(helper_src):
        run(["g++", "-O2", "-std=c++17", "-o", helper, helper_src, "-ldl"])
    return lib


def _rocm() -> str:
    return os.path.dirname(os.path.dirname(os.path.realpath(_hipcc())))


def _clangxx() -> str:
    for cand in (os.path.join(_rocm(), "lib", "llvm", "bin", "clang++"), "/opt/rocm/lib/llvm/bin/clang++"):
        if os.path.exists(cand):
            return cand
    raise RuntimeError("clang++ of the ROCm toolchain not found")


def device_disassembly(source: str) -> dict:
    """{mangled kernel name: [instruction lines]} of the gfx950 code object inside the object file of
    `source` (e.g. "merkle.hip"); builds first if needed.  Used by tests/test_build_isa.py to check
    the memory-scope bits of the cross-workgroup hand-offs."""
    import re
    import tempfile

    build()
    obj = os.path.join(OBJ, source + ".o")
    objdump = os.path.join(_rocm(), "lib", "llvm", "bin", "llvm-objdump")
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(obj))
        shutil.copy(obj, tmp)
        subprocess.run([objdump, "--offloading", tmp], check=True, capture_output=True)  # unbundles beside the input
        cos = [f for f in os.listdir(td) if "amdgcn" in f]
        if not cos:
            raise RuntimeError(f"no gfx950 code object in {obj}")
        text = subprocess.run([objdump, "-d", os.path.join(td, cos[0])], check=True, capture_output=True,
                              text=True).stdout
    out = {}
    for block in re.split(r"\n(?=[0-9a-f]{16} <)", text):
        m = re.match(r"[0-9a-f]{16} <([^>]+)>:", block)
        if m:
            out[m.group(1)] = [l.split("//")[0].strip() for l in block.splitlines()[1:] if l.strip()]
    return out


def asan_runtime() -> str:
    """What LD_PRELOAD needs for the asan build under an uninstrumented python: clang's ASan runtime,
    and libstdc++ so that the runtime's __cxa_throw interceptor finds the real one at start-up (the
    library reports refused inputs with C++ exceptions internally)."""
    import glob

    rt = glob.glob(os.path.join(_rocm(), "lib", "llvm", "lib", "clang", "*", "lib", "linux",
                                "libclang_rt.asan-x86_64.so"))
    r = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True)
    return " ".join([rt[0], os.path.realpath(r.stdout.strip())])


if __name__ == "__main__":
    if "--asan-runtime" in sys.argv:
        print(asan_runtime())
    else:
        print(build(force="--force" in sys.argv, verbose=True, asan="--asan" in sys.argv,
                    defines=tuple(a[2:] for a in sys.argv if a.startswith("-D"))))
