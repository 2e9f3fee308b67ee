"""Where do the sporadic slow proofs of bench.py's timed windows come from?  examples/prove_stream.cpp (no Python in
the process) shows none in 1600 proofs.  This is bench.py's lane loop reduced to the bone -- S threads, one context
each, pre-generated traces, the start gate -- with the suspects switched on one at a time:

    python tools/py_stream_stalls.py [n_proofs=400] [torch] [pool] [gcon]

torch = import torch first (as bench.py does), pool = lanes through a ThreadPoolExecutor instead of raw threads,
gcon = leave the garbage collector on.  Prints ms/proof, median / p99 / max proof wall time and every proof above
1.4 x median with its start time."""
import os
import sys
import threading
import time

flags = set(sys.argv[2:])
if "nothp" in flags:  # no transparent huge pages for this process (khugepaged collapsing the interpreter's heap?)
    import ctypes
    ctypes.CDLL("libc.so.6").prctl(41, 1, 0, 0, 0)  # PR_SET_THP_DISABLE
if "noblas" in flags:
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    os.environ["OMP_NUM_THREADS"] = "1"
if "mallopt" in flags:  # keep glibc from mmap'ing / munmap'ing every buffer above 128 KiB (numpy's proof buffers)
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.mallopt(-3, 1 << 30)   # M_MMAP_THRESHOLD
    libc.mallopt(-1, 1 << 30)   # M_TRIM_THRESHOLD
if "torch" in flags:
    import torch  # noqa: F401
    if "cuda" in flags:
        torch.cuda.device_count()
sys.path.insert(0, os.getcwd())
import gc  # noqa: E402

import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.airs import SynthMulAir  # noqa: E402

n_proofs = int(sys.argv[1]) if len(sys.argv) > 1 else 400
S, n, w, cfg = 4, 1 << 20, 64, (2, 28, 8)
pis = np.zeros(0, dtype=np.uint32)
if "perthread" in flags:
    # as examples/prove_stream.cpp: every lane thread makes its OWN context, AIR and traces, primes, then waits at a
    # barrier; nothing of a lane is ever touched by another thread
    tape = ts.air_tape(SynthMulAir(64), 0)
    bar = threading.Barrier(S + 1)
    lat = [[] for _ in range(S)]
    t_begin_box = [0.0]

    def lane_main(l):
        import ctypes as C
        from tapstark_amd import _lib
        c = ts.Context(0)
        conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
        ca = ts.CompiledAir(c, tape)
        my = [ts.DeviceMatrix.synth_mul(c, n, w) for _ in range(l, n_proofs, S)]
        ts.prove(conf, ca, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(c, n, w), pis)
        c.synchronize()
        L = _lib.lib()
        out = np.zeros(1 << 20, dtype=np.uint32)
        outp = out.ctypes.data_as(_lib.u32p)
        cfgc = conf.pcs.fri._c()
        nw = C.c_size_t()
        bar.wait()
        for m in my:
            ch = ts.BfChallenger()
            t = time.perf_counter()
            rc = L.ts_prove(c.h, C.byref(cfgc), ca.h, ch.h, m.h, None, 0, outp, len(out), C.byref(nw))
            lat[l].append((t, time.perf_counter() - t))
            assert rc == 0
        c.synchronize()

    th = [threading.Thread(target=lane_main, args=(l,)) for l in range(S)]
    for t in th:
        t.start()
    bar.wait()
    t_begin = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t_begin
    allv = sorted(e[1] for v in lat for e in v)
    med = allv[len(allv) // 2]
    slow = [(l, 1e3 * (e[0] - t_begin), 1e3 * e[1]) for l, v in enumerate(lat) for e in v if e[1] > 1.4 * med]
    print(f"flags {sorted(flags)}: {n_proofs} proofs, {1e3 * dt / n_proofs:.3f} ms/proof (no gate); proof wall time median {1e3 * med:.2f} p99 "
          f"{1e3 * allv[int(0.99 * len(allv))]:.2f} max {1e3 * allv[-1]:.2f} ms; above 1.4 x median: {len(slow)} "
          + " ".join(f"[lane {l} @{t:.0f} ms: {d:.1f}]" for l, t, d in slow[:12]), flush=True)
    sys.exit(0)
lanes = []
for _ in range(S):
    c = ts.Context(0)
    lanes.append((c, ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c)), ts.CompiledAir(c, ts.air_tape(SynthMulAir(64), 0))))
mats = [ts.DeviceMatrix.synth_mul(lanes[i % S][0], n, w) for i in range(n_proofs)]
for c, conf, ca in lanes:
    ts.prove(conf, ca, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(c, n, w), pis)
    c.synchronize()
t0 = time.perf_counter()
ts.prove(lanes[0][1], lanes[0][2], ts.BfChallenger(), ts.DeviceMatrix.synth_mul(lanes[0][0], n, w), pis)
gap = 0.25 * (time.perf_counter() - t0)
lock, last = threading.Lock(), [-1e9]


def gate():
    with lock:
        while True:
            wait = last[0] + gap - time.perf_counter()
            if wait <= 0:
                break
            time.sleep(min(wait, 2e-4))
        last[0] = time.perf_counter()


lat = [[] for _ in range(S)]


def lane_job(l):
    c, conf, ca = lanes[l]
    if "raw" in flags:  # the C entry point itself, one output buffer per lane for the whole run (as the C++ example)
        import ctypes as C
        from tapstark_amd import _lib
        L = _lib.lib()
        out = np.zeros(1 << 20, dtype=np.uint32)
        outp = out.ctypes.data_as(_lib.u32p)
        cfgc = conf.pcs.fri._c()
        nw = C.c_size_t()
        for i in range(l, n_proofs, S):
            ch = ts.BfChallenger()
            gate()
            t = time.perf_counter()
            rc = L.ts_prove(c.h, C.byref(cfgc), ca.h, ch.h, mats[i].h, None, 0, outp, len(out), C.byref(nw))
            lat[l].append((t, time.perf_counter() - t))
            assert rc == 0
        return
    for i in range(l, n_proofs, S):
        gate()
        t = time.perf_counter()
        ts.prove(conf, ca, ts.BfChallenger(), mats[i], pis)
        lat[l].append((t, time.perf_counter() - t))
        mats[i] = None


if "gcon" not in flags:
    gc.collect()
    gc.freeze()
    gc.disable()
t_begin = time.perf_counter()
if "pool" in flags:
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=S) as ex:
        list(ex.map(lane_job, range(S)))
else:
    th = [threading.Thread(target=lane_job, args=(l,)) for l in range(S)]
    for t in th:
        t.start()
    for t in th:
        t.join()
for c, _, _ in lanes:
    c.synchronize()
dt = time.perf_counter() - t_begin
allv = sorted(e[1] for v in lat for e in v)
med = allv[len(allv) // 2]
slow = [(l, 1e3 * (e[0] - t_begin), 1e3 * e[1]) for l, v in enumerate(lat) for e in v if e[1] > 1.4 * med]
print(f"flags {sorted(flags) or ['-']}: {n_proofs} proofs, {1e3 * dt / n_proofs:.3f} ms/proof; proof wall time median {1e3 * med:.2f} p99 "
      f"{1e3 * allv[int(0.99 * len(allv))]:.2f} max {1e3 * allv[-1]:.2f} ms; above 1.4 x median: {len(slow)} "
      + " ".join(f"[lane {l} @{t:.0f} ms: {d:.1f}]" for l, t, d in slow[:12]), flush=True)
