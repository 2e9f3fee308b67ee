"""Where k_fri_tail's time goes: in-kernel time stamps (s_memrealtime, 100 MHz) per round and phase, from
the diagnostic build (python -m tapstark_amd.build -DTS_TAIL_STAMPS -> tap-stark_amd/lib_diag/):

    python -m tapstark_amd.build -DTS_TAIL_STAMPS
    TS_LIB_PATH=tap-stark_amd/lib_diag/libtapstark_hip.so python tools/tail_stamps.py
"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, bench, tapstark_amd as ts
from tapstark_amd import _lib
ctx = ts.default_context()
lib = C.CDLL(_lib.LIB_PATH)
for name in ("config3",):
    air, _, pis, desc, cfg, shape, gen = bench.workload(name, 20, False)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    for _ in range(3): ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    ctx.synchronize()
    buf = (C.c_ulonglong * 512)()
    lib.ts_debug_stamps(buf)
    st = [(buf[2*i], buf[2*i+1]) for i in range(256)]
    t0 = st[0]
    def us(i): return (st[i][0] - t0[0]) / 100.0
    print("load", us(1), "clock GHz over kernel", (st[101][1]-st[0][1]) / ((st[101][0]-st[0][0]) * 10.0))
    t = 0
    while st[2 + 5*t][0]:
        a = [us(2+5*t+k) for k in range(5)]
        print(f"round {t}: start {a[0]:.2f} leaf {a[1]-a[0]:.2f} levels {a[2]-a[1]:.2f} sponge {a[3]-a[2]:.2f} fold {a[4]-a[3]:.2f}")
        t += 1
        if t > 15: break
    print("end", us(100), us(101))
    tb = (C.c_ulonglong * 64)()
    lib.ts_debug_tree_stamps(tb)
    names = ["start", "fill done (block 0)", "block levels done", "stores drained + barrier", "ticket taken", "finisher: is last",
             "acquire done", "sub-roots staged + reduced", "sponge start", "end"]
    print("last k_fri_round launch (us from workgroup 0's start):")
    for i, nm in enumerate(names):
        print(f"  {nm:32s} {(tb[i] - tb[0]) / 100.0:8.2f}   shader clock since the previous stamp {((tb[32 + i] - tb[32 + i - 1]) / max(1, (tb[i] - tb[i - 1]) * 10)) if i else 0:5.2f} GHz")
    print("  block 0's levels (us each):", [round((tb[10 + l] - (tb[10 + l - 1] if l else tb[1])) / 100.0, 2) for l in range(9) if tb[10 + l]])
