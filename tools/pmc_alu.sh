#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_alu
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d $O/a -o alu -- python3 $R/tools/prof_alu.py > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
python3 - <<PY
import sqlite3, glob, re
from collections import defaultdict
db = glob.glob("$O/a/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
def key(name):
    m = re.search(r"(k_alu_\w+)", name)
    return m.group(1) if m else ""
dur = defaultdict(list)
for name, d in con.execute("select name, duration from kernels"):
    dur[key(name)].append(d)
cnt = defaultdict(lambda: defaultdict(list))
for name, c, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
    cnt[key(name)][c].append(v)
for k in dur:
    if not k: continue
    d = sorted(dur[k])[len(dur[k]) // 2]
    busy = max(cnt[k]["SQ_BUSY_CYCLES"]); insts = max(cnt[k]["SQ_INSTS_VALU"]); act = max(cnt[k]["SQ_ACTIVE_INST_VALU"])
    print(f"{k}: median {d/1e3:.1f} us  SQ_BUSY_CYCLES/32 = {busy/32:.4g} -> {busy/32/d:.3f} GHz;  INSTS_VALU {insts:.4g}  ACTIVE_INST_VALU {act:.4g} quad-cycles -> VALU busy {act*4/1024/(busy/32):.3f}; {insts/1024/d:.3f} wave-instr/ns/SIMD")
PY
find $O -name "*.db" -delete
