#!/bin/bash
# SQ counters of the Merkle commit kernels alone (tools/time_leaf_tree.py), for the knob settings given
# as VAR=value arguments: pmc_leaf_tree.sh TAG [VAR=value ...]  ->  gpurun_out/r5/pmc_TAG.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/r5/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES -d $O/a -o sq -- python3 $R/tools/time_leaf_tree.py 22 > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY -d $O/b -o sq -- python3 $R/tools/time_leaf_tree.py 22 > $O/b.log 2>&1 || { tail -5 $O/b.log; echo "(pass b failed)"; }
python3 - <<PY
import sqlite3, re, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for db in sorted(set(glob.glob("$O/*/**/*.db", recursive=True))):
    con = sqlite3.connect(db)
    for name, cname, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name)).replace("ts::", "")
        vals[k][cname].append(float(v))
counters = sorted({c for k in vals for c in vals[k]})
with open("$R/gpurun_out/r5/pmc_$TAG.txt", "w") as f:
    f.write("settings: $*\n")
    f.write(f"{'kernel (largest launch)':44s} " + " ".join(f"{c[3:19]:>16s}" for c in counters) + f" {'issue fill':>10s}\n")
    for k in sorted(vals, key=lambda k: -max(vals[k].get("SQ_WAVE_CYCLES", [0]))):
        if not ("leaf" in k or "merkle" in k):
            continue
        i, b = max(vals[k].get("SQ_INSTS_VALU", [0])), max(vals[k].get("SQ_BUSY_CYCLES", [1]))
        f.write(f"{k[:44]:44s} " + " ".join(f"{max(vals[k][c]) if c in vals[k] else 0:16.4g}" for c in counters)
                + f" {i * 4 / 1024 / (b / 32):10.3f}\n")
print(open("$R/gpurun_out/r5/pmc_$TAG.txt").read())
PY
find $O -name "*.db" -delete
