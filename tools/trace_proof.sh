#!/bin/bash
# The launch sequence of ONE proof: every kernel in stream order with its duration and the gap to the
# previous kernel's end (rocprofv3 --kernel-trace; the last of 3 proofs).
#     bash tools/trace_proof.sh TAG config3 [VAR=value ...]   ->  gpurun_out/r5/trace_TAG.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; CFG=$2; shift; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/r5/trace_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python3 $R/tools/prof_prove.py 3 $CFG > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - <<PY
import csv, glob, re
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).replace("ts::", "").replace("(anonymous namespace)::", "")
# the last proof starts at the last trace generator / transpose
starts = [i for i, r in enumerate(rows) if "k_trace_" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
out = open("$R/gpurun_out/r5/trace_$TAG.txt", "w")
out.write("settings: $CFG $*\n")
t_end = None
tot = gaps = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if t_end is None else (s - t_end) / 1e3
    out.write(f"{short(r['Kernel_Name'])[:52]:52s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):7d}  {(e - s) / 1e3:9.1f} us  gap {gap:7.1f}\n")
    tot += (e - s) / 1e3
    gaps += max(gap, 0)
    t_end = e
out.write(f"total kernel time {tot:.1f} us, gaps {gaps:.1f} us, span {(t_end - int(rows[0]['Start_Timestamp'])) / 1e3:.1f} us, {len(rows)} launches\n")
out.close()
print(open("$R/gpurun_out/r5/trace_$TAG.txt").read())
PY
rm -rf $O
