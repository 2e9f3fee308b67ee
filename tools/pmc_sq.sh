#!/bin/bash
# SQ counters of the prover's kernels (counters only, two passes of <= 8 counters): gpurun_out/prof_sq/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/a -o sq -- python3 $R/tools/prof_prove.py 1 ${1:-config3} > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE -d $O/b -o sq -- python3 $R/tools/prof_prove.py 1 ${1:-config3} > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
python3 - <<PY
import sqlite3, re, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for db in sorted(set(glob.glob("$O/*/**/*.db", recursive=True))):
    con = sqlite3.connect(db)
    for name, cname, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))).replace("ts::", "")
        vals[k][cname].append(float(v))
counters = sorted({c for k in vals for c in vals[k]})
with open("$O/sq_table.txt", "w") as f:
    f.write(f"{'kernel (largest launch)':34s} " + " ".join(f"{c[:18]:>18s}" for c in counters) + "\n")
    for k in sorted(vals, key=lambda k: -max(vals[k].get("SQ_WAVE_CYCLES", [0]))):
        f.write(f"{k[:34]:34s} " + " ".join(f"{max(vals[k][c]) if c in vals[k] else 0:18.4g}" for c in counters) + "\n")
with open("$O/sq_table.txt", "a") as f:
    f.write("\nlaunches and SQ_INSTS_VALU summed per kernel:\n")
    for k in sorted(vals, key=lambda k: -sum(vals[k].get("SQ_INSTS_VALU", [0]))):
        f.write(f"  {k[:40]:40s} x{len(vals[k].get('SQ_INSTS_VALU', [])):4d}  {sum(vals[k].get('SQ_INSTS_VALU', [0])):.4g}\n")
# whole-proof totals (one proof was run): everything except the one-off table builders and the
# synthetic trace generator
skip = ("k_build_", "k_trace_", "k_selectors")
tot_i = sum(sum(vals[k].get("SQ_INSTS_VALU", [])) for k in vals if not any(x in k for x in skip))
tot_b = sum(sum(vals[k].get("SQ_BUSY_CYCLES", [])) for k in vals if not any(x in k for x in skip))
with open("$O/sq_table.txt", "a") as f:
    f.write(f"\nwhole proof: SQ_INSTS_VALU {tot_i:.4g} wave-instructions, SQ_BUSY_CYCLES/32 {tot_b/32:.4g} cycles\n")
    f.write(f"  VALU issue ceiling (1024 SIMDs, one wave64 instruction per 4 cycles; tools/pmc_alu.sh measures exactly "
            f"that rate in register-resident loops): {tot_i*4/1024/2.0e9*1e3:.3f} ms at 2.0 GHz, {tot_i*4/1024/2.3e9*1e3:.3f} ms at 2.3 GHz\n")
    f.write(f"  issue slots used while kernels run: {tot_i*4/1024/(tot_b/32):.3f}\n")
print(open("$O/sq_table.txt").read())
PY
find $O -name "*.db" -delete
