"""Per-kernel averages of rocprofv3 --pmc counters: pmc_table.py a_counter_collection.csv [more.csv ...]"""
import csv, re, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"])).replace("ts::", "")
        vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = sorted({c for k in vals for c in vals[k]})
print(f"{'kernel':34s} " + " ".join(f"{c[:16]:>16s}" for c in counters))
for k in sorted(vals, key=lambda k: -sum(vals[k].get("SQ_BUSY_CYCLES", vals[k].get(counters[0], [0])))):
    # the biggest launch of each kernel (the trace LDE / tree), not the average over sizes
    print(f"{k[:34]:34s} " + " ".join(f"{max(vals[k][c]) if c in vals[k] else 0:16.4g}" for c in counters))
