"""rocprofv3 (ROCm 7.2) writes rocpd sqlite databases by default.  This turns them into the small
text summaries committed under profiles/:

    rocpd_stats.py stats  kt_results.db OUT.csv                 # the --stats kernel table
    rocpd_stats.py pmc    FETCH.db WRITE.db N_PROOFS OUT.json CONFIG   # profiles/*_pmc_traffic.json
"""
import csv
import json
import re
import sqlite3
import sys
from collections import defaultdict


def stats(db, out):
    con = sqlite3.connect(db)
    rows = defaultdict(list)
    for name, dur in con.execute("select name, duration from kernels"):
        rows[name].append(dur)
    total = sum(sum(v) for v in rows.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3),
                        min(v), max(v)])


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("ts::", "").replace("(anonymous namespace)::", "")


def load(db, counter):
    con = sqlite3.connect(db)
    tot, cnt = defaultdict(float), defaultdict(int)
    for name, val in con.execute("select kernel_name, value from counters_collection where counter_name = ?",
                                 (counter,)):
        k = short(name)
        tot[k] += float(val)
        cnt[k] += 1
    return tot, cnt


def pmc(f_db, w_db, n_proofs, out, cfg_name):
    sys.path.insert(0, ".")
    from bench import algorithmic_bytes_per_proof
    shape = {"config3": (1 << 20, 64, 2, 2), "config2": (1 << 20, 2, 2, 1), "config4": (1 << 22, 64, 4, 2),
             "config5": (1 << 20, 163, 4, 1)}[cfg_name]
    fetch, cnt = load(f_db, "FETCH_SIZE")
    write, _ = load(w_db, "WRITE_SIZE")
    alg = algorithmic_bytes_per_proof(*shape)
    if "k_leaf_hash_strided" not in fetch:
        alg["k_leaf_hash<1>"] = alg["k_leaf_hash<2>"]
    if "k_merkle_level<2>" not in fetch:
        alg["k_merkle_level<1>"] += alg["k_merkle_level<2>"]
    kernels = {}
    # prof_prove.py runs n_proofs proofs after nothing else: every launch counts
    for k in sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0))):
        a = (alg.get(k) or alg.get(f"({k})") or (alg["k_lde_mid<1>"] if "k_lde_mid" in k else None))
        kernels[k] = {"launches_per_proof": cnt[k] / n_proofs,
                      "fetch_bytes_per_proof_corrected": 2 * 1024 * fetch[k] / n_proofs,
                      "write_bytes_per_proof": 1024 * write.get(k, 0.0) / n_proofs,
                      "alg_bytes_per_proof": a}
    json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, counters only) on "
                           f"tools/prof_prove.py, {n_proofs} proofs of {cfg_name}; KiB units x1024; FETCH_SIZE "
                           "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); per proof",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        t = v["fetch_bytes_per_proof_corrected"] + v["write_bytes_per_proof"]
        print(f"{k:44s} x{v['launches_per_proof']:6.1f} traffic {t/1e6:10.1f} MB  alg {(v['alg_bytes_per_proof'] or 0)/1e6:10.1f} MB")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6])
