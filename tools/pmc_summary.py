"""profiles/*_pmc_traffic.json from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate
passes) of tools/prof_prove.py.

    pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv N_PROOFS OUT.json [config]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both
counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads,
so it is doubled before comparing with a byte count.  Round 3 settled the doubling per kernel with
the raw request counters (tools/pmc_fetch_raw.py, profiles/r03_fetch_raw_config{3,4}.json): counting
every non-32-byte read request as 128 bytes reproduces the exactly known read bytes of the in-place
and single-pass kernels to 0-3 % at both shapes; where it says more than the data (the contiguous NTT
passes at 2^22 rows: 1.12-1.26x) the surplus is real -- twiddle words missing the XCD's L2 (they are
counted whether the Infinity Cache or HBM serves them).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)  # drop the argument list
    name = name.replace("ts::", "")
    # k_leaf_tree<2, StridedLeaf> -> k_leaf_tree<2,strided>: the names the library's kernel timers and
    # bench.py's byte model use (leaf_tree.hpp Leaf::name)
    m = re.match(r"k_leaf_tree<(\d), (\w+)(<(true|false)> ?)?>", name)
    if m:
        kind = {"StridedLeaf": "strided", "TableLeaf": "table", "EfPairLeaf": "ef_pairs",
                "FriLeaf": "fri_fold" if m.group(4) == "true" else "fri_leaf"}.get(m.group(2), m.group(2))
        name = f"k_leaf_tree<{m.group(1)},{kind}>"
    return name


def load(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = short(row["Kernel_Name"])
        tot[k] += float(row["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    f_csv, w_csv, n_proofs, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    cfg_name = sys.argv[5] if len(sys.argv) > 5 else "config3"
    shape = {"config3": (1 << 20, 64, 2, 2), "config2": (1 << 20, 2, 2, 1), "config4": (1 << 22, 64, 4, 2),
             "config5": (1 << 20, 163, 4, 1)}[cfg_name]
    fetch, cnt = load(f_csv, "FETCH_SIZE")
    write, _ = load(w_csv, "WRITE_SIZE")
    sys.path.insert(0, ".")
    from bench import algorithmic_bytes_per_proof
    alg = algorithmic_bytes_per_proof(*shape)
    if "k_leaf_hash_strided" not in fetch:  # no strided launch: the table kernel hashed everything
        alg["k_leaf_hash<1>"] = alg["k_leaf_hash<2>"]
    if "k_merkle_level<2>" not in fetch:  # every per-level launch went through the <1> kernel
        alg["k_merkle_level<1>"] += alg["k_merkle_level<2>"]
    kernels = {}
    for k in sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0))):
        a = (alg.get(k) or alg.get(f"({k})") or (alg["k_lde_mid<1>"] if "k_lde_mid" in k else None)
             or (alg["k_merkle_tree"] if "k_merkle_tree" in k else None) or alg.get(k.split("<")[0]))
        kernels[k] = {"launches_per_proof": cnt[k] / n_proofs,
                      "fetch_bytes_per_proof_corrected": 2 * 1024 * fetch[k] / n_proofs,
                      "write_bytes_per_proof": 1024 * write.get(k, 0.0) / n_proofs,
                      "alg_bytes_per_proof": a}
    json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on "
                           f"tools/prof_prove.py, {n_proofs} proofs of {cfg_name}; KiB units x1024; FETCH_SIZE "
                           "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced "
                           "reads); per proof", "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        t = v["fetch_bytes_per_proof_corrected"] + v["write_bytes_per_proof"]
        print(f"{k:40s} traffic {t/1e6:9.1f} MB  alg {(v['alg_bytes_per_proof'] or 0)/1e6:9.1f} MB")


main()
