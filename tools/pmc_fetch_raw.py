"""Per-kernel raw read-request counters (rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_BUBBLE_sum, one pass) against kernels whose read bytes are known exactly, to settle how FETCH_SIZE
has to be corrected on gfx950 per kernel.

    pmc_fetch_raw.py counter_collection.csv N_PROOFS config > out.json

FETCH_SIZE (rocprofv3 -L) = BUBBLE*128 + (RDREQ - BUBBLE - RDREQ_32B)*64 + RDREQ_32B*32 bytes.
Hypotheses per kernel: A = that formula as it is; B = every non-32-byte request is 128 bytes
(the guide's "double it"): RDREQ_32B*32 + (RDREQ - RDREQ_32B)*128.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("ts::", "")


def main():
    path, n_proofs, cfg = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    shape = {"config3": (1 << 20, 64, 2, 2), "config4": (1 << 22, 64, 4, 2)}[cfg]
    n, w, b, qd = shape
    N, W = n << b, w + 4 * qd
    known_reads = {  # bytes READ per proof, exactly (in-place or single-pass kernels)
        "k_lde_fwd_contig": 4 * N * W, "k_reduce_fused": 4 * N * W, "k_leaf_hash_strided": 4 * N * w,
        "k_intt_contig": 4 * n * W, "k_transpose_bitrev": 4 * n * w}
    tot = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(path)):
        tot[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
    out = {}
    for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", 0)):
        rd, r32, bub = c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_BUBBLE_sum", 0)
        a = (bub * 128 + (rd - bub - r32) * 64 + r32 * 32) / n_proofs
        bb = (r32 * 32 + (rd - r32) * 128) / n_proofs
        stem = k.split("<")[0]
        kn = known_reads.get(stem)
        if stem == "k_intt_contig" or stem == "k_lde_fwd_contig":  # sum over template instances below
            pass
        out[k] = {"RDREQ": rd / n_proofs, "RDREQ_32B": r32 / n_proofs, "BUBBLE": bub / n_proofs,
                  "bytes_formula_A": a, "bytes_all_128B_B": bb, "known_read_bytes": kn}
    # instances of one template together
    groups = defaultdict(lambda: [0.0, 0.0])
    for k, v in out.items():
        stem = k.split("<")[0]
        if stem in known_reads:
            groups[stem][0] += v["bytes_formula_A"]
            groups[stem][1] += v["bytes_all_128B_B"]
    calib = {s: {"known": known_reads[s], "A_over_known": round(g[0] / known_reads[s], 3),
                 "B_over_known": round(g[1] / known_reads[s], 3)} for s, g in groups.items()}
    json.dump({"config": cfg, "calibration": calib, "kernels": out}, sys.stdout, indent=1)


main()
