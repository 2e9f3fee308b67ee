"""Copies what tools/profile_round.sh left under gpurun_out/<tag>/final/ into profiles/<tag>_* (the
tracked evidence the docs cite):  python tools/profile_collect.py r04"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag, "final")
dst = os.path.join(root, "profiles")
plan = {
    "bench_config3_k20.json": "bench_config3_k20.json", "bench_config2.json": "bench_config2.json",
    "bench_config4.json": "bench_config4.json", "bench_config5.json": "bench_config5.json",
    "bench_gpus2_shared_gpu_rehearsal.json": "bench_gpus2_shared_gpu_rehearsal.json",
    "bench_rccl_world1_blocks.json": "bench_rccl_world1_blocks.json",
    "fold_even_odd.json": "fold_even_odd.json", "config4_shard_stages.json": "config4_shard_stages.json",
    "config5_shard_stages.json": "config5_shard_stages.json", "fri_hipgraph_latency.txt": "fri_hipgraph_latency.txt",
    "config3_sq_counters.txt": "config3_sq_counters.txt", "config4_sq_counters.txt": "config4_sq_counters.txt",
    "config3_pmc_traffic.json": "pmc_traffic.json", "config4_pmc_traffic.json": "config4_pmc_traffic.json",
    "config2_pmc_traffic.json": "config2_pmc_traffic.json", "config5_pmc_traffic.json": "config5_pmc_traffic.json",
    "config2_sq_counters.txt": "config2_sq_counters.txt", "config5_sq_counters.txt": "config5_sq_counters.txt",
    "power_vs_working_set.json": "power_vs_working_set.json", "power_per_stage.json": "power_per_stage.json",
    "chain_stamps.txt": "chain_stamps.txt", "config3_launch_sequence.txt": "config3_launch_sequence.txt",
    "config2_launch_sequence.txt": "config2_launch_sequence.txt",
}
for cfg in ("config3", "config2", "config4", "config5"):
    plan[f"kt_{cfg}/kt_kernel_stats.csv"] = f"{cfg}_rocprofv3_kernel_stats.csv"
for s, d in plan.items():
    p = os.path.join(src, s)
    if not os.path.exists(p):
        print("missing", s)
        continue
    out = os.path.join(dst, f"{tag}_{d}")
    if s.endswith(".json") and s.startswith("bench_"):  # one pretty-printed record per file
        rec = json.loads(open(p).read().strip().splitlines()[-1])
        json.dump(rec, open(out, "w"), indent=1)
        open(out, "a").write("\n")
    else:
        shutil.copyfile(p, out)
    print("->", os.path.relpath(out, root))
