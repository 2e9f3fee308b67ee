#!/bin/bash
# Collects a round's evidence on the GPU box (run from the repo root through gpurun):
#     bash tools/profile_round.sh r06 [a|b]
# Everything lands under gpurun_out/<tag>/final/ and is then copied into profiles/<tag>_* by
# tools/profile_collect.py: bench lines (driver protocol for config 3; configs 2, 4, 5; the 2-rank
# shared-GPU rehearsal of --gpus 2 with both sharded blocks), rocprofv3 kernel-trace stats for configs
# 2-5, SQ counters and FETCH_SIZE / WRITE_SIZE passes for configs 3 and 4 (counters only, separate
# passes), the turn-taking shard-stage tables of configs 4 and 5, the TS_FRI_GRAPH latency A/B, the
# reference's fold benchmark.
set -e
set -o pipefail
TAG=${1:-r06}
PART=${2:-all}   # all | a (bench lines, shard stages, latency, launch sequences) | b (power, counters, kernel traces)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/final
mkdir -p $O
cd $R
if [ $PART != b ]; then
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_config3_k20.json 2> $O/bench.err
python3 bench.py --workload config2 --streams 8 --steps 48 --warmup 8 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > $O/bench_config4.json 2>> $O/bench.err
python3 bench.py --workload config5 --streams 2 --steps 8 --warmup 2 --windows 2 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
TS_BENCH_SHARE_GPU=1 TS_BENCH_SHARD_STEPS=2 python3 bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_gpus2_shared_gpu_rehearsal.json 2>> $O/bench.err
TS_BENCH_FORCE_SHARD_BLOCK=1 TS_BENCH_SHARD_STEPS=2 python3 bench.py --headline-only > $O/bench_rccl_world1_blocks.json 2>> $O/bench.err
python3 bench.py --workload fold > $O/fold_even_odd.json 2>> $O/bench.err
echo "benches done"
python3 tools/shard_stages.py config4 8 > $O/config4_shard_stages.json 2>> $O/bench.err
python3 tools/shard_stages.py config5 8 replicated localq > $O/config5_shard_stages.json 2>> $O/bench.err
echo "shard stages done"
{ python3 tools/latency.py; python3 tools/latency.py; } 2>> $O/bench.err | grep -v amdgpu.ids > $O/fri_hipgraph_latency.txt
if [ -f tap-stark_amd/lib_diag/libtapstark_hip.so ]; then  # python -m tapstark_amd.build -DTS_TAIL_STAMPS
  TS_LIB_PATH=tap-stark_amd/lib_diag/libtapstark_hip.so python3 tools/tail_stamps.py 2>> $O/bench.err | grep -v amdgpu.ids > $O/chain_stamps.txt
fi
bash tools/trace_proof.sh final_c3 config3 > /dev/null && cp gpurun_out/r5/trace_final_c3.txt $O/config3_launch_sequence.txt
bash tools/trace_proof.sh final_c2 config2 > /dev/null && cp gpurun_out/r5/trace_final_c2.txt $O/config2_launch_sequence.txt
cd $R
echo "latency done"
fi
if [ $PART = a ]; then exit 0; fi
python3 tools/power_vs_working_set.py > $O/power_vs_working_set.json 2>> $O/bench.err
python3 tools/power_per_stage.py > $O/power_per_stage.json 2>> $O/bench.err
echo "power done"
for cfg in config3 config2 config4 config5; do
  bash tools/pmc_sq.sh $cfg > $O/sq_$cfg.log 2>&1 || { tail -20 $O/sq_$cfg.log; exit 1; }
  cp gpurun_out/prof_sq/sq_table.txt $O/${cfg}_sq_counters.txt
done
cd /tmp && export TMPDIR=/tmp
for cfg in config3 config2 config4 config5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${cfg}_$c -o pmc -- python3 $R/tools/prof_prove.py 2 $cfg > $O/pmc_${cfg}_$c.log 2>&1 || { echo "pmc $cfg $c failed"; tail -5 $O/pmc_${cfg}_$c.log; exit 1; }
  done
done
for cfg in config3 config2 config4 config5; do
  n=3; [ $cfg = config4 ] && n=2; [ $cfg = config3 ] && n=10
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -o kt -- python3 $R/tools/prof_prove.py $n $cfg > $O/kt_$cfg.log 2>&1 || { echo "kt $cfg failed"; tail -5 $O/kt_$cfg.log; exit 1; }
done
cd $R
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -delete
for cfg in config3 config2 config4 config5; do
  python3 tools/pmc_summary.py $O/pmc_${cfg}_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_${cfg}_WRITE_SIZE/pmc_counter_collection.csv 2 $O/${cfg}_pmc_traffic.json $cfg > $O/${cfg}_traffic.txt
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$TAG/final/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["metric"][-28:], round(d["ms_per_step"], 4), d.get("extra", {}).get("windows_ms_per_step"), d.get("single_proof_latency_ms"), d.get("matches_oracle"))
PY
tail -4 $O/config3_sq_counters.txt
tail -3 $O/config4_sq_counters.txt
