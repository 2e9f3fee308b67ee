#!/bin/bash
# Collects the round's evidence on the GPU box (run from the repo root through gpurun); everything lands
# under gpurun_out/r3/final/ and is then copied into profiles/r03_* by hand:
#   bench lines (driver protocol for config 3; configs 2, 4, 5; sharded world-1; the 2-rank shared-GPU
#   rehearsal of --gpus 2), rocprofv3 kernel-trace stats, SQ counters, FETCH_SIZE / WRITE_SIZE passes and
#   the raw request counters for configs 3 and 4, kernel stats for configs 2 and 5, the fold benchmark.
set -e
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3/final
mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_config3_k20.json 2> $O/bench.err
python3 bench.py --workload config2 --streams 8 --steps 48 --warmup 8 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > $O/bench_config4.json 2>> $O/bench.err
python3 bench.py --workload config5 --streams 2 --steps 8 --warmup 2 --windows 2 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python3 bench.py --workload config4 --mode sharded --steps 4 --warmup 1 --windows 1 --no-cpu-baseline --headline-only > $O/bench_config4_sharded_world1.json 2>> $O/bench.err
TS_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 8 --warmup 2 --headline-only > $O/bench_gpus2_shared_gpu_rehearsal.json 2>> $O/bench.err
python3 bench.py --workload fold > $O/fold_even_odd.json 2>> $O/bench.err
echo "benches done"
for cfg in config3 config4; do
  bash tools/pmc_sq.sh $cfg > $O/sq_$cfg.log 2>&1 || { tail -20 $O/sq_$cfg.log; exit 1; }
  cp gpurun_out/prof_sq/sq_table.txt $O/${cfg}_sq_counters.txt
done
cd /tmp && export TMPDIR=/tmp
for cfg in config3 config4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${cfg}_$c -o pmc -- python3 $R/tools/prof_prove.py 2 $cfg > $O/pmc_${cfg}_$c.log 2>&1 || { echo "pmc $cfg $c failed"; tail -5 $O/pmc_${cfg}_$c.log; exit 1; }
  done
  timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $O/raw_$cfg -o pmc -- python3 $R/tools/prof_prove.py 2 $cfg > $O/raw_$cfg.log 2>&1 || { echo "raw $cfg failed"; exit 1; }
done
for cfg in config3 config2 config4 config5; do
  n=3; [ $cfg = config4 ] && n=2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -o kt -- python3 $R/tools/prof_prove.py $n $cfg > $O/kt_$cfg.log 2>&1 || { echo "kt $cfg failed"; tail -5 $O/kt_$cfg.log; exit 1; }
done
cd $R
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -delete
for cfg in config3 config4; do
  python3 tools/pmc_summary.py $O/pmc_${cfg}_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_${cfg}_WRITE_SIZE/pmc_counter_collection.csv 2 $O/${cfg}_pmc_traffic.json $cfg > $O/${cfg}_traffic.txt
  python3 tools/pmc_fetch_raw.py $O/raw_$cfg/pmc_counter_collection.csv 2 $cfg > $O/fetch_raw_$cfg.json
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3/final/bench_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["ms_per_step"], 4), d.get("extra", {}).get("windows_ms_per_step"), d.get("single_proof_latency_ms"))
PY
tail -4 $O/config3_sq_counters.txt
tail -3 $O/config4_sq_counters.txt
