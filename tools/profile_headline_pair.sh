#!/bin/bash
# The headline bench line and the rocprofv3 kernel stats of the same workload from ONE box (their
# per-kernel averages are compared): gpurun_out/r05/final/{bench_config3_k20.json, kt_config3/}
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05}/final
mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_config3_k20.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_config3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_config3 -o kt -- python3 $R/tools/prof_prove.py 10 config3 > $O/kt_config3.log 2>&1
find $O/kt_config3 -name "*kernel_trace.csv" -delete
head -3 $O/kt_config3/kt_kernel_stats.csv | cut -c1-50,150-230
python3 - <<PY
import json
d = json.loads(open("$O/bench_config3_k20.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline_ntt"]["avg_launch_ms"], d["single_proof_latency_ms"])
PY
