set -e
mkdir -p gpurun_out/r3
for cpw in 1 2 4; do
TS_LDE_FWD_CPW=$cpw python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > gpurun_out/r3/c4_cpw$cpw.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/c4_cpw$cpw.json')); print('C4 cpw', $cpw, d['ms_per_step'], d['single_proof_latency_ms'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:3]: print('   ', k, v['ms_per_proof'], v['avg_launch_ms'])"
done
