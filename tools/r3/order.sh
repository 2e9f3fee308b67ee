set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -k "test_commit_lde_and_merkle or config4" > gpurun_out/r3/t_order.log 2>&1 || { tail -30 gpurun_out/r3/t_order.log; exit 1; }
tail -2 gpurun_out/r3/t_order.log
for v in 0 1 0 1; do
TS_LDE_FWD_ORDER=$v python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > gpurun_out/r3/c4_order$v.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/c4_order$v.json')); print('C4 order', $v, d['ms_per_step'], d['single_proof_latency_ms'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:2]: print('   ', k, v['ms_per_proof'], v['avg_launch_ms'])"
done
