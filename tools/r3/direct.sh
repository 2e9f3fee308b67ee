set -e
mkdir -p gpurun_out/r3
TS_LDE_MID_DIRECT=1 python3 -m pytest tests/test_gpu_parity.py -x -q -k "test_commit_lde_and_merkle or prove" > gpurun_out/r3/t_direct.log 2>&1 || { tail -30 gpurun_out/r3/t_direct.log; exit 1; }
tail -2 gpurun_out/r3/t_direct.log
for v in base direct base2 direct2; do
  case $v in direct*) export TS_LDE_MID_DIRECT=1;; *) unset TS_LDE_MID_DIRECT;; esac
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/c3_$v.json 2>> gpurun_out/r3/ab.err
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/c3_$v.json')); print('C3 $v', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['single_proof_latency_ms'], d['roofline']['kernel_ms_total_per_proof'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof']):
    if 'mid' in k : print('   ', k, v['ms_per_proof'], v['avg_launch_ms'])"
done
