set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py -x -q -k "test_commit_lde_and_merkle or 2p22" > gpurun_out/r3/t_padx.log 2>&1 || { tail -30 gpurun_out/r3/t_padx.log; exit 1; }
tail -1 gpurun_out/r3/t_padx.log
for i in 1 2; do
python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > gpurun_out/r3/padx_c4.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/padx_c4.json')); print('C4', d['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:2]: print('   ', k, v['ms_per_proof'])
for k,v in d['kernels'].items():
    if 'intt' in k: print('   ', k, v['ms_per_proof'])"
done
