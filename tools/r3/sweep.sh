set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_sharded.py -x -q -k "failure or timeout or rccl" > gpurun_out/r3/t_sharded_new.log 2>&1 || { tail -30 gpurun_out/r3/t_sharded_new.log; exit 1; }
tail -3 gpurun_out/r3/t_sharded_new.log
for st in 0 0.3 0.7 1.0 1.5 2.8 -1; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --headline-only --stagger-ms $st > gpurun_out/r3/sw_$st.json 2>> gpurun_out/r3/sw.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r3/sw_$st.json')); print('stagger', '$st', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['extra']['one_proof_alone_ms_before_the_run'])"
done
for S in 3 5 6; do
  python3 bench.py --gpus 1 --steps 30 --warmup 5 --headline-only --streams $S > gpurun_out/r3/sw_S$S.json 2>> gpurun_out/r3/sw.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r3/sw_S$S.json')); print('lanes', '$S', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['extra']['stagger_ms'])"
done
TS_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 8 --warmup 2 --headline-only > gpurun_out/r3/share2.json 2> gpurun_out/r3/share2.err || { tail -30 gpurun_out/r3/share2.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/r3/share2.json')); print('n_gpus', d['n_gpus'], d['ms_per_step']); b=d['sharded_config4']; print({k:v for k,v in b.items() if k!='variants' and k!='shard_stages_ms_per_rank'}); print({k:(v['ms_per_step'],v['all_ranks_same_proof']) for k,v in b.get('variants',{}).items()})"
