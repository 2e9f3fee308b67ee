mkdir -p gpurun_out/r6
for P in 0.5 0.0 0.5 0.0; do
TS_BENCH_PRIME_MIN_S=$P TS_BENCH_LIVE_PMC=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prime_min=$P', d['priming']['s'], d['windows_ms_per_step'], 'leaf_tree avg', d['roofline']['avg_launch_ms'], 'ntt', d['roofline_ntt']['avg_launch_ms'], 'kernel total', d['roofline']['kernel_ms_total_per_proof'], 'lat', d['single_proof_latency_ms'])" >> gpurun_out/r6/leg_state.txt
done
cat gpurun_out/r6/leg_state.txt
