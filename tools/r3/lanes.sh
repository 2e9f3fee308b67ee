mkdir -p gpurun_out/r3
for S in 3 4 5 6 8; do
  python3 bench.py --steps $((S*6)) --warmup $S --streams $S --headline-only --windows 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lanes', $S, d['ms_per_step'], d['extra']['windows_ms_per_step'], d['extra']['stagger_ms'])"
done
for st in 0.5 0.7; do
  python3 bench.py --steps 36 --warmup 6 --streams 6 --stagger-ms $st --headline-only --windows 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lanes 6 gate', $st, d['ms_per_step'], d['extra']['windows_ms_per_step'])"
done
