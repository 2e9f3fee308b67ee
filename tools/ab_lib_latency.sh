#!/bin/bash
# same-box A/B of two builds of the library (TS_LIB_PATH): single-proof latency (tools/latency.py, configs 3
# and 2), processes alternating, then the headline.   bash tools/ab_lib_latency.sh tap-stark_amd/lib_prev/libtapstark_hip.so
PREV=${1:-tap-stark_amd/lib_prev/libtapstark_hip.so}
O=gpurun_out/r5
mkdir -p $O
L=$O/ab_lib_latency.txt
: > $L
for rep in 1 2 3; do
  echo "prev:" >> $L; TS_LIB_PATH=$PREV timeout -k 10 200 python tools/latency.py 2>/dev/null | cut -c1-150 >> $L || exit 1
  echo "new:" >> $L; timeout -k 10 200 python tools/latency.py 2>/dev/null | cut -c1-150 >> $L || exit 1
done
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export TS_LIB_PATH=$PREV; else unset TS_LIB_PATH; fi
    timeout -k 10 200 python bench.py --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which headline', d['ms_per_step'], d['extra']['windows_ms_per_step'])" >> $L || exit 1
  done
done
cat $L
