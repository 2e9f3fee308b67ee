#!/bin/bash
# same-box A/B of the headline (config 3, driver protocol, 3 windows): arguments are "NAME VAR=value ..." settings
# separated by ---, each run twice in alternation.   bash tools/ab_headline.sh old TS_LEAF_TREE=0 --- new
O=gpurun_out/r5
mkdir -p $O
settings=(); cur=""
for a in "$@"; do if [ "$a" = "---" ]; then settings+=("$cur"); cur=""; else cur="$cur $a"; fi; done
settings+=("$cur")
for rep in 1 2; do
  for s in "${settings[@]}"; do
    set -- $s; name=$1; shift
    env "$@" python bench.py --headline-only --no-cpu-baseline ${WORKLOAD:+--workload $WORKLOAD} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '$*', d['ms_per_step'], d['extra']['windows_ms_per_step'])"
  done
done | tee $O/ab_headline_$(date +%s).txt
