#!/bin/bash
# A/B on one box: the round-4 Merkle path (TS_LEAF_TREE=0: leaf launch, level launches, tree launch)
# against the leaf-tree kernel (leaf_tree.hpp).  Writes under gpurun_out/r5/.
set -e
O=gpurun_out/r5
mkdir -p $O
TS_LEAF_TREE=0 python tools/latency.py > $O/lat_old.txt 2>&1
python tools/latency.py > $O/lat_new.txt 2>&1
TS_FRI_ROUND_LOG=0 python tools/latency.py > $O/lat_new_frl0.txt 2>&1
TS_LEAF_TREE=0 python bench.py --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err
python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err
TS_LEAF_TREE=0 python bench.py --no-cpu-baseline --workload config2 > $O/bench2_old.json 2> $O/bench2_old.err
python bench.py --no-cpu-baseline --workload config2 > $O/bench2_new.json 2> $O/bench2_new.err
cat $O/lat_*.txt
