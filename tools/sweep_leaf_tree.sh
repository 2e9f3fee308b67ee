#!/bin/bash
# one-box sweep of the leaf-tree kernel's knobs (round 5): leaves per lane, finisher vs second launch,
# occupancy target (library variants built with TS_LEAF_TREE_WAVES = 4 / 5 / 6)
set -e
O=gpurun_out/r5
mkdir -p $O

{
TS_LEAF_TREE=0 python tools/time_leaf_tree.py 22
for r in 3 2 1 0; do TS_LEAF_TREE_R=$r python tools/time_leaf_tree.py 22; done
TS_LEAF_TREE_FINISH=0 python tools/time_leaf_tree.py 22

TS_LEAF_TREE=0 python tools/time_leaf_tree.py 18 20
python tools/time_leaf_tree.py 18 20
} > $O/sweep1.txt 2>&1
grep -v amdgpu.ids $O/sweep1.txt
