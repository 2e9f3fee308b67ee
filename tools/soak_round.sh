#!/bin/bash
# Round soak (beyond the suite): concurrent lanes on the headline shape, seeded shape sweeps against the
# oracle, the sharded prover's option sweep and rendezvous soak, the suite with the hipGraph knob on.
#     bash tools/soak_round.sh  ->  gpurun_out/r5/soak.txt
O=gpurun_out/r5
mkdir -p $O
{
python tools/soak.py 40 4 2>&1 | grep -v amdgpu.ids
python tools/soak_sharded.py 200 2 15 2>&1 | grep -v amdgpu.ids
python tools/soak_sharded.py 200 8 12 2>&1 | grep -v amdgpu.ids
echo "TS_RANDOM_SHARDED=300 TS_RANDOM_SHAPES=300 TS_RANDOM_SEED=50505 python -m pytest tests/test_gpu_random_shapes.py:"
TS_RANDOM_SHARDED=300 TS_RANDOM_SHAPES=300 TS_RANDOM_SEED=50505 python -m pytest tests/test_gpu_random_shapes.py -x -q 2>&1 | tail -2
echo "TS_FRI_GRAPH=1 python -m pytest tests -m gpu (minus the graph test):"
TS_FRI_GRAPH=1 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_graph.py 2>&1 | tail -2
} > $O/soak.txt 2>&1
cat $O/soak.txt
