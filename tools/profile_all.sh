#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   kernel-trace stats for configs 3, 2, 4 and the taptree MMCS, PMC FETCH_SIZE / WRITE_SIZE passes
#   (separate passes, counters only) for configs 3 and 4.  Output under gpurun_out/prof_r02/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in config3 config2 config4; do
  n=3; [ $cfg = config4 ] && n=2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$cfg -o kt -- python3 $R/tools/prof_prove.py $n $cfg > $O/kt_$cfg.log 2>&1 || { echo "kt $cfg failed"; tail -5 $O/kt_$cfg.log; exit 1; }
  echo "kt $cfg done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_taptree -o kt -- python3 $R/tools/prof_taptree.py 16 8 4 16 > $O/kt_taptree.log 2>&1 || { echo "kt taptree failed"; tail -5 $O/kt_taptree.log; exit 1; }
echo "kt taptree done"
for cfg in config3 config4; do
  n=2
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c -d $O/pmc_${cfg}_$c -o pmc -- python3 $R/tools/prof_prove.py $n $cfg > $O/pmc_${cfg}_$c.log 2>&1 || { echo "pmc $cfg $c failed"; tail -5 $O/pmc_${cfg}_$c.log; exit 1; }
    echo "pmc $cfg $c done"
  done
done
# keep only the small CSVs (the traces themselves are large)
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -60
