set -e
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-config3}
O=$R/gpurun_out/r3/pmc_$CFG
mkdir -p $O
bash tools/pmc_sq.sh $CFG > $O/sq.log 2>&1 || { tail -20 $O/sq.log; exit 1; }
cp gpurun_out/prof_sq/sq_table.txt $O/sq_table.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/tools/prof_prove.py 2 $CFG > $O/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $O/pmc_$c.log; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/prof_prove.py 3 $CFG > $O/kt.log 2>&1 || { echo "kt failed"; tail -5 $O/kt.log; exit 1; }
cd $R
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W 2 $O/traffic.json $CFG > $O/traffic.txt; head -12 $O/traffic.txt
tail -8 $O/sq_table.txt
