set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3/pmc_c4
mkdir -p $O
bash tools/pmc_sq.sh config4 > $O/sq.log 2>&1 || { tail -20 $O/sq.log; exit 1; }
cp gpurun_out/prof_sq/sq_table.txt $O/sq_table.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c -d $O/pmc_$c -o pmc -- python3 $R/tools/prof_prove.py 2 config4 > $O/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $O/pmc_$c.log; exit 1; }
done
cd $R
find $O -name "*.csv" -size +20M -delete
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W 2 $O/traffic.json config4 | head -12
head -8 $O/sq_table.txt | cut -c1-330
