R=$PWD
cd /tmp && export TMPDIR=/tmp
for t in ${TUNES:-0 1 2 3}; do
  export VQA_K4_DW_TUNE=$t
  rm -rf /tmp/prof_t$t
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t$t -o t$t -- python3 $R/tools/kbench.py --only k4bwd --rounds 5 --tiles "" > /tmp/kb_$t.log 2>&1 || tail -5 /tmp/kb_$t.log
  echo "TUNE=$t"
  python3 - <<PY
import csv,glob
for f in glob.glob('/tmp/prof_t$t/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'bilinear_dw' in row['Name'] or 'folded' in row['Name']: print(row['Name'][:60], row['Calls'], row['AverageNs'], row['MinNs'])
PY
done
