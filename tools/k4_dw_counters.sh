#!/bin/bash
# counters of K4's weight-gradient kernel (split engine vs fp32 MFMA form), two small --pmc passes each
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_j; mkdir -p $OUT
for v in 1 0; do
  i=0
  for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/pk4_$v_$i
    VQA_K4_DW_SPLIT=$v timeout 120 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/pk4_${v}_$i -- python3 $GRAFT_REPO_ROOT/tools/k4_dw_once.py 2 > $OUT/once_${v}_$i.log 2>&1
  done
done
python3 - <<'PY' > $OUT/k4_counters.txt
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('/tmp/pk4_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'dw_split' in r['Kernel_Name'] or 'dw_rt' in r['Kernel_Name']:
            rows[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in rows.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-28s %14.0f  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
cat $OUT/k4_counters.txt
