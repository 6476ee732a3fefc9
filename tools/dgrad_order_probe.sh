#!/bin/bash
# relation_dgrad_split: row-major tile order (an XCD = 16 row tiles x all 8 column tiles) against column-major (an XCD = one
# column tile): time (tools/dgrad_split_ablate.py) and FETCH_SIZE per launch.   bash tools/dgrad_order_probe.sh <outdir>
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/dgrad_order}; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for order in row col; do
  export VQA_SPLIT_DGRAD_ORDER=$order VQA_SPLIT_DGRAD_TUNE=0 VQA_SPLIT_DGRAD_SHARED=1
  python3 "$ROOT/tools/dgrad_split_ablate.py" 0 1 > "$OUT/time_$order.log" 2>&1
  rm -rf /tmp/dg_$order
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/dg_$order -- python3 "$ROOT/tools/dgrad_split_ablate.py" 0 1 > "$OUT/pmc_$order.log" 2>&1
  python3 - /tmp/dg_$order "$order" >> "$OUT/fetch.log" <<'PY'
import csv, glob, sys
vals = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == "FETCH_SIZE" and "relation_dgrad_split_kernel" in row["Kernel_Name"]:
            vals.setdefault(row["Dispatch_Id"], 0.0)
            vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
v = sorted(vals.values())
print("order %s: FETCH_SIZE per launch (KiB units) median %.0f = %.1f MB raw (x2 for 16-byte streams: %.1f MB), %d launches"
      % (sys.argv[2], v[len(v) // 2], v[len(v) // 2] * 1024 / 1e6, v[len(v) // 2] * 2048 / 1e6, len(v)))
PY
done
cat "$OUT"/time_*.log "$OUT/fetch.log"
