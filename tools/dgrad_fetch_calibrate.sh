#!/bin/bash
# What does FETCH_SIZE count in relation_dgrad_split?  The guide calibrates the counter (x 1/2) only for 16-byte-per-lane streaming
# reads; this kernel's v loads are 16 bytes per lane but 256 contiguous bytes per ROW (4 rows per instruction).  Ablations of the same
# kernel (VQA_SPLIT_DGRAD_TUNE: 0 = whole kernel, 2 = no v loads, 1 = no main loop -> only v is read) separate v's share.
#   bash tools/dgrad_fetch_calibrate.sh <outdir under the repo>
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/dgrad_fetch}; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
: > "$OUT/fetch_by_tune.log"
for tune in 0 2 1; do
  export VQA_SPLIT_DGRAD_TUNE=$tune VQA_SPLIT_DGRAD_SHARED=1
  rm -rf /tmp/dgc_$tune
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/dgc_$tune -- python3 "$ROOT/tools/dgrad_split_ablate.py" $tune 1 > "$OUT/pmc_tune$tune.log" 2>&1
  python3 - /tmp/dgc_$tune "$tune" >> "$OUT/fetch_by_tune.log" <<'PY'
import csv, glob, sys
vals = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == "FETCH_SIZE" and "relation_dgrad_split_kernel" in row["Kernel_Name"]:
            vals.setdefault(row["Dispatch_Id"], 0.0)
            vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
v = sorted(vals.values())
print("tune %s: FETCH_SIZE per launch median %.0f KiB = %.1f MB as counted, %d launches" % (sys.argv[2], v[len(v) // 2], v[len(v) // 2] * 1024 / 1e6, len(v)))
PY
done
cat "$OUT/fetch_by_tune.log"
