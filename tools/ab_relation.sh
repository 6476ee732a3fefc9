#!/bin/bash
# K1 -> K5 fused against the materialising path, alternating, on ONE box:  bash tools/ab_relation.sh [rounds] [steps]
R=${1:-3}; S=${2:-100}
for i in $(seq 1 $R); do
  for f in 1 0; do
    export VQA_RELATION_FUSED=$f
    python3 bench.py --steps $S --warmup 10 --no-cpu-baseline --no-sub-records --detail-file /tmp/ab_detail.json 2>/dev/null > /tmp/ab_line.json
    python3 - $f <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
print("fused=%s  %.1f samples/s  %.4f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
  done
done
