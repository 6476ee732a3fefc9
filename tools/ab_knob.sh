#!/bin/bash
# One library knob, two values, alternating, on ONE box:  bash tools/ab_knob.sh VQA_K2_WSTAGE "1 0" [rounds] [steps] [bench args...]
# (boxes of the pool differ by +-1.5 %: only runs of one call compare.)  Prints the line's rate and the dominant op's time.
K=${1:?knob}; VALS=${2:?values}; R=${3:-3}; S=${4:-100}; shift 4 2>/dev/null || shift $#
for i in $(seq 1 $R); do
  for v in $VALS; do
    export $K=$v
    python3 bench.py --steps $S --warmup 10 --no-cpu-baseline --no-sub-records --detail-file /tmp/ab_detail.json "$@" 2>/dev/null > /tmp/ab_line.json
    python3 - "$K=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d.get("roofline") or {}
print("%s  %.1f %s  %.4f ms/step   [%s mean %s ms]" % (sys.argv[1], d["value"], d["unit"], d["ms_per_step"], r.get("kernel"), r.get("mean_ms")))
PY
  done
done
