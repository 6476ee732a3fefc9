#!/bin/bash
# Issue / wait counters per kernel of a bench.py run, kernel by kernel:  bash tools/wait_counters.sh [bench args...]
# (e.g. --model oda-attention).  Two --pmc passes (no trace domains); prints, per kernel, launches-averaged counters and
#   valu% = 100 SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / waves-per-SIMD) is NOT computed here -- the raw ratios are:
#   wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of a wave's life inside s_waitcnt), wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
P=0
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS"; do
  P=$((P+1)); rm -rf /tmp/wc_$P
  timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/wc_$P -- python3 "$ROOT/bench.py" "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/wc_detail.json > /tmp/wc_$P.log 2>&1
  python3 - /tmp/wc_$P <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vqa::" not in k: continue
        k = k.replace("void ", "")
        if k.endswith(")"):          # drop the argument list: the parenthesis that matches the last one (names hold "(anonymous namespace)")
            depth = 0
            for pos in range(len(k) - 1, -1, -1):
                depth += (k[pos] == ")") - (k[pos] == "(")
                if depth == 0:
                    k = k[:pos]
                    break
        k = k + "|" + r.get("Grid_Size", "")
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
out = []
for k, c in rows.items():
    v = {a: x / max(n[(k, a)], 1) for a, x in c.items()}
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    extra = ""
    if "SQ_WAIT_ANY" in v:
        extra = "wait_any %.2f wait_inst %.2f valu_active/wave_cycles %.3f" % (v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_VALU"] / wc)
    else:
        extra = "gui %.0f valu %.0f mfma %.0f mfma_busy %.0f" % (v.get("GRBM_GUI_ACTIVE", 0), v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_MFMA", 0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
    out.append((wc, "%-86s wave_cycles %11.0f  %s" % (k[:86], wc, extra)))
for _, line in sorted(out, reverse=True)[:40]:
    print(line)
PY
done
