#!/bin/bash
# Kernel sequence of ONE training step, in launch order, with GPU durations and the idle gap in front of each kernel (what is
# next to what: fusion candidates; where the stream waits):
#   [MODE=graph] bash tools/step_sequence.sh [bench.py flags]      (run on the GPU box, e.g. through gpurun)
# MODE=graph traces the hipGraph-replayed step (default: every kernel launched from the host).
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stepseq
rocprofv3 --kernel-trace --output-format csv -d /tmp/stepseq -o s -- python3 $R/bench.py --steps 4 --warmup 3 $([ "${MODE:-eager}" = graph ] || echo --no-graph) --no-cpu-baseline --no-rotate --no-sub-records "$@" > /tmp/stepseq.log 2>&1 || tail -5 /tmp/stepseq.log
python3 - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/stepseq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last step = from the last-but-one adam_kernel (exclusive) to the last one (inclusive)
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]['Start_Timestamp'])
tot = gaps = 0
prev_end = int(rows[lo - 1]['End_Timestamp'])
for r in rows[lo:hi]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev_end) / 1e3
    prev_end = max(prev_end, int(r['End_Timestamp']))
    tot += d
    gaps += max(gap, 0.0)
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n)[:90]
    print("%8.1f  gap %6.1f  %7.1f us  grid %9s  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, gap, d, r.get('Grid_Size', r.get('Grid_Size_X', '?')), n))
print("kernels %d, sum %.1f us, idle gaps %.1f us" % (hi - lo, tot, gaps))
PY
