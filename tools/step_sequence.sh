#!/bin/bash
# Kernel sequence of ONE eager training step, in launch order, with GPU durations (what is next to what: fusion candidates):
#   bash tools/step_sequence.sh [bench.py flags]      (run on the GPU box, e.g. through gpurun)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stepseq
rocprofv3 --kernel-trace --output-format csv -d /tmp/stepseq -o s -- python3 $R/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-rotate "$@" > /tmp/stepseq.log 2>&1 || tail -5 /tmp/stepseq.log
python3 - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/stepseq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last step = from the last-but-one adam_kernel (exclusive) to the last one (inclusive)
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]['Start_Timestamp'])
tot = 0
for r in rows[lo:hi]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n)[:90]
    print("%8.1f  %7.1f us  grid %9s  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d, r.get('Grid_Size', r.get('Grid_Size_X', '?')), n))
print("kernels %d, sum %.1f us" % (hi - lo, tot))
PY
