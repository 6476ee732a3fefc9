#!/bin/bash
# GPU-side kernel durations of one tools/kbench.py section (event timings of short kernels include host launch time):
#   bash tools/kprof.sh k3 [name-filter]      (run on the GPU box, e.g. through gpurun)
R=$PWD
ONLY=${1:-k3}
FILT=${2:-vqa}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kprof_$ONLY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof_$ONLY -o k -- python3 $R/tools/kbench.py --only $ONLY --rounds 10 --tiles "" > /tmp/kprof_$ONLY.log 2>&1 || tail -5 /tmp/kprof_$ONLY.log
python3 - <<PY
import csv,glob
for f in glob.glob('/tmp/kprof_$ONLY/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if '$FILT' in row['Name']:
            print("%-100s calls %5s  avg %8.1f us  min %8.1f us" % (row['Name'][:100], row['Calls'], float(row['AverageNs'])/1e3, float(row['MinNs'])/1e3))
PY
