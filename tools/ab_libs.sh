#!/bin/bash
# Two builds of libvqa_mi355x.so against each other on ONE box, alternating (the pool's boxes differ by several per cent, and one
# box drifts with its temperature): the in-tree library and another one (VQA_LIB_PATH), the same bench command for both.
#   bash tools/ab_libs.sh path/to/other/libvqa_mi355x.so [rounds] [bench.py arguments ...]
# To get the other build: compile the old version of the one changed source into an object of its own and link it with the rest of
# csrc/build/*.o (the library is one object per source file).  It must lie inside the repo to travel with `gpurun`.
OTHER=$(readlink -f "$1"); shift; R=${1:-3}; shift
ARGS=${*:---steps 100 --warmup 12}
for i in $(seq 1 "$R"); do
  for lib in tree other; do
    if [ $lib = other ]; then export VQA_LIB_PATH=$OTHER; else unset VQA_LIB_PATH; fi
    python3 bench.py $ARGS --no-cpu-baseline --no-sub-records --detail-file /tmp/ab_detail.json 2>/dev/null > /tmp/ab_line.json
    python3 -c "
import json; d = json.load(open('/tmp/ab_line.json')); print('%-5s %10.1f samples/s  %.4f ms/step' % ('$lib', d['value'], d['ms_per_step']))"
  done
done
