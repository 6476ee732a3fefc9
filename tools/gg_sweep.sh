# step-level sweep of the grouped head's part sizing (round 4): ms per CoR2 step at B = 512
for cfg in "VQA_GG_TN=whole" "VQA_GG_TN=whole VQA_GG_SPLIT=0" "VQA_GG_TN=whole VQA_GG_SLAB=0.15" "VQA_GG_TN=same" "VQA_GG_TN=same VQA_GG_SPLIT=0" "VQA_GG_TN=whole VQA_GG_LONE=1.3"; do
  if [ "$cfg" = "auto" ]; then e=""; else e="$cfg"; fi
  r=$(env $e python bench.py --steps 20 --warmup 5 --no-sub-records --no-cpu-baseline --no-rotate --detail-file /tmp/d.json 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  g=$(python -c "
import json
d=json.load(open('/tmp/d.json'))
print(round(sum(e['ms_per_step'] for e in d['roofline_all'] if e['kernel']=='grouped_gemm'),4), round(sum(e['ms_per_step'] for e in d['roofline_all'] if e['kernel']=='grouped_epilogue'),4))")
  echo "$cfg: $r ms/step  (grouped gemm, epilogue ms: $g)"
done
