for cfg in "auto" "VQA_HEAD=grouped" "VQA_HEAD=legacy"; do
  if [ "$cfg" = "auto" ]; then e=""; else e="$cfg"; fi
  r=$(env $e python bench.py --dtype bf16 --regions 100 --batch 128 --steps 50 --warmup 10 --no-sub-records --no-cpu-baseline --no-rotate --detail-file /tmp/d.json 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "bf16 $cfg: $r ms/step"
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfp && rocprofv3 --kernel-trace --output-format csv -d /tmp/bfp -o bf -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --regions 100 --batch 128 --steps 20 --warmup 5 --no-sub-records --no-cpu-baseline --no-rotate --detail-file /tmp/d2.json > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,statistics
f=glob.glob('/tmp/bfp/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
g={}
for r in rows:
    n=r['Kernel_Name'].replace('void ','').split('(')[0]
    g.setdefault(n[:90],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
steps=29  # 5 warm + 20 timed + ~4 eager/extra
tot=0
for n,d in sorted(g.items(), key=lambda kv:-sum(kv[1])):
    if 'spin' in n: continue
    print("%-92s n/step %5.1f  med %7.1f us  us/step %7.1f" % (n, len(d)/steps, statistics.median(d), sum(d)/steps)); tot+=sum(d)/steps
print("total us/step", tot)
PY
