#!/bin/bash
# Regenerates the measured evidence of a round on the MI355X box (run through gpurun from the repo root):
#   bash tools/refresh_evidence.sh r05_f [pmc,mfma,bench,trace,conv]      (second argument: which parts; default all)
# Writes everything under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.  The counter tables of part `pmc` are
# also installed as profiles/<round>_pmc_{traffic,mfma}.json on the box, so that the bench lines of the same call read them.
set -u
TAG=${1:-r05_x}
PARTS=${2:-pmc,mfma,bench,trace,conv}
want() { case ",$PARTS," in *",$1,"*) return 0;; *) return 1;; esac; }
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

if want pmc; then
# 1. HBM traffic counters: two separate --pmc passes (never combined with trace domains), kernel by kernel
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_$c.log" 2>&1
done
python3 "$ROOT/tools/pmc_table.py" /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "$OUT/pmc_traffic.json" > "$OUT/pmc_table.log" 2>&1
# 1b. matrix-pipe occupancy counters (their own pass), CoR2 and ODA steps
rm -rf /tmp/pmc_mfma /tmp/pmc_mfma_oda
MFMA_CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --pmc $MFMA_CTRS --output-format csv -d /tmp/pmc_mfma -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_mfma.log" 2>&1
python3 "$ROOT/tools/pmc_mfma.py" /tmp/pmc_mfma "$OUT/pmc_mfma.json" > "$OUT/pmc_mfma_table.log" 2>&1

# 1c. the bf16 / 100-region step (configs[4], one rank's share): matrix-pipe counters and HBM traffic of its kernels
rm -rf /tmp/pmc_mfma_bf16
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma_bf16 -- python3 "$ROOT/bench.py" --dtype bf16 --regions 100 --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_mfma_bf16.log" 2>&1
python3 "$ROOT/tools/pmc_mfma.py" /tmp/pmc_mfma_bf16 "$OUT/pmc_mfma_bf16_n100_b128.json" > "$OUT/pmc_mfma_bf16_table.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_bf16_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_bf16_$c -- python3 "$ROOT/bench.py" --dtype bf16 --regions 100 --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_bf16_$c.log" 2>&1
done
python3 "$ROOT/tools/pmc_table.py" /tmp/pmc_bf16_FETCH_SIZE /tmp/pmc_bf16_WRITE_SIZE "$OUT/pmc_traffic_bf16_n100_b128.json" > "$OUT/pmc_table_bf16.log" 2>&1
# 1d. K2 (the ODA attention op): VALU issue counters and HBM traffic of its kernels
rm -rf /tmp/pmc_valu_oda
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_valu_oda -- python3 "$ROOT/bench.py" --model oda-attention --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_valu_oda.log" 2>&1
python3 "$ROOT/tools/pmc_mfma.py" /tmp/pmc_valu_oda "$OUT/pmc_mfma_oda_attention.json" > "$OUT/pmc_valu_oda_table.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_oda_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_oda_$c -- python3 "$ROOT/bench.py" --model oda-attention --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_oda_$c.log" 2>&1
done
python3 "$ROOT/tools/pmc_table.py" /tmp/pmc_oda_FETCH_SIZE /tmp/pmc_oda_WRITE_SIZE "$OUT/pmc_traffic_oda_attention.json" > "$OUT/pmc_table_oda.log" 2>&1
for f in pmc_mfma_bf16_n100_b128 pmc_traffic_bf16_n100_b128 pmc_mfma_oda_attention pmc_traffic_oda_attention; do
  [ -f "$OUT/$f.json" ] && cp "$OUT/$f.json" "$ROOT/profiles/${TAG%%_*}_$f.json"
done
cp "$OUT/pmc_traffic.json" "$ROOT/profiles/${TAG%%_*}_pmc_traffic.json"
cp "$OUT/pmc_mfma.json" "$ROOT/profiles/${TAG%%_*}_pmc_mfma.json"
fi

if want mfma; then
# 1e. the fp32 MFMA engine (bench.py --f32-products mfma; the default runs the split engine): counters of its kernels, merged into
#     the round's tables (their names -- vqa::rt::gemm_*, relation_dgrad_kernel -- do not occur in the default run), its bench
#     line, its kernel trace, its convergence run
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_mfmaeng_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_mfmaeng_$c -- python3 "$ROOT/bench.py" --f32-products mfma --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_mfmaeng_$c.log" 2>&1
done
python3 "$ROOT/tools/pmc_table.py" /tmp/pmc_mfmaeng_FETCH_SIZE /tmp/pmc_mfmaeng_WRITE_SIZE "$OUT/pmc_traffic_mfmaeng.json" > "$OUT/pmc_table_mfmaeng.log" 2>&1
rm -rf /tmp/pmc_mfma_mfmaeng
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma_mfmaeng -- python3 "$ROOT/bench.py" --f32-products mfma --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate --no-sub-records --detail-file /tmp/pmc_detail.json > "$OUT/pmc_mfma_mfmaeng.log" 2>&1
python3 "$ROOT/tools/pmc_mfma.py" /tmp/pmc_mfma_mfmaeng "$OUT/pmc_mfma_mfmaeng.json" > "$OUT/pmc_mfma_mfmaeng_table.log" 2>&1
python3 - "$OUT" "$ROOT/profiles/${TAG%%_*}" <<'PY'
import json, os, sys
out, prof = sys.argv[1], sys.argv[2]
for kind in ("traffic", "mfma"):
    src = os.path.join(out, "pmc_%s_mfmaeng.json" % kind)
    if not os.path.exists(src):
        continue
    rows = {k: v for k, v in json.load(open(src)).items() if "rt::gemm_" in k or "relation_dgrad_kernel" in k or "linear_dw_reduce" in k}
    for dst in (os.path.join(out, "pmc_%s.json" % kind), "%s_pmc_%s.json" % (prof, kind)):
        if os.path.exists(dst):
            table = json.load(open(dst))
            table.update({k: v for k, v in rows.items() if k not in table})
            json.dump(table, open(dst, "w"), indent=1, sort_keys=True)
    print(kind, "fp32 MFMA engine rows merged:", sorted(rows))
PY
cd "$ROOT"
timeout 900 python3 bench.py --detail-file "$OUT/bench_b512_mfma_detail.json" --f32-products mfma --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records > "$OUT/bench_b512_mfma.json" 2> "$OUT/bench_b512_mfma.log"
timeout 900 python3 bench.py --detail-file "$OUT/bench_oda_b512_mfma_detail.json" --model oda --f32-products mfma --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records > "$OUT/bench_oda_b512_mfma.json" 2> "$OUT/bench_oda_b512_mfma.log"
cd /tmp
rm -rf /tmp/kt_mfmaeng
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_mfmaeng -- python3 "$ROOT/bench.py" --f32-products mfma --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --detail-file /tmp/kt_detail.json > "$OUT/kt_mfmaeng.log" 2>&1
f=$(find /tmp/kt_mfmaeng -name "*kernel_stats.csv" | sort | sed -n 1p)
[ -n "$f" ] && cp "$f" "$OUT/bench_b512_mfma_kernel_stats.csv"
python3 "$ROOT/tools/by_grid.py" /tmp/kt_mfmaeng 1 "hand-written kernels, per-dispatch durations by (kernel, grid work-items); rocprofv3 --kernel-trace of bench.py --f32-products mfma --steps 20 --warmup 5 --no-cpu-baseline (the last column is the total over the run)" > "$OUT/vqa_kernels_by_grid_mfma.txt" 2>&1
VQA_F32_PRODUCTS=mfma timeout 600 python3 "$ROOT/tools/convergence.py" --steps 3000 --model cor2 --out "$OUT/convergence_cor2_mfma.json" > "$OUT/convergence_cor2_mfma.log" 2>&1
cd "$ROOT"
fi

if want bench; then
# 2. bench lines
cd "$ROOT"
# (bench.py prints its compact line on stdout and the full record on stderr + --detail-file)
run() { name=$1; shift; timeout 900 python3 bench.py --detail-file "$OUT/${name}_detail.json" "$@" > "$OUT/$name.json" 2> "$OUT/$name.log"; }
run bench_default
run bench_b512 --steps 20 --warmup 5
VQA_HEAD=legacy run bench_b512_legacy_head --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records
VQA_HEAD=grouped run bench_b512_grouped_head --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records
VQA_HEAD=grouped run bench_oda_b512_grouped_head --model oda --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records
run bench_b512_eager --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-rotate --no-sub-records
run bench_b512_copy_inputs --steps 20 --warmup 5 --copy-inputs --no-cpu-baseline --no-sub-records
VQA_GROUPED_ENGINE=mfma run bench_b512_grouped_mfma --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records
VQA_GROUPED_ENGINE=split run bench_b512_grouped_split --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records
VQA_K4_BF16_FORM=fold run bench_bf16_n100_b128_k4fold --no-sub-records --dtype bf16 --regions 100 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline
run bench_b512_pairwise --no-sub-records --steps 20 --warmup 5 --relation-mode 0 --no-cpu-baseline
run bench_f32_n100_b128 --no-sub-records --regions 100 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16_n100_b128 --no-sub-records --dtype bf16 --regions 100 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline
run bench_b512_encoder --no-sub-records --encoder --steps 10 --warmup 5 --no-cpu-baseline
run bench_oda_b512 --no-sub-records --model oda --steps 20 --warmup 5 --no-cpu-baseline
run bench_oda_attention_b512 --model oda-attention --steps 20 --warmup 5
VQA_K4_FORM=engine run bench_b512_k4_engine --no-sub-records --steps 20 --warmup 5 --no-cpu-baseline
# round 6: the step with the relation tensor written to HBM again (K1 -> K5 unfused), the encoder step with its weight gradients on
# the grouped launches, and the headline once more at the end (the pool's boxes differ: before / after on THIS box)
VQA_RELATION_FUSED=0 run bench_b512_relation_unfused --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records
VQA_TN_SPLIT=0 run bench_b512_encoder_grouped_dw --no-sub-records --encoder --steps 10 --warmup 5 --no-cpu-baseline
run bench_b512_again --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records
fi

if want trace; then
# 3. rocprofv3 kernel traces of the same commands (graph replay and kernel by kernel)
cd /tmp
for mode in graph eager; do
  extra=""; [ $mode = eager ] && extra="--no-graph"
  rm -rf /tmp/kt_$mode
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --detail-file /tmp/kt_detail.json $extra > "$OUT/kt_$mode.log" 2>&1
  f=$(find /tmp/kt_$mode -name "*kernel_stats.csv" | sort | sed -n 1p)
  [ -n "$f" ] && cp "$f" "$OUT/bench_b512_${mode}_kernel_stats.csv"
  python3 "$ROOT/tools/by_grid.py" /tmp/kt_$mode 1 "hand-written kernels, per-dispatch durations by (kernel, grid work-items); rocprofv3 --kernel-trace of bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra (the last column is the total over the run)" "$OUT/trace_$mode.json" > "$OUT/vqa_kernels_by_grid_$mode.txt" 2>&1
done
# (the replayed step's per-kernel medians, stamped with their sources: bench.py's `trace_ms`)
[ -f "$OUT/trace_graph.json" ] && cp "$OUT/trace_graph.json" "$ROOT/profiles/${TAG%%_*}_trace.json"
# 4. the other two configurations, kernel by kernel is not needed: the replayed run's stats and per-dispatch durations
for cfg in "oda_b512|--model oda" "bf16_n100_b128|--dtype bf16 --regions 100 --batch 128"; do
  name=${cfg%%|*}; args=${cfg#*|}
  rm -rf /tmp/kt_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -- python3 "$ROOT/bench.py" $args --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --detail-file /tmp/kt_detail.json > "$OUT/kt_$name.log" 2>&1
  f=$(find /tmp/kt_$name -name "*kernel_stats.csv" | sort | sed -n 1p)
  [ -n "$f" ] && cp "$f" "$OUT/bench_${name}_kernel_stats.csv"
  python3 "$ROOT/tools/by_grid.py" /tmp/kt_$name 1 "hand-written kernels, per-dispatch durations by (kernel, grid work-items); rocprofv3 --kernel-trace of bench.py $args --steps 20 --warmup 5 --no-cpu-baseline (the last column is the total over the run)" "$OUT/trace_$name.json" > "$OUT/vqa_kernels_by_grid_$name.txt" 2>&1
  [ -f "$OUT/trace_$name.json" ] && cp "$OUT/trace_$name.json" "$ROOT/profiles/${TAG%%_*}_trace_$name.json"
done
fi
if want conv; then
# 5. does the replayed step still train on this build?  (student vs a fixed teacher, the reference's recipe)
timeout 600 python3 "$ROOT/tools/convergence.py" --steps 3000 --model cor2 --out "$OUT/convergence_cor2.json" > "$OUT/convergence_cor2.log" 2>&1
timeout 600 python3 "$ROOT/tools/convergence.py" --steps 3000 --model oda --out "$OUT/convergence_oda.json" > "$OUT/convergence_oda.log" 2>&1
fi
ls -la "$OUT"
