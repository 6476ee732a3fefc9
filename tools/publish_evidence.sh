#!/bin/bash
# Copies what is to be judged from gpurun_out/<tag>/ (written by tools/refresh_evidence.sh on the GPU box) into profiles/:
#   bash tools/publish_evidence.sh r05_f
# compact bench lines (+ the full records of the two headline runs), rocprofv3 kernel stats, per-dispatch tables, counter tables
# (also as the un-suffixed profiles/<round>_pmc_*.json that bench.py reads), convergence records.
set -eu
TAG=${1:?tag, e.g. r05_f}
SRC=gpurun_out/$TAG
DST=profiles
for f in bench_default bench_b512 bench_b512_mfma; do
  cp "$SRC/$f.json" "$DST/${TAG}_$f.json"
  cp "$SRC/${f}_detail.json" "$DST/${TAG}_${f}_detail.json"
done
for f in bench_b512_legacy_head bench_b512_grouped_head bench_oda_b512 bench_oda_b512_grouped_head bench_oda_attention_b512 \
         bench_bf16_n100_b128 bench_f32_n100_b128 bench_b512_pairwise bench_b512_encoder bench_b512_k4_engine bench_b512_eager bench_oda_b512_mfma \
         bench_b512_copy_inputs bench_bf16_n100_b128_k4fold bench_b512_grouped_mfma bench_b512_grouped_split \
         bench_b512_relation_unfused bench_b512_encoder_grouped_dw bench_b512_again; do
  cp "$SRC/$f.json" "$DST/${TAG}_$f.json"
done
for f in bench_b512_graph_kernel_stats.csv bench_b512_eager_kernel_stats.csv bench_oda_b512_kernel_stats.csv \
         bench_bf16_n100_b128_kernel_stats.csv vqa_kernels_by_grid_graph.txt vqa_kernels_by_grid_eager.txt \
         vqa_kernels_by_grid_oda_b512.txt vqa_kernels_by_grid_bf16_n100_b128.txt pmc_traffic.json pmc_mfma.json \
         pmc_mfma_bf16_n100_b128.json pmc_traffic_bf16_n100_b128.json pmc_mfma_oda_attention.json pmc_traffic_oda_attention.json \
         convergence_cor2.json convergence_oda.json \
         bench_b512_mfma_kernel_stats.csv vqa_kernels_by_grid_mfma.txt convergence_cor2_mfma.json; do
  cp "$SRC/$f" "$DST/${TAG}_$f"
done
for f in trace_graph trace_bf16_n100_b128 trace_oda_b512; do
  [ -f "$SRC/$f.json" ] && cp "$SRC/$f.json" "$DST/${TAG}_$f.json"
done
[ -f "$SRC/trace_graph.json" ] && cp "$SRC/trace_graph.json" "$DST/${TAG%%_*}_trace.json"
[ -f "$SRC/trace_bf16_n100_b128.json" ] && cp "$SRC/trace_bf16_n100_b128.json" "$DST/${TAG%%_*}_trace_bf16_n100_b128.json"
cp "$SRC/pmc_traffic.json" "$DST/${TAG%%_*}_pmc_traffic.json"
cp "$SRC/pmc_mfma.json" "$DST/${TAG%%_*}_pmc_mfma.json"
for f in pmc_mfma_bf16_n100_b128 pmc_traffic_bf16_n100_b128 pmc_mfma_oda_attention pmc_traffic_oda_attention; do
  cp "$SRC/$f.json" "$DST/${TAG%%_*}_$f.json"
done
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("bench_default", "bench_b512", "bench_b512_mfma", "bench_oda_b512_mfma", "bench_oda_b512", "bench_oda_attention_b512", "bench_bf16_n100_b128", "bench_b512_encoder"):
    d = json.load(open("profiles/%s_%s.json" % (tag, f)))
    print("%-28s %10.1f %s  %.3f ms/step  roofline %s frac %s" % (f, d["value"], d["unit"], d["ms_per_step"], d["roofline"].get("kernel"), d["roofline"].get("frac")))
PY
