#!/bin/bash
# Records the library-GEMM solution table of vqa_playground_pytorch_amd/tuned_gemms.py on the MI355X box:
#   bash tools/tune_library_gemms.sh        (through gpurun; writes gpurun_out/tuned_gemms_gfx950.csv -- copy it into the package)
# Every configuration runs its eager warm-up steps with TunableOp tuning on; new shapes are appended to the one file.
R=$PWD
mkdir -p $R/gpurun_out
export VQA_TUNED_GEMMS=tune VQA_TUNED_GEMMS_FILE=$R/gpurun_out/tuned_gemms_gfx950.csv
rm -f $VQA_TUNED_GEMMS_FILE
run() { timeout 900 python3 $R/bench.py "$@" --steps 5 --warmup 5 --no-cpu-baseline --no-rotate 2>&1 | tail -1 | cut -c1-160; }
run
run --model oda
run --regions 100 --batch 128
run --dtype bf16 --regions 100 --batch 128
run --encoder
for b in 256 1024 2048 4096; do run --batch $b; done
wc -l $VQA_TUNED_GEMMS_FILE
