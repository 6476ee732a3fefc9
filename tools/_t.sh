mkdir -p gpurun_out/r03x
timeout 900 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -4 > gpurun_out/r03x/bf16_tests.txt
for i in 1 2; do timeout 300 python bench.py --dtype bf16 --regions 100 --batch 128 --no-sub-records --no-cpu-baseline > gpurun_out/r03x/b_bf16_$i.json 2> gpurun_out/r03x/b_bf16.err; done
VQA_BF16_TILE=128x64 timeout 300 python bench.py --dtype bf16 --regions 100 --batch 128 --no-sub-records --no-cpu-baseline > gpurun_out/r03x/b_bf16_off.json 2> gpurun_out/r03x/b_bf16.err
MODE=graph bash tools/step_sequence.sh --dtype bf16 --regions 100 --batch 128 > gpurun_out/r03x/seq_bf16.txt 2>&1
