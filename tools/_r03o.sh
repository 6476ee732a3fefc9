mkdir -p gpurun_out/r03o
timeout 900 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -8 > gpurun_out/r03o/bf16_tests.txt
timeout 300 python bench.py --dtype bf16 --regions 100 --batch 128 --no-sub-records --no-cpu-baseline > gpurun_out/r03o/b_bf16.json 2> gpurun_out/r03o/b_bf16.err
MODE=graph bash tools/step_sequence.sh --dtype bf16 --regions 100 --batch 128 > gpurun_out/r03o/seq_bf16.txt 2>&1
