mkdir -p gpurun_out/r03p
for cfg in "512 up" "256 down" "512 down" "768 down" "1024 down"; do
  set -- $cfg
  VQA_BF16_TN_ITEMS=$1 VQA_BF16_TN_ROUND=$2 MODE=graph bash tools/step_sequence.sh --dtype bf16 --regions 100 --batch 128 2>&1 | grep "gemm_bf16_tn\|slab_reduce\|kernels" > gpurun_out/r03p/seq_$1_$2.txt
done
