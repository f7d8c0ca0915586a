# A/B of the EMD phase kernels' rows-per-lane (env-switched): whole step, two readings each
for cfg in "1 1" "2 4" "4 4" "4 2" "2 4" "4 4"; do set -- $cfg; export HP_EMD_ROWS1_R=$1 HP_EMD_ROWS2_R=$2; echo "R1=$1 R2=$2: $(python tools/ab_step.py 2>&1 | grep 'emd=0.05:' | tail -2 | tr '\n' ' ')"; done
export HP_EMD_ROWS1_R=4 HP_EMD_ROWS2_R=4; python -m pytest tests/test_structural_losses_gpu.py -x -q 2>&1 | tail -2
