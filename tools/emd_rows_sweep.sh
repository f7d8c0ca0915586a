# A/B of the EMD kernels' rows-per-lane (env-switched): whole step, two readings each
for g in 1 2 1 2; do export HP_EMD_GRAD2_R=$g; echo "GRAD2_R=$g: $(python tools/ab_step.py 2>&1 | grep 'emd=0.05:' | tail -2 | tr '\n' ' ')"; done
export HP_EMD_GRAD2_R=2; python -m pytest tests/test_structural_losses_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
