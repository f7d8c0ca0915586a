#!/bin/bash
# GPU box: phase-skipping runs of conv_pp_kernel (HP_PP_DBG bits: 1 no MFMA, 2 no DMA, 4 no fragment reads, 8 no barriers)
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 4 8 3 5 6 7 15; do echo "== HP_PP_DBG=$d"; HP_PP_DBG=$d ONLY=1 python tools/bench_pp.py 2>&1 | grep "M="; done
