#!/bin/bash
B="python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 10"
for c in 2 1; do
  echo "chains=$c plain:        $(HP_EMD_CHAINS=$c $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  echo "chains=$c one-rank rccl: $(HP_EMD_CHAINS=$c HP_BENCH_FORCE_EXCHANGE=1 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  echo "chains=$c q4 plain:      $(GPU_MAX_HW_QUEUES=4 HP_EMD_CHAINS=$c $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  echo "chains=$c q4 one-rank:   $(GPU_MAX_HW_QUEUES=4 HP_EMD_CHAINS=$c HP_BENCH_FORCE_EXCHANGE=1 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('full line: step', d['ms_per_step'], 'chamfer-only', d['breakdown']['chamfer_only_ms_per_step'], 'strict', d['breakdown']['strict_fp32_ms_per_step'], 'one-rank', d['breakdown']['one_rank_rccl_exchange'], 'ref loop', d['reference_loop']['ms_per_step'], d['reference_loop']['with_flat_adam']['ms_per_step'])"
