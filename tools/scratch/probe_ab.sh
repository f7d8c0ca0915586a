#!/bin/bash
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1: step', d['ms_per_step'], 'one-rank', d['breakdown']['one_rank_rccl_exchange'].get('ms_per_step'), d['breakdown']['one_rank_rccl_exchange'].get('exposed_comm_ms'))"; }
run chains2
HP_PROBE_RESET_S2=1 run chains2-reset-s2
