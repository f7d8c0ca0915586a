#!/bin/bash
for i in 1 2 3; do
  sleep 5
  a=$(HP_BENCH_PREHEAT_MS=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  sleep 5
  b=$(python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "no preheat $a | preheat 300 ms $b"
done
