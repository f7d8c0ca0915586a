#!/bin/bash
for i in 1 2 3 4 5 6; do
  sleep 3
  echo "$(python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
