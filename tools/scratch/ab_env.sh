#!/bin/bash
# GPU box: A/B of an environment switch on the whole step, interleaved runs of `python bench.py --no-extras --no-cpu-baseline --steps 60`
# usage: bash tools/ab_env.sh VAR=VALUE [rounds]
KV=$1; N=${2:-3}
for i in $(seq $N); do
  a=$(python bench.py --no-extras --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  b=$(env $KV python bench.py --no-extras --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "default $a | $KV $b"
done
