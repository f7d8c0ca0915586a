#!/bin/bash
# GPU box: whole step (no profiler) per library variant, interleaved, REPS rounds
L=3d-point-clouds-autocomplete_amd/hyperpocket_amd/libhyperpocket_hip.so
cp $L /tmp/lib_default.so
run() { echo "$1 $(python bench.py --no-extras --no-cpu-baseline --steps 80 --warmup 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
for rep in $(seq ${1:-3}); do
  cp /tmp/lib_default.so $L; run default
  for v in tools/scratch/libs/*.so; do cp $v $L; run $(basename $v .so); done
done
cp /tmp/lib_default.so $L
