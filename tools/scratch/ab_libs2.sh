#!/bin/bash
# GPU box: whole step + one kernel's average time per library variant
L=3d-point-clouds-autocomplete_amd/hyperpocket_amd/libhyperpocket_hip.so
K=${1:-enc_bwd_gather}
cp $L /tmp/lib_default.so
cd /tmp; export TMPDIR=/tmp
run() {
  rm -rf /tmp/p_$1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > /tmp/p_$1.log 2>&1
  ms=$(grep -o '"ms_per_step": [0-9.]*' /tmp/p_$1.log)
  kt=$(grep -h "$K" $(find /tmp/p_$1 -name "*kernel_stats.csv") | head -1 | awk -F, '{print $1, "calls", $2, "avg_ns", $4}')
  echo "$1: $ms (under rocprof) | $kt"
}
for rep in 1 2; do
  cp /tmp/lib_default.so $GRAFT_REPO_ROOT/$L; run default
  for v in $GRAFT_REPO_ROOT/tools/scratch/libs/*.so; do cp $v $GRAFT_REPO_ROOT/$L; run $(basename $v .so); done
done
cp /tmp/lib_default.so $GRAFT_REPO_ROOT/$L
