#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_x; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HP_BENCH_FORCE_EXCHANGE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 6 --warmup 4 --no-extras --no-cpu-baseline > $O/log.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
grep -o "ms_per_step.: [0-9.]*" $O/log.txt
