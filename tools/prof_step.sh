#!/bin/bash
# GPU box: kernel trace of the plain N=1 step (tools/step_timeline.py reads it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_now; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 6 --warmup 4 --no-extras --no-cpu-baseline > $O/log.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
grep -o "ms_per_step.: [0-9.]*" $O/log.txt
