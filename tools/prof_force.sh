#!/bin/bash
# GPU box: kernel trace of the step with the exchange forced on in a one-rank RCCL group (tools/step_timeline.py reads it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_force; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HP_BENCH_FORCE_EXCHANGE=1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 4 --warmup 3 --no-extras --no-cpu-baseline > $O/log.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
tail -2 $O/log.txt; du -sh $O
