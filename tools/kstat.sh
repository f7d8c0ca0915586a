#!/bin/bash
# usage (GPU box): tools/kstat.sh OUT [bench args...]  -> per-kernel stats of a short bench run (top 25 by time)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; O="$R/gpurun_out/$1"; shift; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline "$@" > $O/step.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
python3 $R/tools/summarize_profile.py $O/step $O/step.md "kstat" 7 > /dev/null; head -32 $O/step.md | cut -c1-150
