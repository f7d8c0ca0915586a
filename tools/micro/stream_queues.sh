#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/streamq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/micro/stream_queues.py > $O/a.log 2>&1
export GPU_MAX_HW_QUEUES=8
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/micro/stream_queues.py > $O/b.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
for d in a b; do echo == $d; grep fill $O/$d.log; python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("$O/$d/*/*_kernel_trace.csv")[0])):
    if 'Fill' in r['Kernel_Name'] or 'fill' in r['Kernel_Name']:
        print(r['Queue_Id'], r['Stream_Id'], r['Grid_Size_X'])
PY
done
