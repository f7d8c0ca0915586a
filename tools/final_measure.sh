#!/bin/bash
# Runs on the GPU box (gpurun): the numbers and rocprof summaries committed under profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 $R/bench.py --no-emd --no-cpu-baseline > $O/bench_chamfer.json 2>/dev/null
python3 $R/bench.py --roofline-only > $O/roofline_events.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/step.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python3 $R/bench.py --roofline-only > $O/roof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/roof_pmc_$c -- python3 $R/bench.py --roofline-only > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/step_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/roof_pmc_BUSY -- python3 $R/bench.py --roofline-only > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/step_pmc_BUSY -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
# keep only the small csv summaries
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O; ls $O
