#!/bin/bash
# Runs on the GPU box (gpurun): the numbers and rocprof summaries committed under profiles/ (per round: HP_ROUND, default r06).
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; O="$R/gpurun_out/final"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B=$R/bench.py
python3 $B > $O/bench_n1.json 2> $O/bench_n1.err
python3 $B --no-emd --no-cpu-baseline > $O/bench_chamfer.json 2>/dev/null
python3 $B --batch 32 --no-cpu-baseline --no-extras > $O/bench_b32.json 2>/dev/null   # BASELINE configs[1]: B=32, N=2048, Chamfer+EMD on one GPU
python3 $B --workload chamfer-stress > $O/bench_stress.json 2>/dev/null
python3 $B --roofline-only > $O/roofline_events.json 2>/dev/null
prof() { d=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- python3 $B "$@" > $O/$d.log 2>&1; }
pmc() { d=$1; c=$2; shift; shift; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$d -- python3 $B "$@" > /dev/null 2>&1; }
# (bench.py pre-conditions the model with 400 untimed steps: the traces hold 400 + warm-up + timed steps; refresh_profiles.py divides by that)
STEP="--steps 5 --warmup 2 --no-extras --no-cpu-baseline"
prof step $STEP
prof roof --roofline-minimal
prof roof_emd --roofline-emd-only
prof stress --workload chamfer-stress --steps 5 --warmup 2 --no-extras
# Round 6: the step's traces hold 400 pre-conditioning steps — a PMC pass over them is hundreds of MB of csv.  The counters are
# collected on the dominant family's own command instead (the `roofline` object's call, shipped sweeps only: 31 calls), on the
# conv5 launch and on the Chamfer stress; the PMC view of the step's other kernels is round 5's (profiles/r05_pmc_step_kernels.md:
# those kernels did not change).
export HP_BENCH_EMD_NO_UNCULL=1
for c in FETCH_SIZE WRITE_SIZE; do
  pmc roof_pmc_$c $c --roofline-minimal
  pmc emd_pmc_$c $c --roofline-emd-only
  pmc stress_pmc_$c $c --workload chamfer-stress --steps 3 --warmup 1 --no-extras
done
BUSY="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
pmc roof_pmc_BUSY "$BUSY" --roofline-minimal
pmc emd_pmc_BUSY "$BUSY" --roofline-emd-only
pmc stress_pmc_BUSY "$BUSY" --workload chamfer-stress --steps 3 --warmup 1 --no-extras
unset HP_BENCH_EMD_NO_UNCULL
find $O/step -name "*kernel_trace.csv" -delete
# keep only the small csv summaries
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O; ls $O
