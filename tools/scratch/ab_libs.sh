#!/bin/bash
# GPU box: EMD call time + whole step per library variant (tools/scratch/libs/*.so copied over the package's library)
L=3d-point-clouds-autocomplete_amd/hyperpocket_amd/libhyperpocket_hip.so
cp $L /tmp/lib_default.so
run() {
  python - <<PY
import bench, json, torch
torch.cuda.set_device(0)
r = bench.roofline_emd(64, 2048)
print("$1 emd call ms", r["avg_call_ms"])
PY
  python bench.py --no-extras --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
}
for rep in 1 2; do
  cp /tmp/lib_default.so $L; run default
  for v in tools/scratch/libs/*.so; do cp $v $L; run $(basename $v); done
done
cp /tmp/lib_default.so $L
