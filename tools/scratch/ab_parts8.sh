#!/bin/bash
L=3d-point-clouds-autocomplete_amd/hyperpocket_amd/libhyperpocket_hip.so
cp $L /tmp/lib_default.so
B="python bench.py --no-extras --no-cpu-baseline --steps 80 --warmup 10"
emd() { python - <<PY
import bench, torch
torch.cuda.set_device(0)
print("$1 emd", min(bench.roofline_emd(64, 2048)["avg_call_ms"] for _ in range(3)))
PY
}
for rep in 1 2 3; do
  cp /tmp/lib_default.so $L; echo "default $($B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; emd default
  cp tools/scratch/libs/lib_parts8.so $L
  echo "parts8 r442 $(HP_EMD_ROWS1_R=4 HP_EMD_ROWS2_R=4 HP_EMD_GRAD2_R=2 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; HP_EMD_ROWS1_R=4 HP_EMD_ROWS2_R=4 HP_EMD_GRAD2_R=2 emd parts8-442
  echo "parts8 r242 $(HP_EMD_ROWS1_R=2 HP_EMD_ROWS2_R=4 HP_EMD_GRAD2_R=2 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
cp /tmp/lib_default.so $L
