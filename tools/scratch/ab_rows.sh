#!/bin/bash
L=3d-point-clouds-autocomplete_amd/hyperpocket_amd/libhyperpocket_hip.so
cp $L /tmp/lib_default.so
emd() { python - <<PY
import bench, torch
torch.cuda.set_device(0)
print("$1", bench.roofline_emd(64, 2048)["avg_call_ms"])
PY
}
for lib in default parts8; do
  [ $lib = parts8 ] && cp tools/scratch/libs/lib_parts8.so $L
  for r1 in 2 4; do for r2 in 2 4; do for g2 in 1 2; do
    HP_EMD_ROWS1_R=$r1 HP_EMD_ROWS2_R=$r2 HP_EMD_GRAD2_R=$g2 emd "$lib rows1=$r1 rows2=$r2 grad2=$g2"
  done; done; done
done
cp /tmp/lib_default.so $L
