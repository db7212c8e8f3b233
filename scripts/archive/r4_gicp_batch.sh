#!/bin/bash
# GICP through icpgpu_align_batch: host solver (evaluation servers) vs device solver, by thread count
for dev in 0 1; do for th in auto 4 8; do
  if [ $th = auto ]; then unset ICPGPU_BATCH_THREADS; else export ICPGPU_BATCH_THREADS=$th; fi
  echo "== ICPGPU_GICP_DEVICE=$dev threads=$th"; ICPGPU_GICP_DEVICE=$dev python scripts/gicp_batch_probe2.py 2>&1 | grep -v amdgpu
done; done
