#!/bin/bash
# Re-makes detection_3d_amd/tuned/gemm_gfx950.csv on an MI355X box: one bench pass with TunableOp's tuning on records
# the best hipBLASLt / rocBLAS solution for every library GEMM shape of the detector tail (4c bs=1 and the bs=4 region).
#   /usr/local/graft/bin/gpurun -- 'bash scripts/tune_gemms.sh'   ->  gpurun_out/tunableop0.csv ; copy it over the table
set -e
mkdir -p gpurun_out
D3D_TUNED_GEMMS=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop.csv \
  python bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline
cat gpurun_out/tunableop0.csv
