#!/bin/bash
# round-4 GPU call C: fused SAGE layer tests, TCGA-shape runs + kernel stats
set -e
R=$(pwd)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_aggregate_gpu.py tests/test_hub_gpu.py tests/test_tcga_shape_gpu.py tests/test_sage_layer_gpu.py tests/test_models_gpu.py tests/test_graph_gpu.py tests/test_harness_gpu.py -x -q > gpurun_out/c7_tests.log 2>&1 || { tail -60 gpurun_out/c7_tests.log; exit 1; }
tail -3 gpurun_out/c7_tests.log
for s in kirc gbm; do
  timeout -k 10 300 python tools/bench_tcga.py --shape $s --json gpurun_out/c7_tcga_$s.json > gpurun_out/c7_tcga_$s.log 2>&1 || { tail -30 gpurun_out/c7_tcga_$s.log; exit 1; }
  tail -1 gpurun_out/c7_tcga_$s.log
  timeout -k 10 300 python tools/bench_tcga.py --shape $s --no-shared-topology > gpurun_out/c7_tcga_${s}_unfused.log 2>&1 || true
  tail -1 gpurun_out/c7_tcga_${s}_unfused.log
done
cd /tmp && export TMPDIR=/tmp
for s in kirc gbm; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tcga8_$s -- python3 $R/tools/bench_tcga.py --shape $s --steps 10 > $R/gpurun_out/prof_tcga8_$s.log 2>&1
  echo "tcga $s stats done"
done
cd $R
find gpurun_out -name "*kernel_trace.csv" -delete
