#!/bin/bash
# round-4 GPU call C: fused SAGE layer tests, TCGA-shape runs + kernel stats
set -e
R=$(pwd)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_skinny_linear_gpu.py tests/test_project_gpu.py tests/test_tcga_shape_gpu.py tests/test_workload_gpu.py tests/test_models_gpu.py -x -q > gpurun_out/c8_tests.log 2>&1 || { tail -60 gpurun_out/c8_tests.log; exit 1; }
tail -3 gpurun_out/c8_tests.log
for s in kirc gbm; do
  timeout -k 10 300 python tools/bench_tcga.py --shape $s --json gpurun_out/c8_tcga_$s.json > gpurun_out/c8_tcga_$s.log 2>&1 || { tail -30 gpurun_out/c8_tcga_$s.log; exit 1; }
  tail -1 gpurun_out/c8_tcga_$s.log
  timeout -k 10 300 python tools/bench_tcga.py --shape $s --no-shared-topology > gpurun_out/c8_tcga_${s}_unfused.log 2>&1 || true
  tail -1 gpurun_out/c8_tcga_${s}_unfused.log
done
cd /tmp && export TMPDIR=/tmp
for s in kirc gbm; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tcga9_$s -- python3 $R/tools/bench_tcga.py --shape $s --steps 10 > $R/gpurun_out/prof_tcga9_$s.log 2>&1
  echo "tcga $s stats done"
done
cd $R
find gpurun_out -name "*kernel_trace.csv" -delete
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c8_bench.log 2>&1
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/c8_bench.log') if l.startswith('{')][-1]); print('headline', d['value'], d['ms_per_step'], d['roofline']['stream_copy_variants_GBps'])"
