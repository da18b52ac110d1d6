#!/bin/bash
# round-4 GPU call A: >4 GiB dense tests, strong-scaling N=1 leg, headline bench, TCGA-shape kernel stats
set -e
R=$(pwd)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dense_past_4gib_gpu.py tests/test_linear_bwd_gpu.py tests/test_wgrad_gpu.py -x -q > gpurun_out/a_tests.log 2>&1 || { tail -30 gpurun_out/a_tests.log; exit 1; }
tail -3 gpurun_out/a_tests.log
timeout -k 10 600 python bench.py --gpus 1 --global-batch 512 --steps 3 --warmup 1 --pool-batches 2 --no-cpu-baseline --no-extras > gpurun_out/a_strong_n1.log 2>&1 || { tail -30 gpurun_out/a_strong_n1.log; exit 1; }
tail -c 1500 gpurun_out/a_strong_n1.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/a_bench.log 2>&1
tail -c 600 gpurun_out/a_bench.log
for s in kirc gbm; do
  timeout -k 10 300 python tools/bench_tcga.py --shape $s --json gpurun_out/a_tcga_$s.json > gpurun_out/a_tcga_$s.log 2>&1 || { tail -30 gpurun_out/a_tcga_$s.log; exit 1; }
  tail -1 gpurun_out/a_tcga_$s.log
  timeout -k 10 300 python tools/bench_tcga.py --shape $s --torch-adam > gpurun_out/a_tcga_${s}_torchadam.log 2>&1 || true
  tail -1 gpurun_out/a_tcga_${s}_torchadam.log
done
cd /tmp && export TMPDIR=/tmp
for s in kirc gbm; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tcga_$s -- python3 $R/tools/bench_tcga.py --shape $s --steps 10 > $R/gpurun_out/prof_tcga_$s.log 2>&1
  echo "tcga $s stats done"
done
cd $R
find gpurun_out -name "*kernel_trace.csv" -delete
