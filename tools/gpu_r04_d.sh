#!/bin/bash
R=$(pwd); mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_sage_layer_gpu.py tests/test_hub_gpu.py tests/test_graph_gpu.py tests/test_tcga_shape_gpu.py tests/test_models_gpu.py -x -q > gpurun_out/d_tests.log 2>&1 || { tail -30 gpurun_out/d_tests.log; exit 1; }
tail -2 gpurun_out/d_tests.log
python tools/bench_tcga.py --shape kirc | tail -1 | cut -c300-420
python tools/bench_tcga.py --shape gbm | tail -1 | cut -c300-420
