#!/bin/bash
R=$(pwd); mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_skinny_linear_gpu.py tests/test_tcga_shape_gpu.py -x -q > gpurun_out/d_tests.log 2>&1 || { tail -30 gpurun_out/d_tests.log; exit 1; }
tail -2 gpurun_out/d_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sk -- python3 $R/tools/bench_tcga.py --shape kirc --steps 10 > $R/gpurun_out/prof_sk.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_sk/**/*kernel_stats.csv',recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'skinny' in r['Name']: print('%-80s %7.1f us'%(r['Name'][:80], float(r['AverageNs'])/1e3))
PY
find gpurun_out/prof_sk -name "*kernel_trace.csv" -delete
python tools/bench_tcga.py --shape kirc | tail -1 | cut -c300-420
