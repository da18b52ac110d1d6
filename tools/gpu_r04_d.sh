#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
python bench.py --gpus 1 --global-batch 512 --steps 4 --warmup 2 --pool-batches 2 --no-cpu-baseline --no-extras > gpurun_out/d_strong_$i.log 2>&1
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/d_strong_$i.log') if l.startswith('{')][-1]); print('strong', d['value'], d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done
