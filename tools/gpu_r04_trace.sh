#!/bin/bash
set -e
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_trace_hl -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-overlap --no-kernel-timer > $R/gpurun_out/prof_trace_hl.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_trace_hl/**/*kernel_trace.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'csr_count_kernel' in r['Kernel_Name']]
lo,hi=marks[-2],marks[-1]
out=open('gpurun_out/trace_last_step.txt','w')
t0=int(rows[lo]['Start_Timestamp']); prev=None; busy=0
for r in rows[lo:hi]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=0 if prev is None else max(0,s-prev)
    out.write('%9.1f us  +%7.1f gap  %8.1f us  %s\n'%((s-t0)/1e3,gap/1e3,(e-s)/1e3,r['Kernel_Name'].replace('void ','')[:110]))
    busy+=e-s; prev=max(prev or e,e)
out.write('# step: %d dispatches, span %.2f ms, busy %.2f ms\n'%(hi-lo,(prev-t0)/1e6,busy/1e6))
print(open('gpurun_out/trace_last_step.txt').read()[-200:])
PY
find gpurun_out/prof_trace_hl -name "*kernel_trace.csv" -delete
