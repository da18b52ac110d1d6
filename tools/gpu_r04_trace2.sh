#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_trace_kirc -- python3 $R/tools/bench_tcga.py --shape kirc --steps 4 > $R/gpurun_out/prof_trace_kirc.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_trace_kirc/**/*kernel_trace.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'node_embed_fwd' in r['Kernel_Name']]
lo,hi=marks[-2],marks[-1]
out=open('gpurun_out/trace_kirc_step.txt','w')
t0=int(rows[lo]['Start_Timestamp']); prev=None
for r in rows[lo:hi]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=0 if prev is None else max(0,s-prev)
    out.write('%9.1f us  +%6.1f gap  %7.1f us  %s\n'%((s-t0)/1e3,gap/1e3,(e-s)/1e3,r['Kernel_Name'].replace('void ','')[:120]))
    prev=max(prev or e,e)
out.write('# step: %d dispatches, span %.2f ms\n'%(hi-lo,(prev-t0)/1e6))
PY
find gpurun_out/prof_trace_kirc -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/trace_kirc_step.txt
