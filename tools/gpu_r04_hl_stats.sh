#!/bin/bash
# kernel stats of the headline step (development): tools/gpu_r04_hl_stats.sh [substring ...]
R=$(pwd); mkdir -p $R/gpurun_out; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hl_dev -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_hl_dev.log 2>&1) || echo FAILED
python3 - "$@" <<'PY'
import csv, glob, sys
hits = sorted(glob.glob("gpurun_out/prof_hl_dev/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(hits[-1])))
for r in rows:
    if not sys.argv[1:] or any(a in r["Name"] for a in sys.argv[1:]):
        print("%8.1f us x %4d  min %7.1f max %7.1f  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:90]))
PY
find gpurun_out/prof_hl_dev -name "*kernel_trace.csv" -delete
