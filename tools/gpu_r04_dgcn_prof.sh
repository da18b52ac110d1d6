#!/bin/bash
# kernel stats of the default-flag DeeperGCN step (development; the committed summary comes from tools/profile_round.sh tcga)
R=$(pwd); mkdir -p $R/gpurun_out; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dgcn_dev -- python3 $R/tools/bench_deepergcn.py "$@" > $R/gpurun_out/prof_dgcn_dev.log 2>&1) || echo FAILED
python3 - <<'PY'
import csv, glob
hits = sorted(glob.glob("gpurun_out/prof_dgcn_dev/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(hits[-1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%8.1f us x %4d  %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:110]))
PY
tail -1 $R/gpurun_out/prof_dgcn_dev.log
