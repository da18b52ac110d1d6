#!/bin/bash
# Collects the three rocprofv3 passes the profiles/ summaries are built from (run on the GPU box from the
# repo root):  kernel-trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes.
# usage: tools/profile_round.sh <tag>      outputs: gpurun_out/prof_{stats,fetch,write}_<tag>/
set -e
TAG=${1:-r01}
R=$(pwd)
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$TAG -- $CMD > $R/gpurun_out/prof_stats_$TAG.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_$TAG -- $CMD > $R/gpurun_out/prof_fetch_$TAG.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_$TAG -- $CMD > $R/gpurun_out/prof_write_$TAG.log 2>&1
echo "write pass done"
cd $R
python3 tools/summarize_prof.py --stats "gpurun_out/prof_stats_$TAG/**/*kernel_stats.csv" --tag $TAG \
  --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" \
  --pmc-fetch "gpurun_out/prof_fetch_$TAG/**/*counter_collection.csv" --pmc-write "gpurun_out/prof_write_$TAG/**/*counter_collection.csv"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_kernel_stats.md profiles/${TAG}_pmc_traffic.json profiles/traffic.json gpurun_out/profiles_$TAG/
# keep only the CSVs the summaries came from (the traces are large)
find gpurun_out/prof_stats_$TAG gpurun_out/prof_fetch_$TAG gpurun_out/prof_write_$TAG -name "*kernel_trace.csv" -delete
