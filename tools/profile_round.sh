#!/bin/bash
# Regenerates every file under profiles/ for one round (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag> [commit]
# 1. headline bench (configs[1]): kernel-trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes
#    -> profiles/<tag>_kernel_stats.md, <tag>_pmc_traffic.json, traffic.json (with provenance; bench.py refuses it
#    once the kernel sources change), <tag>_bench.json (the bench line of the un-profiled run)
# 2. configs[4] stress run (bf16): stats -> profiles/<tag>_stress_bf16_kernel_stats.md, <tag>_stress_configs4_bf16.json
# 3. configs[4] DiffPool (4096 nodes / 1024 clusters): bench + stats + MFMA counters
#    -> profiles/<tag>_diffpool_configs4.json (+ _fp32.json), <tag>_diffpool_kernel_stats.md (+ _fp32_), <tag>_diffpool_mfma_pmc.json
# 4. hub-row benchmark -> profiles/<tag>_skew.json
# 5. topology build on its own -> profiles/<tag>_csr_build.json, <tag>_csr_kernel_stats.md
# The program always follows `rocprofv3 ... --` directly (no wrapper process).  Outputs: gpurun_out/prof_*_<tag>/.
set -e
TAG=${1:-r03}
export MLGNN_COMMIT=${2:-unknown}
R=$(pwd)
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
STRESS="python3 $R/tools/stress.py --steps 3 --dtype bf16"
DP="python3 $R/tools/bench_diffpool.py --skip-library --iters 10"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$TAG -- $BENCH > $R/gpurun_out/prof_stats_$TAG.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_$TAG -- $BENCH > $R/gpurun_out/prof_fetch_$TAG.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_$TAG -- $BENCH > $R/gpurun_out/prof_write_$TAG.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stress_$TAG -- $STRESS > $R/gpurun_out/prof_stress_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stress32_$TAG -- python3 $R/tools/stress.py --steps 3 --dtype fp32 > $R/gpurun_out/prof_stress32_$TAG.log 2>&1 || echo "fp32 stress pass failed"
echo "stress pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dp_stats_$TAG -- $DP > $R/gpurun_out/prof_dp_stats_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_dp_pmc_$TAG -- $DP > $R/gpurun_out/prof_dp_pmc_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dp32_stats_$TAG -- $DP --dtype fp32 > $R/gpurun_out/prof_dp32_stats_$TAG.log 2>&1
echo "diffpool passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_csr_$TAG -- python3 $R/tools/bench_csr.py > $R/gpurun_out/prof_csr_$TAG.log 2>&1
echo "csr pass done"
cd $R
# from here on a failing optional step must not lose the summaries of the passes that succeeded
set +e
python3 tools/stress.py --steps 3 --dtype bf16 > gpurun_out/stress_$TAG.log 2>&1
python3 - <<PY
txt = open("gpurun_out/stress_$TAG.log").read()
if "{" in txt:
    open("profiles/${TAG}_stress_configs4_bf16.json", "w").write(txt[txt.index("{"):])
else:
    print("WARNING: no JSON line in gpurun_out/stress_$TAG.log -- profiles/${TAG}_stress_configs4_bf16.json not written")
PY
python3 tools/bench_diffpool.py --json profiles/${TAG}_diffpool_configs4.json > gpurun_out/dp_$TAG.log 2>&1
python3 tools/bench_diffpool.py --dtype fp32 --iters 10 --json profiles/${TAG}_diffpool_configs4_fp32.json > gpurun_out/dp32_$TAG.log 2>&1
python3 tools/bench_dense.py --iters 30 --json profiles/${TAG}_dense_kernels.json > gpurun_out/dense_$TAG.log 2>&1
python3 tools/bench_skew.py > gpurun_out/skew_$TAG.log 2>&1
if grep -q '^{' gpurun_out/skew_$TAG.log; then grep '^{' gpurun_out/skew_$TAG.log | tail -1 > profiles/${TAG}_skew.json
else echo "WARNING: no JSON line in gpurun_out/skew_$TAG.log -- profiles/${TAG}_skew.json not written"; fi
python3 tools/summarize_prof.py --stats "gpurun_out/prof_stats_$TAG/**/*kernel_stats.csv" --tag $TAG --commit "$MLGNN_COMMIT" \
  --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" \
  --pmc-cmd "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one pass each) --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" \
  --pmc-fetch "gpurun_out/prof_fetch_$TAG/**/*counter_collection.csv" --pmc-write "gpurun_out/prof_write_$TAG/**/*counter_collection.csv"
python3 tools/summarize_prof.py --stats "gpurun_out/prof_stress_$TAG/**/*kernel_stats.csv" --tag ${TAG}_stress_bf16 --commit "$MLGNN_COMMIT" \
  --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/stress.py --steps 3 --dtype bf16   (configs[4]: N=200000 E=3000000 d=256, 28 layers, bf16 storage; 4 steps incl. 1 warm-up)"
python3 tools/summarize_prof.py --stats "gpurun_out/prof_stress32_$TAG/**/*kernel_stats.csv" --tag ${TAG}_stress_fp32 --commit "$MLGNN_COMMIT" \
  --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/stress.py --steps 3 --dtype fp32   (configs[4] shape in fp32 -- NOT the dtype the config names: hidden width 512 is past the fp32 tall kernels, those products run on the library)"
python3 tools/stress.py --steps 3 --dtype fp32 > gpurun_out/stress32_$TAG.log 2>&1
python3 - <<PY
txt = open("gpurun_out/stress32_$TAG.log").read()
if "{" in txt:
    open("profiles/${TAG}_stress_configs4_fp32.json", "w").write(txt[txt.index("{"):])
else:
    print("WARNING: no JSON line in gpurun_out/stress32_$TAG.log")
PY
python3 tools/summarize_prof.py --stats "gpurun_out/prof_dp_stats_$TAG/**/*kernel_stats.csv" --tag ${TAG}_diffpool --commit "$MLGNN_COMMIT" \
  --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_diffpool.py --skip-library --iters 10   (configs[4] DiffPool: 4096 nodes, 1024 clusters, 256 channels, bf16)"
python3 tools/summarize_prof.py --stats "gpurun_out/prof_dp32_stats_$TAG/**/*kernel_stats.csv" --tag ${TAG}_diffpool_fp32 --commit "$MLGNN_COMMIT" \
  --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_diffpool.py --skip-library --iters 10 --dtype fp32   (configs[4] DiffPool on fp32 tensors: three-term bf16 products, mlgnn_diffpool_large_f32_fwd / _bwd)"
python3 tools/bench_csr.py > gpurun_out/csr_$TAG.log 2>&1
if grep -q '^{' gpurun_out/csr_$TAG.log; then grep '^{' gpurun_out/csr_$TAG.log | tail -1 > profiles/${TAG}_csr_build.json; fi
python3 tools/summarize_prof.py --stats "gpurun_out/prof_csr_$TAG/**/*kernel_stats.csv" --tag ${TAG}_csr --commit "$MLGNN_COMMIT" \
  --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_csr.py   (topology build of a configs[1] batch on its own: 64 graphs x 10k nodes / 160k edges, both CSR orderings + the rank-1 edge table)"
python3 tools/summarize_mfma.py "gpurun_out/prof_dp_pmc_$TAG/**/*counter_collection.csv" "gpurun_out/prof_dp_stats_$TAG/**/*kernel_stats.csv" profiles/${TAG}_diffpool_mfma_pmc.json "$MLGNN_COMMIT"
# the un-profiled bench line LAST: it picks up the traffic.json written above (same kernel sources -> not stale)
python3 bench.py > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log > profiles/${TAG}_bench.json
# keep only the CSVs the summaries came from (the traces are large)
find gpurun_out -name "*kernel_trace.csv" -delete
# gpurun brings back gpurun_out/ only: a copy of what this run wrote under profiles/ travels in it
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_* profiles/traffic.json gpurun_out/profiles_$TAG/
ls -la gpurun_out/profiles_$TAG/
