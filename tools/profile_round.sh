#!/bin/bash
# Regenerates the files under profiles/ for one round (run on the GPU box from the repo root), in stages that each fit one
# 20-minute GPU call:
#   tools/profile_round.sh <tag> <commit> <stage>      stage = headline | stress | tcga | pmc | all
#
# headline  configs[1] bench: kernel-trace + stats, FETCH_SIZE and WRITE_SIZE in their own --pmc passes
#           -> profiles/<tag>_kernel_stats.md, <tag>_pmc_traffic.json, traffic.json (with provenance; bench.py refuses it once
#           the kernel sources change); the topology build on its own -> <tag>_csr_build.json, <tag>_csr_kernel_stats.md;
#           one GENConv layer's dense kernels -> <tag>_dense_kernels.json; the strong-scaling N = 1 leg of configs[3] (512
#           graphs on one GPU) -> <tag>_strong_n1.json; LAST the un-profiled bench line -> <tag>_bench.json
# stress    configs[4]: 28-layer bf16 (and fp32) stress step -> <tag>_stress_*; DiffPool 4096 / 1024: bench + stats + MFMA
#           counters -> <tag>_diffpool_*; hub rows -> <tag>_skew.json
# tcga      the model the reference ships configs for (MultilevelGNN, gnn_name sage) at kirc / gbm shape through FlatAdam:
#           stats + FETCH / WRITE passes -> <tag>_tcga_{kirc,gbm}_kernel_stats.md, <tag>_tcga_*_pmc_traffic.json, <tag>_tcga.json;
#           DeeperGCN with the reference's default flags -> <tag>_deepergcn_default*
# pmc       SQ / TCC counters of the aggregation kernels at configs[1] (fp32) and configs[4] (bf16) shape
#           -> <tag>_aggregate_pmc.json, <tag>_aggregate_pmc_configs4.json; of the one-pass Linear backward -> <tag>_linear_bwd_pmc.json;
#           of the max backward's kernels in the default-flag DeeperGCN step -> <tag>_max_sparse_pmc.json
# The program always follows `rocprofv3 ... --` directly (no wrapper process).  Raw outputs: gpurun_out/prof_*_<tag>/; a copy
# of what a stage wrote under profiles/ travels back in gpurun_out/profiles_<tag>/.
TAG=${1:-r04}
export MLGNN_COMMIT=${2:-unknown}
STAGE=${3:-all}
R=$(pwd)
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
prof() {   # prof <output dir name> <rocprofv3 options...> -- <program...>
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace "$@" > $R/gpurun_out/$name.log 2>&1) || echo "FAILED: $name"
}
STATS="--stats --output-format csv"
want() { [ "$STAGE" = "all" ] || [ "$STAGE" = "$1" ]; }
jsonline() {   # jsonline <log> <destination>: the last line that is a JSON object
  if grep -q '^{' $1; then grep '^{' $1 | tail -1 > $2; else echo "WARNING: no JSON line in $1"; tail -3 $1; fi
}
jsonblob() {   # jsonblob <log> <destination>: everything from the first '{' (a pretty-printed object)
  python3 -c "import sys; t = open(sys.argv[1]).read(); i = t.find('{'); open(sys.argv[2], 'w').write(t[i:]) if i >= 0 else print('WARNING: no JSON in', sys.argv[1])" $1 $2
}

if want headline; then
  BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
  prof prof_stats_$TAG $STATS -d $R/gpurun_out/prof_stats_$TAG -- $BENCH
  prof prof_fetch_$TAG --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_$TAG -- $BENCH
  prof prof_write_$TAG --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_$TAG -- $BENCH
  prof prof_csr_$TAG $STATS -d $R/gpurun_out/prof_csr_$TAG -- python3 $R/tools/bench_csr.py
  echo "headline passes done"
  python3 tools/summarize_prof.py --stats "gpurun_out/prof_stats_$TAG/**/*kernel_stats.csv" --tag $TAG --commit "$MLGNN_COMMIT" \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" \
    --pmc-cmd "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one pass each) --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" \
    --pmc-fetch "gpurun_out/prof_fetch_$TAG/**/*counter_collection.csv" --pmc-write "gpurun_out/prof_write_$TAG/**/*counter_collection.csv"
  python3 tools/bench_csr.py > gpurun_out/csr_$TAG.log 2>&1
  jsonline gpurun_out/csr_$TAG.log profiles/${TAG}_csr_build.json
  python3 tools/summarize_prof.py --stats "gpurun_out/prof_csr_$TAG/**/*kernel_stats.csv" --tag ${TAG}_csr --commit "$MLGNN_COMMIT" \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_csr.py   (topology build of a configs[1] batch on its own: 64 graphs x 10k nodes / 160k edges, both CSR orderings + the rank-1 edge table)"
  python3 tools/bench_dense.py --iters 30 --json profiles/${TAG}_dense_kernels.json > gpurun_out/dense_$TAG.log 2>&1
  python3 bench.py --gpus 1 --global-batch 512 --steps 4 --warmup 2 --pool-batches 2 --no-cpu-baseline --no-extras > gpurun_out/strong_n1_$TAG.log 2>&1
  jsonline gpurun_out/strong_n1_$TAG.log profiles/${TAG}_strong_n1.json
  # the un-profiled bench line LAST: it picks up the traffic.json written above (same kernel sources -> not stale)
  python3 bench.py > gpurun_out/bench_$TAG.log 2>&1
  jsonline gpurun_out/bench_$TAG.log profiles/${TAG}_bench.json
fi

if want stress; then
  DP="python3 $R/tools/bench_diffpool.py --skip-library --iters 10"
  prof prof_stress_$TAG $STATS -d $R/gpurun_out/prof_stress_$TAG -- python3 $R/tools/stress.py --steps 3 --dtype bf16
  prof prof_stress32_$TAG $STATS -d $R/gpurun_out/prof_stress32_$TAG -- python3 $R/tools/stress.py --steps 3 --dtype fp32
  prof prof_dp_stats_$TAG $STATS -d $R/gpurun_out/prof_dp_stats_$TAG -- $DP
  prof prof_dp_pmc_$TAG --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_dp_pmc_$TAG -- $DP
  prof prof_dp32_stats_$TAG $STATS -d $R/gpurun_out/prof_dp32_stats_$TAG -- $DP --dtype fp32
  echo "stress / diffpool passes done"
  python3 tools/stress.py --steps 3 --dtype bf16 > gpurun_out/stress_$TAG.log 2>&1
  jsonblob gpurun_out/stress_$TAG.log profiles/${TAG}_stress_configs4_bf16.json
  python3 tools/stress.py --steps 3 --dtype fp32 > gpurun_out/stress32_$TAG.log 2>&1
  jsonblob gpurun_out/stress32_$TAG.log profiles/${TAG}_stress_configs4_fp32.json
  python3 tools/bench_diffpool.py --json profiles/${TAG}_diffpool_configs4.json > gpurun_out/dp_$TAG.log 2>&1
  python3 tools/bench_diffpool.py --dtype fp32 --iters 10 --json profiles/${TAG}_diffpool_configs4_fp32.json > gpurun_out/dp32_$TAG.log 2>&1
  python3 tools/bench_skew.py > gpurun_out/skew_$TAG.log 2>&1
  jsonline gpurun_out/skew_$TAG.log profiles/${TAG}_skew.json
  python3 tools/summarize_prof.py --stats "gpurun_out/prof_stress_$TAG/**/*kernel_stats.csv" --tag ${TAG}_stress_bf16 --commit "$MLGNN_COMMIT" \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/stress.py --steps 3 --dtype bf16   (configs[4]: N=200000 E=3000000 d=256, 28 layers, bf16 storage; 4 steps incl. 1 warm-up)"
  python3 tools/summarize_prof.py --stats "gpurun_out/prof_stress32_$TAG/**/*kernel_stats.csv" --tag ${TAG}_stress_fp32 --commit "$MLGNN_COMMIT" \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/stress.py --steps 3 --dtype fp32   (configs[4] shape in fp32 -- NOT the dtype the config names)"
  python3 tools/summarize_prof.py --stats "gpurun_out/prof_dp_stats_$TAG/**/*kernel_stats.csv" --tag ${TAG}_diffpool --commit "$MLGNN_COMMIT" \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_diffpool.py --skip-library --iters 10   (configs[4] DiffPool: 4096 nodes, 1024 clusters, 256 channels, bf16)"
  python3 tools/summarize_prof.py --stats "gpurun_out/prof_dp32_stats_$TAG/**/*kernel_stats.csv" --tag ${TAG}_diffpool_fp32 --commit "$MLGNN_COMMIT" \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_diffpool.py --skip-library --iters 10 --dtype fp32   (configs[4] DiffPool on fp32 tensors: three-term bf16 products)"
  python3 tools/summarize_mfma.py "gpurun_out/prof_dp_pmc_$TAG/**/*counter_collection.csv" "gpurun_out/prof_dp_stats_$TAG/**/*kernel_stats.csv" profiles/${TAG}_diffpool_mfma_pmc.json "$MLGNN_COMMIT"
fi

if want tcga; then
  for shape in kirc gbm; do
    TC="python3 $R/tools/bench_tcga.py --shape $shape --steps 10"
    prof prof_tcga_${shape}_$TAG $STATS -d $R/gpurun_out/prof_tcga_${shape}_$TAG -- $TC
    prof prof_tcga_${shape}_fetch_$TAG --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_tcga_${shape}_fetch_$TAG -- $TC
    prof prof_tcga_${shape}_write_$TAG --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_tcga_${shape}_write_$TAG -- $TC
  done
  prof prof_dgcn_$TAG $STATS -d $R/gpurun_out/prof_dgcn_$TAG -- python3 $R/tools/bench_deepergcn.py
  echo "tcga / deepergcn passes done"
  for shape in kirc gbm; do
    python3 tools/bench_tcga.py --shape $shape --json gpurun_out/tcga_${shape}_$TAG.json > gpurun_out/tcga_${shape}_$TAG.log 2>&1
    python3 tools/bench_tcga.py --shape $shape --no-shared-topology --json gpurun_out/tcga_${shape}_noshare_$TAG.json > gpurun_out/tcga_${shape}_noshare_$TAG.log 2>&1
    python3 tools/summarize_prof.py --stats "gpurun_out/prof_tcga_${shape}_$TAG/**/*kernel_stats.csv" --tag ${TAG}_tcga_${shape} --commit "$MLGNN_COMMIT" --top 40 --no-traffic-json \
      --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_tcga.py --shape $shape --steps 10   (MultilevelGNN, gnn_name sage, config/$shape.yaml shape, synthetic data; 13 steps incl. 3 warm-up; FlatAdam + clip)" \
      --pmc-fetch "gpurun_out/prof_tcga_${shape}_fetch_$TAG/**/*counter_collection.csv" --pmc-write "gpurun_out/prof_tcga_${shape}_write_$TAG/**/*counter_collection.csv"
  done
  python3 tools/summarize_tcga.py $TAG "$MLGNN_COMMIT"
  python3 tools/bench_deepergcn.py > gpurun_out/dgcn_$TAG.log 2>&1
  python3 tools/summarize_prof.py --stats "gpurun_out/prof_dgcn_$TAG/**/*kernel_stats.csv" --tag ${TAG}_deepergcn_default --commit "$MLGNN_COMMIT" \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_deepergcn.py   (DeeperGCN with the reference's DEFAULT flags: gcn_aggr=max, global_edge=onehot, 64 graphs of configs[1] size)"
  python3 - <<PY
import json, re
txt = open("gpurun_out/dgcn_$TAG.log").read()
m = re.search(r"([0-9.]+) ms/step, ([0-9.]+) graphs/s", txt)
if m:
    json.dump({"workload": "DeeperGCN, reference default flags (gcn_aggr=max, global_edge=onehot -> TableEdge, dropout 0.5), 64 graphs x 10000 nodes / 160000 edges, torch.optim.Adam",
               "ms_per_step": float(m.group(1)), "graphs_per_s": float(m.group(2)), "commit": "$MLGNN_COMMIT"}, open("profiles/${TAG}_deepergcn_default.json", "w"), indent=1)
PY
fi

if want pmc; then
  bash tools/pmc_aggregate.sh $TAG "$MLGNN_COMMIT" > gpurun_out/pmc_agg_$TAG.log 2>&1 || echo "aggregate PMC (configs[1]) failed"
  tail -1 gpurun_out/pmc_agg_$TAG.log
  bash tools/pmc_aggregate.sh $TAG "$MLGNN_COMMIT" configs4 > gpurun_out/pmc_agg4_$TAG.log 2>&1 || echo "aggregate PMC (configs[4]) failed"
  tail -1 gpurun_out/pmc_agg4_$TAG.log
  bash tools/pmc_linear_bwd.sh $TAG "$MLGNN_COMMIT" > gpurun_out/pmc_lb_$TAG.log 2>&1 || echo "linear_bwd PMC failed"
  bash tools/pmc_max_sparse.sh 16 $TAG "$MLGNN_COMMIT" > gpurun_out/pmc_ms_$TAG.log 2>&1 || echo "max_sparse PMC failed"
  tail -1 gpurun_out/pmc_ms_$TAG.log
  echo "pmc passes done"
fi

# keep only the CSVs the summaries came from (the traces are large)
find gpurun_out -name "*kernel_trace.csv" -delete
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_* profiles/traffic.json gpurun_out/profiles_$TAG/ 2>/dev/null
ls gpurun_out/profiles_$TAG/ | tr '\n' ' '
