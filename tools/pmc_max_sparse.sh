#!/bin/bash
# SQ / TCC counters of the max backward's three kernels (csrc/max_sparse.hip) in the default-flag DeeperGCN step, one
# rocprofv3 --pmc pass per counter group (run on the GPU box from the repo root):
#   tools/pmc_max_sparse.sh [graphs] [tag] [commit]   ->  profiles/<tag>_max_sparse_pmc.json
#   SUMMARIZE_ONLY=1: only summarise the passes already under gpurun_out/pmc_max_sparse
R=$(pwd)
G=${1:-16}
TAG=${2:-dev}
COMMIT=${3:-unknown}
OUT=$R/gpurun_out/pmc_max_sparse
mkdir -p $OUT
if [ -z "$SUMMARIZE_ONLY" ]; then
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/bench_deepergcn.py --graphs $G --steps 2 > $OUT/$tag.log 2>&1 || echo "FAILED $grp"
done
fi
cd $R
python3 - "$OUT" "$TAG" "$COMMIT" "$G" <<'PY'
import collections, csv, glob, json, sys
out, tag, commit, graphs = sys.argv[1:5]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        for name in ('max_winners', 'max_sparse_bwd', 'max_sparse_table_grad'):
            if name in k:
                res[name][row['Counter_Name']].append(float(row['Counter_Value']))
blob = {"_source": {"commit": commit, "command": "tools/pmc_max_sparse.sh %s (rocprofv3 --kernel-trace --pmc <group> -- python3 "
                                                  "tools/bench_deepergcn.py --graphs %s --steps 2, one pass per group)" % (graphs, graphs),
                    "shape": "%s graphs x 10000 nodes x 160000 edges, d=128, max aggregator, 20000-row edge-type table" % graphs,
                    "note": "averages per launch, summed over the chip; GRBM_GUI_ACTIVE / 8 = launch duration in cycles; "
                            "SQ_LDS_IDX_ACTIVE / 256 CUs / that = fraction of the launch the LDS index unit was busy"}}
for name, c in res.items():
    ent = {k: sum(v) / len(v) for k, v in sorted(c.items())}
    if "GRBM_GUI_ACTIVE" in ent and "SQ_LDS_IDX_ACTIVE" in ent:
        ent["lds_busy_fraction"] = ent["SQ_LDS_IDX_ACTIVE"] / 256.0 / (ent["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_WAIT_INST_LDS" in ent and "SQ_WAVE_CYCLES" in ent:
        ent["wait_lds_fraction_of_wave_cycles"] = ent["SQ_WAIT_INST_LDS"] / ent["SQ_WAVE_CYCLES"]
    if "GRBM_GUI_ACTIVE" in ent and "SQ_ACTIVE_INST_VALU" in ent:
        ent["valu_issue_fraction"] = ent["SQ_ACTIVE_INST_VALU"] / (1024.0 * ent["GRBM_GUI_ACTIVE"] / 8.0 / 4.0)
    blob[name] = ent
    print(name, {k: round(v, 3) for k, v in ent.items() if "fraction" in k})
path = "profiles/%s_max_sparse_pmc.json" % tag
json.dump(blob, open(path, "w"), indent=1)
print("wrote", path)
PY
