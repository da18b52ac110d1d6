#!/bin/bash
# SQ / TCC counters of the max backward's three kernels (csrc/max_sparse.hip) in the default-flag DeeperGCN step, one
# rocprofv3 --pmc pass per counter group (run on the GPU box from the repo root):  tools/pmc_max_sparse.sh [graphs]
R=$(pwd)
G=${1:-16}
OUT=$R/gpurun_out/pmc_max_sparse
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/bench_deepergcn.py --graphs $G --steps 2 > $OUT/$tag.log 2>&1 || echo "FAILED $grp"
done
cd $R
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        for name in ('max_winners', 'max_sparse_bwd', 'max_sparse_table_grad'):
            if name in k:
                res[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name, c in res.items():
    print(name)
    for k, v in sorted(c.items()):
        print("   %-28s %14.0f" % (k, sum(v) / len(v)))
PY
