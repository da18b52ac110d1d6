#!/bin/bash
# SQ counters of the two aggregation kernels at the bench shape (64 graphs x 10 000 nodes x 160 000 edges, d = 128, GEN
# softmax, rank-1 edge term), one rocprofv3 --pmc pass per counter group (run on the GPU box from the repo root):
#   tools/pmc_aggregate.sh <tag> [commit]    ->  profiles/<tag>_aggregate_pmc.json
set -e
TAG=${1:-r02}
COMMIT=${2:-unknown}
R=$(pwd)
OUT=$R/gpurun_out/pmc_agg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/one_aggregate.py > $OUT/$tag.log 2>&1 || echo "FAILED $grp"
done
cd $R
python3 - "$OUT" "$TAG" "$COMMIT" <<'PY'
import collections, csv, glob, json, sys
out, tag, commit = sys.argv[1:4]
res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'csr_aggregate' not in k or 'true>' in k.split('(')[0][-8:]:      # skip the (empty) long-row launches
            continue
        res['fwd' if 'fwd' in k else 'bwd'][row['Counter_Name']].append(float(row['Counter_Value']))
for f in glob.glob(out + '/GRBM*/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'csr_aggregate' in k and 'true>' not in k.split('(')[0][-8:]:
            dur['fwd' if 'fwd' in k else 'bwd'].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
blob = {"_source": {"commit": commit, "command": "tools/pmc_aggregate.sh (rocprofv3 --kernel-trace --pmc <group> -- python3 tools/one_aggregate.py, one pass per group)",
                    "shape": "64 graphs x 10000 nodes x 160000 edges, d=128, GEN softmax, rank-1 edge term, fp32",
                    "note": "averages over the launches of one pass; SQ_* cycle counters are in units of 4 cycles per SIMD "
                            "(SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 / 4) = fraction of the launch a SIMD's VALU was issuing)"}}
for name in ('fwd', 'bwd'):
    c = {k: sum(v) / len(v) for k, v in sorted(res[name].items())}
    c['kernel_us_in_pmc_pass'] = sum(dur[name]) / max(len(dur[name]), 1)
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_ACTIVE_INST_VALU' in c:
        c['valu_issue_fraction'] = c['SQ_ACTIVE_INST_VALU'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0 / 4.0)
    if 'TCC_HIT_sum' in c:
        c['l2_hit_rate'] = c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
    blob[name] = c
json.dump(blob, open('profiles/%s_aggregate_pmc.json' % tag, 'w'), indent=1)
print(json.dumps({k: {kk: round(vv, 3) for kk, vv in v.items() if kk in ('valu_issue_fraction', 'l2_hit_rate', 'kernel_us_in_pmc_pass', 'SQ_INSTS_VALU')} for k, v in blob.items() if k != '_source'}))
PY
