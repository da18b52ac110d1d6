#!/bin/bash
# SQ counters of the two aggregation kernels at the bench shape (64 graphs x 10 000 nodes x 160 000 edges, d = 128, GEN
# softmax, rank-1 edge term), one rocprofv3 --pmc pass per counter group (run on the GPU box from the repo root):
#   tools/pmc_aggregate.sh <tag> [commit]    ->  profiles/<tag>_aggregate_pmc.json
set -e
TAG=${1:-r02}
COMMIT=${2:-unknown}
# third argument "configs4": the bf16 stress shape (one graph, 200 000 nodes / 3 M edges, d = 256) -> profiles/<tag>_aggregate_pmc_configs4.json
export SHAPE=${3:-configs1}
SUF=""; [ "$SHAPE" = "configs4" ] && SUF="_configs4"
R=$(pwd)
OUT=$R/gpurun_out/pmc_agg_$TAG$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/one_aggregate.py > $OUT/$tag.log 2>&1 || echo "FAILED $grp"
done
cd $R
python3 - "$OUT" "$TAG" "$COMMIT" "$SUF" <<'PY'
import collections, csv, glob, json, os, sys
out, tag, commit, suf = sys.argv[1:5]
def main_launch(k):
    """csr_aggregate_{fwd,bwd}_kernel<T, VEC, MODE, AGGR, flag, VIRT, ...>: the main launch (VIRT = false), not the
    (here empty) long-row launch"""
    if 'csr_aggregate' not in k or '<' not in k:
        return False
    args = [a.strip() for a in k[k.index('<') + 1:k.index('>(') if '>(' in k else k.rindex('>')].split(',')]
    return len(args) > 5 and args[5] == 'false'
res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if not main_launch(k):
            continue
        res['fwd' if 'fwd' in k else 'bwd'][row['Counter_Name']].append(float(row['Counter_Value']))
for f in glob.glob(out + '/GRBM*/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if main_launch(k):
            dur['fwd' if 'fwd' in k else 'bwd'].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
blob = {"_source": {"commit": commit, "command": "tools/pmc_aggregate.sh (rocprofv3 --kernel-trace --pmc <group> -- python3 tools/one_aggregate.py, one pass per group)",
                    "shape": ("1 graph x 200000 nodes x 3000000 edges, d=256, GEN softmax, rank-1 edge term, bf16 storage (BASELINE configs[4])"
                              if suf else "64 graphs x 10000 nodes x 160000 edges, d=128, GEN softmax, rank-1 edge term, fp32"),
                    "note": "averages over the launches of one pass; SQ_* cycle counters are in units of 4 cycles per SIMD "
                            "(SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 / 4) = fraction of the launch a SIMD's VALU was issuing)"}}
for name in ('fwd', 'bwd'):
    c = {k: sum(v) / len(v) for k, v in sorted(res[name].items())}
    c['kernel_us_in_pmc_pass'] = sum(dur[name]) / max(len(dur[name]), 1)
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_ACTIVE_INST_VALU' in c:
        c['valu_issue_fraction'] = c['SQ_ACTIVE_INST_VALU'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0 / 4.0)
    if 'TCC_HIT_sum' in c:
        c['l2_hit_rate'] = c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
    if 'SQ_WAIT_INST_ANY' in c and 'SQ_WAVE_CYCLES' in c:
        c['wait_inst_fraction_of_wave_cycles'] = c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']
    blob[name] = c
import hashlib
sys.path.insert(0, 'multilevel-gnn_amd')
import build_native
blob['_source']['kernel_sources_sha256'] = build_native.sources_digest()
json.dump(blob, open('profiles/%s_aggregate_pmc%s.json' % (tag, suf), 'w'), indent=1)
print(json.dumps({k: {kk: round(vv, 3) for kk, vv in v.items() if kk in ('valu_issue_fraction', 'l2_hit_rate', 'kernel_us_in_pmc_pass', 'SQ_INSTS_VALU')} for k, v in blob.items() if k != '_source'}))
PY
