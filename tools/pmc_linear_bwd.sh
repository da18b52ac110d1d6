#!/bin/bash
# SQ counters of the one-pass Linear backward kernels (csrc/linear_bwd.hip) at the bench shape (640 000 rows, 128 <-> 256),
# one rocprofv3 --pmc pass per counter group (run on the GPU box from the repo root):
#   tools/pmc_linear_bwd.sh <tag> [commit]    ->  profiles/<tag>_linear_bwd_pmc.json
set -e
TAG=${1:-r03}
COMMIT=${2:-unknown}
R=$(pwd)
OUT=$R/gpurun_out/pmc_lb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/bench_dense.py --iters 3 > $OUT/$tag.log 2>&1 || echo "FAILED $grp"
done
cd $R
python3 - "$OUT" "$TAG" "$COMMIT" <<'PY'
import collections, csv, glob, json, sys
out, tag, commit = sys.argv[1:4]
names = {"linear_bwd_kernel<128, 256, 0>": "linear_bwd<LN>", "linear_bwd_kernel<256, 128, 2>": "linear_bwd<SHIFT|plain>",
         "tallgemm_kernel<8, 8, 3": "tallgemm<LN-bwd>", "linear_wgrad_kernel<8, 2, 4": "linear_wgrad (go^T act)"}
res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def key(k):
    for pat, short in names.items():
        if pat in k:
            return short
    return None
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        s = key(row['Kernel_Name'])
        if s:
            res[s][row['Counter_Name']].append(float(row['Counter_Value']))
for f in glob.glob(out + '/GRBM*/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        s = key(row['Kernel_Name'])
        if s:
            dur[s].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
blob = {"_source": {"commit": commit, "command": "tools/pmc_linear_bwd.sh (rocprofv3 --kernel-trace --pmc <group> -- python3 tools/bench_dense.py --iters 3, one pass per group)",
                    "shape": "640000 rows, Linear 128 <-> 256, fp32 (BASELINE configs[1], one GENConv layer)",
                    "note": "averages over the launches of one pass; SQ_* cycle counters are in units of 4 cycles per SIMD; "
                            "mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); "
                            "valu_issue_fraction = SQ_ACTIVE_INST_VALU / (1024 x GRBM_GUI_ACTIVE / 8 / 4)"}}
for name, cs in res.items():
    c = {k: sum(v) / len(v) for k, v in sorted(cs.items())}
    c['kernel_us_in_pmc_pass'] = sum(dur[name]) / max(len(dur[name]), 1)
    g = c.get('GRBM_GUI_ACTIVE')
    if g and 'SQ_ACTIVE_INST_VALU' in c:
        c['valu_issue_fraction'] = c['SQ_ACTIVE_INST_VALU'] / (1024.0 * g / 8.0 / 4.0)
    if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        c['mfma_busy_fraction'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * g / 8.0)
    if 'SQ_LDS_IDX_ACTIVE' in c and c['SQ_LDS_IDX_ACTIVE']:
        c['lds_conflict_fraction'] = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']
    if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES']:
        c['wait_any_fraction'] = c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES'] if 'SQ_WAIT_ANY' in c else None
    blob[name] = c
json.dump(blob, open('profiles/%s_linear_bwd_pmc.json' % tag, 'w'), indent=1)
print(json.dumps({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if 'fraction' in kk or kk == 'kernel_us_in_pmc_pass'} for k, v in blob.items() if k != '_source'}, indent=1))
PY
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_linear_bwd_pmc.json gpurun_out/profiles_$TAG/
