#!/bin/bash
# SQ counters of named kernels in the headline step (development): tools/pmc_kernels.sh <substring> [<substring> ...]
R=$(pwd); OUT=$R/gpurun_out/pmc_kernels; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/$tag.log 2>&1 || echo "FAILED $grp"
done
cd $R
python3 - "$OUT" "$@" <<'PY'
import collections, csv, glob, sys
out, names = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if names == ["ALL"]:
            if "mlgnn" in k:
                res[k.split("(")[0].replace("void ", "")[:70]][row['Counter_Name']].append(float(row['Counter_Value']))
            continue
        for name in names:
            if name in k:
                res[name + " grid=" + row.get('Grid_Size', '?')][row['Counter_Name']].append(float(row['Counter_Value']))
if names == ["ALL"]:
    rows = []
    for name, c in res.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        n = len(c.get("GRBM_GUI_ACTIVE", []))
        dur = m.get("GRBM_GUI_ACTIVE", 0) / 8 / 2.4e3                      # us at 2.4 GHz
        rows.append((dur * n, name, n, dur, m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
                     m.get("SQ_ACTIVE_INST_VALU", 0) / max(1024.0 * m.get("GRBM_GUI_ACTIVE", 1) / 8 / 4, 1),
                     m.get("SQ_LDS_IDX_ACTIVE", 0) / 256.0 / max(m.get("GRBM_GUI_ACTIVE", 1) / 8, 1),
                     m.get("SQ_WAVE_CYCLES", 0) * 4 / max(m.get("GRBM_GUI_ACTIVE", 1) / 8, 1) / 256))
    print("%-70s %5s %9s %8s %8s %8s %9s" % ("kernel", "n", "avg us", "wait", "valu", "lds", "waves/CU"))
    for tot, name, n, dur, w, v, l, occ in sorted(rows, reverse=True)[:40]:
        print("%-70s %5d %9.1f %8.2f %8.2f %8.2f %9.1f" % (name, n, dur, w, v, l, occ))
else:
    for name, c in sorted(res.items()):
        print(name)
        for k, v in sorted(c.items()):
            print("   %-24s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
find $OUT -name "*kernel_trace.csv" -delete
