#!/bin/bash
# rocprofv3 PMC pass over the ellipsoid kernels (wave-cycle split and VALU activity)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_ell
rm -rf "$OUT"; mkdir -p "$OUT"
S="$PWD/scripts/time_ellipsoids.py"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -T --kernel-include-regex "ellipsoid" -d "$OUT/pmc" -- python3 "$S" 400000 > "$OUT/run.log" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1.0)
    print(k, {c: ("%.3g" % v) for c, v in m.items()})
    print("   fractions of wave cycles: wait_any %.2f wait_inst %.2f active %.2f valu-active %.2f" % (
        m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_VALU", 0) / wc))
PY
