#!/bin/bash
# where the fused kernel's INSTRUCTIONS are: the DIAGNOSTICS build (blocks stop after phase N, tools/ablate.sh) under
# rocprofv3 --pmc: VALU / SALU / LDS wave-instructions and the four fp64 VALU counters of the dominant kernel per phase.
#   usage: tools/ablate_counters.sh <workload> "<phases, default 5 1 7 2 3 0>"   -> gpurun_out/ablate_counters_<workload>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-c4}; PH=${2:-"5 1 7 2 3 0"}
mkdir -p $R/tools/_ab $R/gpurun_out
make -C $R/rf_inv_amd/csrc -s OUT=$R/tools/_ab/librfgpu_diag.so EXTRA=-DRFGPU_DIAGNOSTICS || exit 1
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/ablate_counters_$WL
rm -rf $OUT; mkdir -p $OUT
for ab in $PH; do
  opt=""; [ $ab != 0 ] && opt="--opt ablate=$ab"
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/a${ab}_p$i -- python3 $R/tools/ablate_bench.py $WL $ab > $OUT/a${ab}_p$i.log 2>&1
  done
done
python3 - $OUT <<'PY' | tee $R/gpurun_out/ablate_counters_$(basename $OUT | sed s/ablate_counters_//).txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "a*_p*/**/*counter_collection.csv"), recursive=True):
    ab = os.path.relpath(f, out).split("_")[0][1:]
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if "fused" not in k or int(r["Grid_Size"]) < 100000:
                continue
            a = res[ab][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
print("# phase: VALU wave-instructions, of which fp64 (ADD+MUL+FMA+TRANS), other VALU, SALU, LDS   (dominant fused kernel, per launch)")
for ab in sorted(res, key=lambda x: {"5": 0, "1": 1, "7": 2, "2": 3, "3": 4, "4": 5, "0": 9}.get(x, 8)):
    m = {k: v[1] / v[0] for k, v in res[ab].items()}
    f64 = sum(m.get("SQ_INSTS_VALU_%s_F64" % x, 0.0) for x in ("ADD", "MUL", "FMA", "TRANS"))
    print(f"ablate={ab}: VALU {m.get('SQ_INSTS_VALU', 0):.4e}  fp64 {f64:.4e}  other VALU {m.get('SQ_INSTS_VALU', 0) - f64:.4e}  "
          f"SALU {m.get('SQ_INSTS_SALU', 0):.4e}  LDS {m.get('SQ_INSTS_LDS', 0):.4e}")
PY
