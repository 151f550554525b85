#!/bin/bash
# Runs on the GPU box (via gpurun): hardware counters of every kernel bench.py launches, per launch and per
# launch shape, each counter set in its own rocprofv3 run (--pmc alone, never with a trace domain;
# FETCH_SIZE and WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes), plus one
# --kernel-trace --stats run.  tools/summarize_counters.py condenses them into
# gpurun_out/counters_<tag>/counters.json (copy to profiles/<round>_counters.json: bench.py reads it for
# roofline.frac / roofline.traffic) and kernel_stats.txt.
# usage: tools/collect_counters.sh <tag> [bench args...]        (default bench args: the default run)
set -u
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/counters_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BARGS="--no-cpu-baseline --prewarm-seconds 0 $*"
rocprofv3 -L > $OUT/available.txt 2>&1
filter() { # keep the counters this rocprofv3 knows
  local keep=""
  for c in "$@"; do grep -qw "$c" $OUT/available.txt && keep="$keep $c"; done
  echo $keep
}
i=0; fails=0
# small sets (a failed pass costs one short run, and a set that faults is easy to single out); every pass under its
# own timeout -- a faulted rocprofv3 otherwise waits for ever on its incomplete dispatches
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  use=$(filter $set)
  [ -z "$use" ] && continue
  timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 --pmc $use --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 $BARGS > $OUT/p$i.json 2> $OUT/p$i.log
  rc=$?
  echo "pass $i [$use] rc=$rc" >> $OUT/passes.txt
  if [ $rc -ne 0 ]; then fails=$((fails+1)); else fails=0; fi
  [ $fails -ge 2 ] && { echo "two passes in a row failed: stopping" >> $OUT/passes.txt; break; }
done
timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 50 --warmup 5 $BARGS > $OUT/bench_trace.json 2> $OUT/trace.log
python3 $R/tools/summarize_counters.py $OUT $R > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete    # keep the merge-back small
find $OUT -name "*.db" -delete
