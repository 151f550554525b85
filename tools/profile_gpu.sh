#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats and the two HBM PMC
# passes (separately, as MI355X_MICROARCH.md prescribes) over bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BARGS="--no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 50 --warmup 5 $BARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 $BARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 $BARGS > $OUT/bench_write.json 2> $OUT/write.log
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge-back small
find $OUT -name "*.csv" -size +8M -delete
