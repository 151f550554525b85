#!/bin/bash
# interleaved timing of several builds of librfgpu on one GPU box:
#   tools/ab3.sh "c4 c3" A:tools/_ab/librfgpu_A.so B:tools/_ab/librfgpu_B.so C:        (an empty path = rf_inv_amd/lib/librfgpu.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
WLS=$1; shift
for rep in 1 2 3; do
for wl in $WLS; do
for spec in "$@"; do
  v=${spec%%:*}; path=${spec#*:}
  lib=""; [ -n "$path" ] && lib="--lib $R/$path"
  python bench.py --workload $wl --also "" --steps 80 --warmup 10 --no-cpu-baseline $lib 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep $wl $v', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), 'kernel_ms', d['roofline']['kernel_ms'], d['parity_in_bench'])"
done; done; done
