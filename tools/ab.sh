#!/bin/bash
# A/B timing of two builds of librfgpu, interleaved, on one GPU box:
#   A = tools/_ab/librfgpu_A.so (tools/build_baseline.sh <rev>), B = rf_inv_amd/lib/librfgpu.so
# usage: tools/ab.sh [workloads...]   (default c2 c4)
R=${GRAFT_REPO_ROOT:-$(pwd)}
WLS=${@:-c2 c4}
for rep in 1 2 3; do
for wl in $WLS; do
for v in A B; do
  lib=""; [ $v = A ] && lib="--lib $R/tools/_ab/librfgpu_A.so"
  python bench.py --workload $wl --also "" --steps 80 --warmup 10 --no-cpu-baseline $lib 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep $wl $v', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), 'median', d['ms_per_step_median'], d['kernel_ms'], d['parity_in_bench'])"
done; done; done
