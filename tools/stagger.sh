#!/bin/bash
# the phase-stagger experiment (profiles/EXPERIMENTS.md, round 6): DIAGNOSTICS build, "ablate" 100 + X = every other block
# idles X us before it starts (1000 + X: only in the first round of resident blocks); results stay valid.
#   tools/stagger.sh "c4common c2 c4" "0 108 116 124 1016"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for wl in $1; do
for ab in $2; do
  opt=""; [ $ab != 0 ] && opt="--opt ablate=$ab"
  python bench.py --workload $wl --also "" --steps 80 --warmup 10 --no-cpu-baseline --lib $R/tools/_ab/librfgpu_diag.so $opt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep $wl stagger=$ab', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), 'kernel_ms', d['roofline']['kernel_ms'], d['parity_in_bench']['max_rel_dlogl'])"
done; done; done
