#!/bin/bash
# interleaved A/B of one launch-plan option (rf_set_option): tools/ab_opt.sh NAME valA valB [workloads...]
NAME=$1; A=$2; B=$3; shift 3; WLS=${@:-c2 c4}
for rep in 1 2 3; do
for wl in $WLS; do
for v in $A $B; do
  python bench.py --workload $wl --also "" --steps 80 --warmup 10 --no-cpu-baseline --opt $NAME=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep $wl $NAME=$v', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), d['kernel_ms'])"
done; done; done
