#!/bin/bash
# interleaved A/B of one environment knob: tools/ab_env.sh VAR valA valB [workloads...]
VAR=$1; A=$2; B=$3; shift 3; WLS=${@:-c2 c4}
for rep in 1 2 3; do
for wl in $WLS; do
for v in $A $B; do
  env $VAR=$v python bench.py --workload $wl --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep $wl $VAR=$v', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), d['kernel_ms'])"
done; done; done
