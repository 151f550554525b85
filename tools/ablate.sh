#!/bin/bash
# where the fused kernel spends its time: stop after phase N (results invalid)
#   5: launch + staging only, 1: + propagator phase (no tail), 2: + FFT, 3: + max/shift/store,
#   4: + quadratic form, 0: full kernel
for wl in ${@:-c2 c4}; do
for ab in 5 1 2 3 4 0; do
  RFGPU_ABLATE=$ab python bench.py --workload $wl --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl ablate=$ab', d['kernel_ms'], 'step', round(d['ms_per_step'], 4))"
done; done
