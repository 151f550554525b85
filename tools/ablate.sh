#!/bin/bash
# where the fused kernel's tail spends its time: stop after phase N (results invalid)
for wl in c2 c4; do
for ab in 1 2 3 4 0; do
  RFGPU_ABLATE=$ab python bench.py --workload $wl --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl ablate=$ab (1: no tail, 2: +FFT, 3: +max/shift/store, 4: +quad form, 0: full)', d['kernel_ms'])" 2>/dev/null || echo "$wl ablate=$ab failed (non-finite logL expected)"
done; done
