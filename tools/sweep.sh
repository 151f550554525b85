#!/bin/bash
# quick launch-shape sweep on the GPU box: prints evals/s + kernel ms per setting
for wl in c2 c4; do
for bins in 1 2; do
for ns in 0 1 2 4 8; do
  RFGPU_NSPLIT=$ns RFGPU_BINS_PER_LANE=$bins python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl bins=$bins nsplit=$ns', round(d['value']), 'evals/s', {k: round(v, 4) for k, v in d['kernel_ms'].items()})"
done; done; done
