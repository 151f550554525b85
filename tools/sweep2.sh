#!/bin/bash
for wl in c2 c4; do
for ch in 1 2 4 8; do
  RFGPU_CHUNKS=$ch python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl chunks=$ch', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), 'ms/step', {k: round(v, 4) for k, v in d['kernel_ms'].items()})"
done; done
