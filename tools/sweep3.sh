#!/bin/bash
for ch in 0 2 3 4 8; do
  echo "== RFGPU_CHAIN=$ch"
  RFGPU_CHAIN=$ch python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
  for wl in c2 c4 c1; do
  RFGPU_CHAIN=$ch python bench.py --workload $wl --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl chain=$ch', round(d['value']), 'evals/s', {k: round(v, 4) for k, v in d['kernel_ms'].items()}, d.get('parity_in_bench'))"
  done
done
