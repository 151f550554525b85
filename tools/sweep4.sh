#!/bin/bash
for wpb in 1 2 4; do
for ns in 1 2 4 8; do
  RFGPU_NSPLIT=$ns RFGPU_WPB=$wpb python bench.py --workload c2 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c2 wpb=$wpb nsplit=$ns', round(d['value']), 'evals/s', {k: round(v, 4) for k, v in d['kernel_ms'].items()})"
done; done
