#!/bin/bash
# A/B timing of two builds of librfgpu in one process sequence on one GPU box:
#   A = rf_inv_amd/lib/librfgpu_A.so (reference build), B = rf_inv_amd/lib/librfgpu.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
for wl in c2 c4; do
for v in A B; do
  lib=$R/rf_inv_amd/lib/librfgpu.so; [ $v = A ] && lib=$R/rf_inv_amd/lib/librfgpu_A.so
  RFGPU_LIB=$lib python bench.py --workload $wl --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep $wl $v', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), d['kernel_ms'], '' )"
done; done; done
# parity of both builds against the CPU oracle on the same walkers (bench's cpu_baseline leg)
for wl in c2 c4; do
for v in A B; do
  lib=$R/rf_inv_amd/lib/librfgpu.so; [ $v = A ] && lib=$R/rf_inv_amd/lib/librfgpu_A.so
  RFGPU_LIB=$lib python bench.py --workload $wl --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('parity $wl $v', d['parity_in_bench'])"
done; done
