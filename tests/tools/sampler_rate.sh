#!/bin/bash
# End-to-end sampler rate (MCMC steps/s) of the Fortran drop-in on the GPU box:
#   mode 0 = reference pt_control, one GPU call per chain step (per-call drop-in)
#   mode 1 = pt_control_batched (one rf_eval_batch per iteration)
# usage: tests/tools/sampler_rate.sh <nchains> <niter>
NCH=${1:-4096}; NIT=${2:-100}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for mode in 0 1; do
  W=$(mktemp -d)
  cp -r $R/tests/golden/sample_syn/* $W/; mkdir -p $W/rslt
  # N_CHAINS is the 5th non-comment line of params.in
  python3 - "$W/params.in" $NCH <<'PY'
import sys
path, nch = sys.argv[1], sys.argv[2]
out, n = [], 0
for line in open(path):
    if line.strip() and not line.lstrip().startswith("#"):
        n += 1
        if n == 5:
            line = nch + "\n"
    out.append(line)
open(path, "w").writelines(out)
PY
  it=$NIT; [ $mode = 0 ] && it=$((NIT / 10 > 0 ? NIT / 10 : 1))
  s=$(date +%s.%N)
  (cd $W && $R/oracle/_ref/drive_rfinv params.in 0 $it $mode > run.log 2>&1) || { tail -3 $W/run.log; }
  e=$(date +%s.%N)
  # subtract a zero-iteration run (init cost: first evaluation of every chain etc.)
  s0=$(date +%s.%N); (cd $W && $R/oracle/_ref/drive_rfinv params.in 0 0 $mode > run0.log 2>&1); e0=$(date +%s.%N)
  python3 -c "
t = ($e - $s) - ($e0 - $s0)
print('mode $mode nchains $NCH iters $it: %.3f s loop -> %.3e MCMC steps/s' % (t, $NCH * $it / max(t, 1e-9)))"
  rm -rf $W
done
