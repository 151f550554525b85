#!/bin/bash
# Soak of rows f1 / f3 (GPU box): 20000 iterations of the shipped sample_syn inversion, once through the reference's
# unmodified pt_control on the per-call drop-in (mode 0) and once through pt_control_batched with the posterior
# recorded on the device (mode 1) -- one checksum over the trajectory dump and every result file mcmc_out writes.
# The two lines must be equal (the pytest suites compare the same things over 300 iterations).
R=$GRAFT_REPO_ROOT
for mode in 0 1; do
  W=$(mktemp -d); cp -r $R/tests/golden/sample_syn/* $W/; mkdir -p $W/rslt
  (cd $W && $R/oracle/_ref/drive_rfinv params.in 4000 16000 $mode out > run.log 2>&1; tail -1 run.log; md5sum rfinv_dump.txt rslt/* | awk '{print $1}' | md5sum)
done
