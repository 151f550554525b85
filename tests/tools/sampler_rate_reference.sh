#!/bin/bash
# Whole-box CPU baseline of the SAMPLER ("mpiexec -np <physical cores> of rf_inv on the same
# params.in"): the reference's own main program on its OWN forward / likelihood modules (oracle/_ref/rf_inv_reference: all
# of the reference compiled unmodified; c2r through the drop-in module fftw = GPU round trips, dgesvd from MKL), one MPI
# rank per core, on a run directory with the shape of a BASELINE config (tests/tools/shape_run.py) -- next to the batched
# GPU sampler on the same directory (drive_rfinv mode 4, one rank).
#   usage: tests/tools/sampler_rate_reference.sh <c3|c4|c5> <chains per rank> <ranks> [iterations = 100 as shape_run writes]
SHAPE=${1:-c4}; NCH=${2:-8}; NP=${3:-16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
MPIEXEC=${MPIEXEC:-/opt/conda/bin/mpiexec}
W=$(mktemp -d)
python3 $R/tests/tools/shape_run.py $SHAPE $NCH $W > /dev/null || { echo "shape_run failed"; exit 1; }
NIT=100
t0=$(date +%s.%N)
(cd $W && HSA_ENABLE_SDMA=0 GPU_MAX_HW_QUEUES=1 $MPIEXEC -np $NP $R/oracle/_ref/rf_inv_reference params.in > ref.log 2>&1) || { tail -5 $W/ref.log; exit 1; }
t1=$(date +%s.%N)
python3 - <<PY
nit, nch, np_ = $NIT, $NCH, $NP
dt = $t1 - $t0
print(f"shape $SHAPE: the reference's own rf_inv (its forward / likelihood modules), {np_} MPI ranks x {nch} chains, {nit} iterations: "
      f"{dt:.2f} s wall incl. start-up and output -> {np_ * nch * nit / dt:.3e} MCMC steps/s on {np_} cores")
PY
rm -rf $W/rslt/*
(cd $W && $R/oracle/_ref/drive_rfinv params.in 0 $NIT 4 > gpu.log 2>&1) || { tail -5 $W/gpu.log; exit 1; }
grep "loop seconds" $W/gpu.log | head -2
rm -rf $W
