#!/bin/bash
# End-to-end sampler rate (MCMC steps/s) of the Fortran host on the GPU box at BASELINE shapes: the batched driver
# (pt_control_batched, rf_inv_amd/fortran/pt_mcmc_batched.f90) on top of librfgpu, `total` chains on ONE GPU split
# over R MPI ranks (ranks that share a GPU exchange temperatures over MPI; each has its own context and stream, so
# one rank's proposal loop on the host overlaps another rank's kernels).
#   usage: tests/tools/sampler_rate_shapes.sh <c3|c4|c4w20> <total chains> <iterations> ["1 2 4"] [driver mode]
#          driver mode 1: pt_control_batched with its two-segment pipeline (default); 2: without it; 4: one or two
#          segments by the chain count (pt_control_batched's default); 0: the reference's own
#          pt_control on the per-call drop-in.  RFINV_TIME_KERNELS=1: HIP-event kernel totals of the loop are printed too
# The driver times its own loop (mpi_wtime around pt_control*, barriers on both sides); a short warm-up run comes
# first so that the timed one does not pay the image's first page-in.  (ref: the loop timed is src/pt_mcmc.f90:488-571)
SHAPE=${1:-c4}; TOTAL=${2:-8192}; NIT=${3:-200}; RANKS=${4:-"1 2 4"}; MODE=${5:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
MPIEXEC=${MPIEXEC:-/opt/conda/bin/mpiexec}
for np in $RANKS; do
  W=$(mktemp -d)
  nch=$((TOTAL / np))
  python3 $R/tests/tools/shape_run.py $SHAPE $nch $W > /dev/null || { echo "shape_run failed"; exit 1; }
  run() { (cd $W && $MPIEXEC -np $np $R/oracle/_ref/drive_rfinv params.in 0 $1 $MODE > run_$1.log 2>&1) || { tail -5 $W/run_$1.log; }; }
  run 3
  run $NIT
  python3 - "$W/run_$NIT.log" $SHAPE $MODE <<'PY'
import sys
for line in open(sys.argv[1]):
    if "loop seconds" in line and "batched" not in line:
        t = line.split()
        sec, ranks, nch, nit = float(t[3]), int(t[5]), int(t[7]), int(t[9])
        print(f"shape {sys.argv[2]} mode {sys.argv[3]} ranks {ranks} x {nch} chains, {nit} iterations: {sec:.3f} s loop -> "
              f"{ranks * nch * nit / sec:.3e} MCMC steps/s, {1e3 * sec / nit:.3f} ms per iteration")
    if "batched loop seconds" in line:
        sec = float(line.split()[-1])
        print(f"    the iteration loop alone (set-up excluded): {sec:.3f} s -> {ranks * nch * nit / sec:.3e} MCMC steps/s, "
              f"{1e3 * sec / nit:.3f} ms per iteration")
    if "phase seconds" in line:
        ph = [float(x) for x in line.split()[-5:]]
        print("    rank 0, ms per iteration: propose %.3f  eval %.3f  accept+commit %.3f  record %.3f  swap %.3f"
              % tuple(1e3 * x / nit for x in ph))
    if "engine call seconds" in line:
        ph = [float(x) for x in line.split()[-4:]]
        print("    rank 0, engine calls, ms per iteration: wait for the GPU %.3f  commit %.3f  record %.3f  begin %.3f"
              % tuple(1e3 * x / nit for x in ph))
PY
  rm -rf $W
done
