#!/bin/bash
# Whole-box CPU baseline of the SAMPLER ("mpiexec -np <physical cores> of rf_inv on the same
# params.in"): the reference itself -- all twelve sources unmodified, its own module fftw on the FFTW3 interface of the
# image's Intel MKL, dgesvd from MKL; oracle/_ref/cpu_o2/rf_inv (oracle/Makefile.ref), NO GPU involved -- one MPI rank
# per core, on a run directory with the shape of a BASELINE config (tests/tools/shape_run.py).  The rate is that of the
# sampler's LOOP: the program is run with N1 and with N2 iterations and the difference of the two wall times is divided by
# the difference of the iteration counts (start-up -- reading, init_model's rejection loop, the SVD of init_r_inv -- and
# the output files cancel).  Next to it, when a GPU is there: the batched GPU sampler on the same directory
# (drive_rfinv mode 4, one rank), whose own timer brackets its loop.
#   usage: tests/tools/sampler_rate_reference.sh <c3|c4|c5> <chains per rank> <ranks> [N1 = 100] [N2 = 400]
SHAPE=${1:-c4}; NCH=${2:-8}; NP=${3:-16}; N1=${4:-100}; N2=${5:-400}
R=${GRAFT_REPO_ROOT:-$(pwd)}
MPIEXEC=${MPIEXEC:-/opt/conda/bin/mpiexec}
EXE=$R/oracle/_ref/cpu_o2/rf_inv
[ -x $EXE ] || { echo "$EXE not built (make -C oracle -f Makefile.ref)"; exit 1; }
W=$(mktemp -d)
python3 $R/tests/tools/shape_run.py $SHAPE $NCH $W > /dev/null || { echo "shape_run failed"; exit 1; }
run() {  # iterations -> seconds of the whole program
  python3 - $W/params.in $1 <<'PY'
import sys
path, nit = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
body = [i for i, l in enumerate(lines) if l.strip() and not l.lstrip().startswith("#")]
lines[body[2]] = nit                    # N_ITER: the third value of params.in (src/params.f90:101-130)
open(path, "w").write("\n".join(lines))
PY
  rm -rf $W/rslt/*
  local t0=$(date +%s.%N)
  (cd $W && MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 $MPIEXEC -np $NP $EXE params.in > ref.log 2>&1) || { tail -5 $W/ref.log >&2; echo nan; return; }
  local t1=$(date +%s.%N)
  python3 -c "print($t1 - $t0)"
}
T1=$(run $N1); T2=$(run $N2)
python3 - <<PY
n1, n2, nch, np_ = $N1, $N2, $NCH, $NP
t1, t2 = $T1, $T2
print(f"shape $SHAPE: the reference's own rf_inv on host cores only (oracle/_ref/cpu_o2), {np_} MPI ranks x {nch} chains: "
      f"{n1} iterations {t1:.2f} s, {n2} iterations {t2:.2f} s -> loop {np_ * nch * (n2 - n1) / (t2 - t1):.3e} MCMC steps/s on {np_} cores "
      f"(whole program at {n2} iterations: {np_ * nch * n2 / t2:.3e})")
PY
if [ -e /dev/kfd ]; then
  rm -rf $W/rslt/*
  (cd $W && $R/oracle/_ref/drive_rfinv params.in 0 $N1 4 > gpu.log 2>&1) && grep "loop seconds" $W/gpu.log | head -2
fi
rm -rf $W
