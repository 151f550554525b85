"""The reference's own calc_likelihood on the host cores: evaluations/s per process with 1, 4 and <cores> processes at once,
for the -O2 and the -O0 build of oracle/Makefile.ref (the WHOLE reference, unmodified, MKL's FFTW3 interface + LAPACK; no
GPU anywhere).  bench.py's cpu_baseline is the <cores> row of the -O2 table; this tool shows that the per-core rate at
<cores> is the per-core rate at 1 (round 5's baseline ran its transforms on the GPU and was shaped by GPU queueing).
usage: python tests/tools/reference_cpu_scaling.py [workload, default c4] [seconds per point, default 4]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    from oracle import refrun

    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    p = bench.make_params(dict(bench.WORKLOADS[wl]))
    # observed traces: the reference's own synthetic of bench.py's fixed model is not needed for a RATE: zeros do
    obs = np.zeros((p.ntrc, p.nsmp))
    cores = bench.physical_cores()
    print(f"# workload {wl}: calc_likelihood(fwd_flag = .true.) on 24 of its walkers, {cores} physical cores, {secs:.0f} s per point")
    for build in refrun.BUILDS[::-1]:
        for procs in sorted({1, min(4, cores), cores}):
            rec, _ = bench.reference_path_rate(p, obs, budget_s=secs, count=24, procs=procs, build=build)
            if rec is None:
                raise SystemExit(f"{build}: not built or failed")
            print(f"{build} {procs:3d} process(es): {rec['value']:9.1f} evals/s, {rec['per_core']:7.1f} per process")


if __name__ == "__main__":
    main()
