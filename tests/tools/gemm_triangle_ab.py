#!/usr/bin/env python3
"""Long-window likelihood: phi_gemm_kernel on the quadratic form's upper triangle (default) against the full product
(rf_set_option "gemm_triangle" 0): kernel times at bench.py's c4w20 / c4w60, interleaved, with the in-bench parity of
each.   usage: tests/tools/gemm_triangle_ab.py [reps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(reps):
    for rep in range(reps):
        for wl in ("c4w60", "c4w20"):
            for tri in (0, 1):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--also", "", "--steps", "60",
                                    "--warmup", "10", "--no-cpu-baseline", "--opt", f"gemm_triangle={tri}"], capture_output=True, text=True)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                q, par = d["quadratic_form_gemm"], d["parity_in_bench"]
                print(f"rep{rep} {wl} triangle {tri}: step {d['ms_per_step']:.3f} ms, fused {d['kernel_ms']['fused']:.3f}, GEMM + logL "
                      f"{q['ms']:.3f} ms = {q['achieved']:.1f} TF run, {q['algorithmic']['tflops']:.1f} TF algorithmic; parity n {par['n']} worst rel {par['max_rel_dlogl']:.1e} "
                      f"within tolerance {par['within_tolerance']}", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
