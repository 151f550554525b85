"""bench.py on the DIAGNOSTICS library with one ablation phase, 3 steps (tools/ablate_counters.sh runs it under rocprofv3):
the ablated kernels return early, so logL is garbage -- the finiteness check is relaxed here, nothing else."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, ab = sys.argv[1], sys.argv[2]
sys.argv = ["bench.py", "--workload", wl, "--also", "", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--prewarm-seconds", "0",
            "--lib", os.path.join(R, "tools", "_ab", "librfgpu_diag.so")] + (["--opt", f"ablate={ab}"] if ab != "0" else [])
import numpy as np

np.isfinite = lambda x: np.ones_like(np.asarray(x), dtype=bool)
exec(open(os.path.join(R, "bench.py")).read(), {"__name__": "__main__", "__file__": os.path.join(R, "bench.py")})
