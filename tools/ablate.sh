#!/bin/bash
# where the fused kernel spends its time: a DIAGNOSTICS build (-DRFGPU_DIAGNOSTICS, tools/_ab/librfgpu_diag.so,
# built here; never the shipped library) stops every block after phase N -- results invalid, timing only:
#   5: launch + staging only, 1: + propagator phase (no tail), 2: + FFT, 3: + max/shift/store,
#   4: + quadratic form, 0: full kernel   (8-wave kernels: 1, 2, 3, 0; fusedc_kernel also 6: one trace only;
#   4-wave kernel also 7: everything but the boundary condition + deposit of the bins)
# ABL_ARGS: extra bench.py arguments (e.g. "--walkers 512": one round of blocks = pure block latency)
# ABL_LIST: the phases to run (default "5 1 2 3 4 0")
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/tools/_ab
make -C $R/rf_inv_amd/csrc -s OUT=$R/tools/_ab/librfgpu_diag.so EXTRA=-DRFGPU_DIAGNOSTICS || exit 1
for wl in ${@:-c2 c4}; do
for ab in ${ABL_LIST:-5 1 2 3 4 0}; do
  python - $wl $ab <<'PY'
import json, subprocess, sys
wl, ab = sys.argv[1], sys.argv[2]
# bench.py asserts finite logL; the ablated runs are timed through the same loop with the check relaxed here
import os
sys.argv = ["bench.py", "--workload", wl, "--also", "", "--steps", "80", "--warmup", "10", "--no-cpu-baseline",
            "--lib", os.path.join("tools", "_ab", "librfgpu_diag.so")] + (["--opt", f"ablate={ab}"] if ab != "0" else []) \
           + os.environ.get("ABL_ARGS", "").split()
import numpy as np
_isfinite = np.isfinite
np.isfinite = lambda x: np.ones_like(np.asarray(x), dtype=bool)
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    exec(open("bench.py").read(), {"__name__": "__main__", "__file__": os.path.abspath("bench.py")})
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(wl, "ablate=" + ab, d["kernel_ms"], "step", round(d["ms_per_step"], 4))
PY
done; done
