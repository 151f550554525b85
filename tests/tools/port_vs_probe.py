"""Single-core rate of the CPU port (oracle/rf_oracle.c, speed build) at exactly the shapes SURVEY.md section 6
probed the UNMODIFIED reference at (amdflang -O2, one core of this container class), so that bench.py's
cpu_baseline (kind "port") can carry a reference/port ratio.  The reference's forward/likelihood modules cannot
be built here without stand-ins for FFTW3 / LAPACK: the probe figures are the surveyor's.

Run in the BUILD container (same machine class as the probe):  python tests/tools/port_vs_probe.py > profiles/rNN_cpu_port_vs_reference_probe.json
"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DELTA, make_cfg, pack_layers, random_stack  # noqa: E402
from oracle import rf_oracle as orc  # noqa: E402

PROBE = {   # SURVEY.md section 6, "[probe]" rows
    "c2@15": dict(nl=15, rayps=[0.06], ipha=[1], reference=[175.0, 205.0],
                  what="calc_rf, nfft 4096, 1 P trace, 15 layers"),
    "c4@30": dict(nl=30, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], reference=[29.0, 37.0],
                  what="calc_rf, nfft 4096, 3 traces (P,P,S; rays 0.06/0.08/0.10), 30 layers"),
}


def main():
    rng = np.random.default_rng(1)
    out = {"host": platform.processor() or platform.machine(), "cpu_model": "", "flags": " ".join(orc.FAST_FLAGS),
           "source": "tests/tools/port_vs_probe.py; reference figures: SURVEY.md section 6 probe of the unmodified reference",
           "shapes": {}}
    try:
        out["cpu_model"] = [ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    for name, s in PROBE.items():
        cfg = make_cfg(nfft=4096, rayps=s["rayps"], ipha=s["ipha"])
        stacks = [random_stack(rng, s["nl"]) for _ in range(16)]
        nlay, layers = pack_layers(stacks, s["nl"] + 2)
        ntrc = len(s["rayps"])
        obs = np.zeros((ntrc, 101))
        r_inv = orc.build_r_inv(101, cfg["a_gus"], DELTA)
        sig = np.full((16, ntrc), 0.01)
        best = 0.0
        for _ in range(3):
            t = time.perf_counter()
            orc.eval_batch(cfg, obs, r_inv, nlay, layers, sig, 101, nthreads=1, fast=True)
            best = max(best, 16 / (time.perf_counter() - t))
        lo, hi = s["reference"]
        out["shapes"][name] = {"what": s["what"], "port_evals_per_s_per_core": round(best, 1),
                               "reference_probe_evals_per_s_per_core": [lo, hi],
                               "reference_over_port": [round(lo / best, 2), round(hi / best, 2)]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
