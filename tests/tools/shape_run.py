#!/usr/bin/env python3
"""A run directory (params.in + SAC traces + reference velocity table) with the SHAPE of a BASELINE config, for the
end-to-end sampler-rate measurements (tests/tools/sampler_rate_shapes.sh): what `drive_rfinv` / `python -m
rf_inv_amd.run` read.

    shape_run.py <c3|c4|c4w20|c5> <nchains per rank> <dir> [ncool]

c3: nfft 4096, 1 P trace (p 0.06), k_max 15.   c4: 3 traces (P .06, P .08, S .10), k_max 30.   c4w20: c4 with a 20 s
window.  c5: 4 traces (P .06, P .08, S .10, S .12) under a 2 km ocean layer (interfaces 2 .. 22 km).  Observed traces: the noise-free synthetic of bench.py's fixed 3-interface model through the CPU oracle (test
infrastructure; this script is a test tool), written in the reference's SAC layout (rf_inv_amd.make_syn.write_sac).
Everything else is tests/golden/sample_syn/params.in, with two departures that make the reference's own init_model
(src/model.f90:66-95: whole-model rejection) finish at k_max 30: dVs prior width 0.3 km/s instead of 2.0 (with 2.0
a 30-layer model is valid with probability ~1e-5).  The proposal widths are the sample's.
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPES = {
    "c3": dict(rayps=[0.06], ipha=[1], k_max=15, t_end=5.0),
    "c4": dict(rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, t_end=5.0),
    "c4w20": dict(rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, t_end=20.0),
    "c5": dict(rayps=[0.06, 0.08, 0.10, 0.12], ipha=[1, 1, -1, -1], k_max=30, t_end=5.0, sdep=2.0),
}


def main():
    shape, nchains, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    ncool = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, nchains // 8)
    s = SHAPES[shape]
    ntrc = len(s["rayps"])
    golden = os.path.join(ROOT, "tests", "golden", "sample_syn")
    os.makedirs(os.path.join(out, "data"), exist_ok=True)
    os.makedirs(os.path.join(out, "model"), exist_ok=True)
    os.makedirs(os.path.join(out, "rslt"), exist_ok=True)
    shutil.copy(os.path.join(golden, "model", "sample.velmod"), os.path.join(out, "model", "sample.velmod"))

    from oracle import rf_oracle as orc
    from rf_inv_amd import format_model, get_params, read_ref_model
    from rf_inv_amd.make_syn import write_sac
    from rf_inv_amd.params import _nint

    orc.build()
    p = get_params(os.path.join(golden, "params.in"))
    ref = read_ref_model(os.path.join(golden, "model", "sample.velmod"))
    sdep = float(s.get("sdep", 0.0))
    p.k_max, p.sdep = s["k_max"], sdep
    delta = float(np.float32(0.05))
    nsmp = _nint(s["t_end"] / delta) + 1
    zt = np.zeros(p.k_max - 1); dvt = np.zeros(p.k_max); dst = np.zeros(p.k_max)
    zt[:3] = [3.1 + sdep, 7.7 + sdep, 14.2 + sdep]; dst[:3] = [-0.6, 0.2, 0.5]; dst[p.k_max - 1] = 0.9
    nl, a, b, r, h, ok = format_model(p, ref, 3, zt, dvt, dst)
    assert ok
    cfg = dict(nfft=4096, deconv_mode=0, delta=delta, t_start=0.0, sdep=sdep, rayps=np.array(s["rayps"]),
               a_gus=np.full(ntrc, 4.0), ipha=np.array(s["ipha"], dtype=np.int32))
    rft = orc.calc_rf(cfg, a, b, r, h)                        # [ntrc, nfft]
    for t in range(ntrc):
        write_sac(os.path.join(out, "data", f"shape_{t + 1}.trc"), rft[t, :nsmp], delta, 0.0, s["t_end"])

    L = ["'./rslt'", "0", "100", "10", str(nchains), str(ncool), "15.0", "12345678", str(ntrc)]
    L += [repr(x) for x in s["rayps"]] + ["4.0"] * ntrc + [str(x) for x in s["ipha"]] + ["4096"]
    L += [f"'data/shape_{t + 1}.trc'" for t in range(ntrc)]
    L += [f"0.0 {s['t_end']}", "0", repr(sdep), '"model/sample.velmod"', "0", f"1 {s['k_max']}", f"{sdep} {20.0 + sdep}", "0.05", "2",
          "0.3", "0.2"]
    L += ["0.01 0.01"] * ntrc
    L += ["0.02", "0.02", "0.02", "0.002", "100", "50", "50", "100", "50", "100", "-0.8 0.8", "0.1 8.6", "0.001 5.0",
          "0.0 5.0"]
    with open(os.path.join(out, "params.in"), "w") as fh:
        fh.write(f"# {shape}-shaped run written by tests/tools/shape_run.py\n" + "\n".join(L) + "\n")
    q = get_params(os.path.join(out, "params.in"))            # the file parses the way it was meant
    assert (q.nfft, q.ntrc, q.k_max, q.nchains) == (4096, ntrc, s["k_max"], nchains)
    print(f"{out}: {shape} shape, {nchains} chains per rank ({ncool} at T = 1), nsmp {nsmp}")


if __name__ == "__main__":
    main()
