"""Host-side mirror of the reference's `module params` (src/params.f90): the
positional params.in parser and the SAC reader.  Same field order, same '#'
comment rule, same float32 header arithmetic.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

NPTS_MAX = 2000   # src/params.f90:44
NLAY_MAX = 200    # src/params.f90:44


@dataclass
class Params:
    """All public variables of `module params` (src/params.f90:46-96)."""
    out_dir: str = "."
    nburn: int = 0
    niter: int = 0
    ncorr: int = 1
    nchains: int = 1
    ncool: int = 1
    t_high: float = 1.0
    iseed: int = 0
    ntrc: int = 0
    rayps: np.ndarray = field(default_factory=lambda: np.zeros(0))
    a_gus: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ipha: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    nfft: int = 0
    obs_files: List[str] = field(default_factory=list)
    t_start: float = 0.0
    t_end: float = 0.0
    deconv_mode: int = 0
    sdep: float = 0.0
    vel_file: str = ""
    vp_mode: int = 0
    k_min: int = 1
    k_max: int = 2
    z_min: float = 0.0
    z_max: float = 0.0
    h_min: float = 0.0
    prior_mode: int = 2
    dvs_prior: float = 0.0
    dvp_prior: float = 0.0
    sig_min: np.ndarray = field(default_factory=lambda: np.zeros(0))
    sig_max: np.ndarray = field(default_factory=lambda: np.zeros(0))
    sig_mode: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    dev_z: float = 0.0
    dev_dvs: float = 0.0
    dev_dvp: float = 0.0
    dev_sig: float = 0.0
    nbin_z: int = 0
    nbin_vs: int = 0
    nbin_vp: int = 0
    nbin_vpvs: int = 0
    nbin_sig: int = 0
    nbin_amp: int = 0
    amp_min: float = 0.0
    amp_max: float = 0.0
    vp_min: float = 0.0
    vp_max: float = 0.0
    vs_min: float = 0.0
    vs_max: float = 0.0
    vpvs_min: float = 0.0
    vpvs_max: float = 0.0
    # filled by read_obs (src/params.f90:422-476)
    nsmp: int = 0
    delta: float = 0.0
    obs: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))  # obs[itrc, :NPTS_MAX]
    base_dir: str = "."


def _lines(path):
    """get_line (src/params.f90:392-405): left-adjust, skip lines starting with '#'."""
    with open(path, "r") as f:
        for raw in f:
            line = raw.strip()
            if not line or line.startswith("#"):
                if line.startswith("#") or not line:
                    # the reference only skips '#' lines; blank lines would fail its
                    # list-directed read, so treating them as skippable is a superset
                    continue
            yield line


def _fields(line):
    # Fortran list-directed input: blanks or commas separate, quotes delimit strings
    out, cur, quote = [], "", None
    for ch in line:
        if quote:
            if ch == quote:
                quote = None
                out.append(cur)
                cur = ""
            else:
                cur += ch
        elif ch in "'\"":
            quote = ch
        elif ch in " ,\t":
            if cur:
                out.append(cur)
                cur = ""
        elif ch == "/":
            break
        else:
            cur += ch
    if cur:
        out.append(cur)
    return out


def _f(tok):
    return float(tok.lower().replace("d", "e"))


def get_params(param_file: str, verb: bool = False) -> Params:
    """subroutine get_params (src/params.f90:101-388).  File order is the API."""
    if not os.path.exists(param_file):
        raise FileNotFoundError("ERROR: cannot open : params.in")  # src/params.f90:108-111
    it = _lines(param_file)
    nxt = lambda: _fields(next(it))
    p = Params()
    p.base_dir = os.path.dirname(os.path.abspath(param_file))
    p.out_dir = nxt()[0]
    p.nburn = int(nxt()[0]); p.niter = int(nxt()[0]); p.ncorr = int(nxt()[0])
    p.nchains = int(nxt()[0]); p.ncool = int(nxt()[0]); p.t_high = _f(nxt()[0])
    p.iseed = int(nxt()[0])
    p.ntrc = int(nxt()[0])
    n = p.ntrc
    p.rayps = np.array([_f(nxt()[0]) for _ in range(n)])
    p.a_gus = np.array([_f(nxt()[0]) for _ in range(n)])
    p.ipha = np.array([int(nxt()[0]) for _ in range(n)], dtype=np.int32)
    p.nfft = int(nxt()[0])
    p.obs_files = [nxt()[0] for _ in range(n)]
    t = nxt(); p.t_start, p.t_end = _f(t[0]), _f(t[1])
    p.deconv_mode = int(nxt()[0])
    if p.deconv_mode not in (0, 1):
        raise ValueError("ERROR: deconv_mode must be either 0 or 1")  # src/params.f90:195-199
    p.sdep = _f(nxt()[0])
    p.vel_file = nxt()[0]
    p.vp_mode = int(nxt()[0])
    t = nxt(); p.k_min, p.k_max = int(t[0]), int(t[1])
    t = nxt(); p.z_min, p.z_max = _f(t[0]), _f(t[1])
    p.h_min = _f(nxt()[0])
    p.prior_mode = int(nxt()[0])
    p.dvs_prior = _f(nxt()[0]); p.dvp_prior = _f(nxt()[0])
    p.sig_min = np.zeros(n); p.sig_max = np.zeros(n); p.sig_mode = np.zeros(n, dtype=np.int32)
    for i in range(n):
        t = nxt(); p.sig_min[i], p.sig_max[i] = _f(t[0]), _f(t[1])
        # src/params.f90:270: compared against the single-precision literal 1.0e-5
        p.sig_mode[i] = 1 if p.sig_max[i] - p.sig_min[i] > float(np.float32(1.0e-5)) else 0
    p.dev_z = _f(nxt()[0]); p.dev_dvs = _f(nxt()[0]); p.dev_dvp = _f(nxt()[0]); p.dev_sig = _f(nxt()[0])
    p.nbin_z = int(nxt()[0]); p.nbin_vs = int(nxt()[0]); p.nbin_vp = int(nxt()[0])
    p.nbin_vpvs = int(nxt()[0]); p.nbin_sig = int(nxt()[0]); p.nbin_amp = int(nxt()[0])
    t = nxt(); p.amp_min, p.amp_max = _f(t[0]), _f(t[1])
    t = nxt(); p.vp_min, p.vp_max = _f(t[0]), _f(t[1])
    t = nxt(); p.vs_min, p.vs_max = _f(t[0]), _f(t[1])
    t = nxt(); p.vpvs_min, p.vpvs_max = _f(t[0]), _f(t[1])
    if verb:
        print("--- Parameters ---")
        for k, v in p.__dict__.items():
            if k not in ("obs",):
                print(f"{k.upper()}: {v}")
    return p


def write_params(path: str, p: Params, header: str = ""):
    """The inverse of get_params: `p` as a params.in in the reference's positional order (src/params.f90:101-388), one
    value group per line, quoted strings where the reference's list-directed read needs them (OUT_DIR, the observation
    files, the velocity file).  get_params(write_params(p)) == p for every field of the file."""
    n = int(p.ntrc)
    r = lambda x: repr(float(x))
    L = [f"'{p.out_dir}'", str(int(p.nburn)), str(int(p.niter)), str(int(p.ncorr)), str(int(p.nchains)), str(int(p.ncool)),
         r(p.t_high), str(int(p.iseed)), str(n)]
    L += [r(x) for x in p.rayps[:n]] + [r(x) for x in p.a_gus[:n]] + [str(int(x)) for x in p.ipha[:n]] + [str(int(p.nfft))]
    L += [f"'{f}'" for f in p.obs_files[:n]]
    L += [f"{r(p.t_start)} {r(p.t_end)}", str(int(p.deconv_mode)), r(p.sdep), f"'{p.vel_file}'", str(int(p.vp_mode)),
          f"{int(p.k_min)} {int(p.k_max)}", f"{r(p.z_min)} {r(p.z_max)}", r(p.h_min), str(int(p.prior_mode)), r(p.dvs_prior),
          r(p.dvp_prior)]
    L += [f"{r(p.sig_min[i])} {r(p.sig_max[i])}" for i in range(n)]
    L += [r(p.dev_z), r(p.dev_dvs), r(p.dev_dvp), r(p.dev_sig)]
    L += [str(int(x)) for x in (p.nbin_z, p.nbin_vs, p.nbin_vp, p.nbin_vpvs, p.nbin_sig, p.nbin_amp)]
    L += [f"{r(p.amp_min)} {r(p.amp_max)}", f"{r(p.vp_min)} {r(p.vp_max)}", f"{r(p.vs_min)} {r(p.vs_max)}",
          f"{r(p.vpvs_min)} {r(p.vpvs_max)}"]
    with open(path, "w") as fh:
        if header:
            fh.write("".join(f"# {h}\n" for h in header.splitlines()))
        fh.write("\n".join(L) + "\n")
    return path


def _nint(x: float) -> int:
    return int(np.floor(x + 0.5)) if x >= 0 else -int(np.floor(0.5 - x))


def read_sac(path: str, t_start: float, t_end: float):
    """One file of read_obs (src/params.f90:436-459).  Direct access, recl = 4:
    delta @ record 1, b @ record 6, npts @ record 80, samples from record 159.
    delta4 and t_beg4 are default REAL, t_start / t_end are real(8) (src/params.f90:66), so
    `(t_start - t_beg4) / delta4` (:449-450) is evaluated in DOUBLE on the float32-valued header
    fields (Fortran promotes the mixed expression) -- a window edge half a sample off a grid point
    rounds differently in float32.  Returns (samples[nsmp] as float64, delta, nsmp)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"ERROR: cannot open {path}")  # src/params.f90:439-444
    raw = np.fromfile(path, dtype="<f4")
    delta4 = np.float32(raw[0])
    t_beg4 = np.float32(raw[5])
    it1 = _nint((float(t_start) - float(t_beg4)) / float(delta4)) + 1
    it2 = _nint((float(t_end) - float(t_beg4)) / float(delta4)) + 1
    nsmp = it2 - it1 + 1
    if nsmp > NPTS_MAX:
        raise ValueError("time window longer than npts_max = 2000 samples (src/params.f90:44)")
    lo = 158 + it1 - 1
    return raw[lo:lo + nsmp].astype(np.float64), float(delta4), nsmp


def read_obs(p: Params, verb: bool = False) -> Params:
    """subroutine read_obs (src/params.f90:422-476): fills p.obs (leading dimension
    npts_max like the reference's obs(npts_max, ntrc)), p.delta, p.nsmp."""
    p.obs = np.zeros((p.ntrc, NPTS_MAX))
    for i, f in enumerate(p.obs_files):
        path = f if os.path.isabs(f) else os.path.join(p.base_dir, f)
        data, delta, nsmp = read_sac(path, p.t_start, p.t_end)
        p.obs[i, :nsmp] = data
        p.delta, p.nsmp = delta, nsmp
        if verb:
            print("Finish reading", path)
    return p
