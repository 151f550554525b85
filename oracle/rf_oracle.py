"""ctypes front-end of the CPU oracle (oracle/rf_oracle.c) + the init-time pieces
that need LAPACK / file IO.

TEST INFRASTRUCTURE ONLY -- see the header of rf_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference lines restated here (all under /root/reference):
  build_r_inv      src/likelihood.f90:168-222 (init_r_inv, LAPACK dgesvd)
  read_sac         src/params.f90:422-476     (read_obs)
  calc_seis_numpy  src/forward.f90:212-287    (independent numpy complex128
                   restatement of calc_seis used to cross-check the C one)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librf_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    """Compile librf_oracle.so with the committed Makefile (gcc)."""
    src = os.path.join(_HERE, "rf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "librf_oracle.so"])
    return _LIB_PATH


def _typed(path):
    h = C.CDLL(path)
    h.rfo_direct_arrival.restype = C.c_double
    h.rfo_log_likelihood.restype = C.c_double
    h.rfo_vp_to_rho.restype = C.c_double
    h.rfo_vp_to_rho.argtypes = [C.c_double]
    h.rfo_max_threads.restype = C.c_int
    return h


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _typed(_LIB_PATH)
    return _lib


FAST_FLAGS = ["-O3", "-march=native", "-fno-math-errno", "-fno-trapping-math", "-ffp-contract=off", "-fno-fast-math"]
_fast = None


def lib_fast():
    """The same source compiled for speed on THIS host (bench.py's cpu_baseline leg only): -O3 -march=native,
    still no FMA contraction and no fast-math, so it returns bit-identical values (tests/test_oracle_kat.py);
    built into the temp directory because -march=native must match the machine it runs on."""
    global _fast
    if _fast is None:
        import hashlib
        import tempfile

        src = os.path.join(_HERE, "rf_oracle.c")
        tag = hashlib.sha256(open(src, "rb").read() + " ".join(FAST_FLAGS).encode()).hexdigest()[:16]
        out = os.path.join(tempfile.gettempdir(), f"librf_oracle_fast_{os.getuid()}_{tag}.so")
        if not os.path.exists(out):
            tmp = out + f".{os.getpid()}.tmp"
            subprocess.check_call(["gcc", *FAST_FLAGS, "-fPIC", "-std=c99", "-fopenmp", "-shared", "-o", tmp, src, "-lm"])
            os.replace(tmp, out)
        _fast = _typed(out)
    return _fast


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


# ----------------------------------------------------------------------------
def init_filter(nfft, delta, a_gus):
    """flt[itrc, :nh]  (src/forward.f90:95-119)."""
    a_gus, pa = _d(a_gus)
    ntrc = a_gus.size
    nh = nfft // 2 + 1
    flt = np.empty((ntrc, nh))
    lib().rfo_init_filter(C.c_int(nfft), C.c_int(ntrc), C.c_double(delta), pa,
                          flt.ctypes.data_as(_dp))
    return flt


def calc_seis(nfft, delta, rayp, ipha, alpha, beta, rho, h):
    """(ur_freq, uz_freq) complex128[nh]  (src/forward.f90:212-287)."""
    alpha, pa = _d(alpha); beta, pb = _d(beta); rho, pr = _d(rho); h, ph = _d(h)
    nh = nfft // 2 + 1
    ur = np.empty(nh, dtype=np.complex128)
    uz = np.empty(nh, dtype=np.complex128)
    lib().rfo_calc_seis(C.c_int(alpha.size), C.c_int(nfft), C.c_double(delta),
                        C.c_double(rayp), C.c_int(ipha), pa, pb, pr, ph,
                        ur.ctypes.data_as(C.c_void_p), uz.ctypes.data_as(C.c_void_p))
    return ur, uz


def direct_arrival(h, v, rayp, sdep):
    h, ph = _d(h); v, pv = _d(v)
    return lib().rfo_direct_arrival(C.c_int(h.size), ph, pv, C.c_double(rayp), C.c_double(sdep))


def c2r(cx, n, naive=False):
    cx = np.ascontiguousarray(cx, dtype=np.complex128)
    rx = np.empty(n)
    f = lib().rfo_c2r_naive if naive else lib().rfo_c2r
    f(C.c_int(n), cx.ctypes.data_as(C.c_void_p), rx.ctypes.data_as(_dp))
    return rx


def calc_rf(cfg, alpha, beta, rho, h, want_stages=False):
    """rft[itrc, :nfft]  (src/forward.f90:123-208).

    cfg: dict(nfft, deconv_mode, delta, t_start, sdep, rayps, a_gus, ipha).
    With want_stages also returns (npre[ntrc], rff[ntrc, nh], freq_v[ntrc, nh]).
    """
    rayps, prp = _d(cfg["rayps"]); a_gus, pag = _d(cfg["a_gus"]); ipha, pip = _i(cfg["ipha"])
    alpha, pa = _d(alpha); beta, pb = _d(beta); rho, pr = _d(rho); h, ph = _d(h)
    ntrc, nfft = rayps.size, int(cfg["nfft"])
    nh = nfft // 2 + 1
    rft = np.empty((ntrc, nfft))
    npre = np.zeros(ntrc, dtype=np.int32)
    spec = np.empty((ntrc, 2, nh), dtype=np.complex128)
    lib().rfo_calc_rf(C.c_int(nfft), C.c_int(ntrc), C.c_int(int(cfg["deconv_mode"])),
                      C.c_double(cfg["delta"]), C.c_double(cfg["t_start"]),
                      C.c_double(cfg["sdep"]), prp, pag, pip, C.c_int(alpha.size),
                      pa, pb, pr, ph, rft.ctypes.data_as(_dp), npre.ctypes.data_as(_ip),
                      spec.ctypes.data_as(C.c_void_p))
    if want_stages:
        return rft, npre, spec[:, 0, :].copy(), spec[:, 1, :].copy()
    return rft


def log_likelihood(rft, obs, r_inv, sig, nsmp):
    """src/likelihood.f90:84-98.  rft[ntrc, nfft], obs[ntrc, >=nsmp],
    r_inv[ntrc, nsmp, nsmp] stored so that r_inv[t, j, i] = Fortran r_inv(i, j, t)."""
    rft, prf = _d(rft); obs, po = _d(obs); r_inv, pri = _d(r_inv); sig, ps = _d(sig)
    ntrc, nfft = rft.shape
    return lib().rfo_log_likelihood(C.c_int(nfft), C.c_int(ntrc), C.c_int(nsmp), prf, po,
                                    C.c_int(obs.shape[1]), pri, ps)


def eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=False, nthreads=1, fast=False, want_kappa=False):
    """nb x calc_likelihood(fwd_flag=.true.)  (src/likelihood.f90:56-101).

    layers[nb, 4, nlay_pad] rows = alpha, beta, rho, h;  sig[nb, ntrc].
    Returns logL[nb] (and rft[nb, ntrc, nfft]).  fast: the speed build (lib_fast), same values.
    want_kappa: also kappa[nb] = max over traces of max|rx| / |maxval(rx)| of the filtered vertical trace each
    item is normalised by (the conditioning number of the tests' kappa rule; 1 with deconvolution)."""
    rayps, prp = _d(cfg["rayps"]); a_gus, pag = _d(cfg["a_gus"]); ipha, pip = _i(cfg["ipha"])
    obs, po = _d(obs); r_inv, pri = _d(r_inv); layers, pl = _d(layers); sig, ps = _d(sig)
    nlay, pn = _i(nlay)
    nb, _, nlay_pad = layers.shape
    ntrc, nfft = rayps.size, int(cfg["nfft"])
    logl = np.empty(nb)
    rft = np.empty((nb, ntrc, nfft)) if want_rft else None
    kappa = np.ones(nb) if want_kappa else None
    (lib_fast() if fast else lib()).rfo_eval_batch_kappa(C.c_int(nfft), C.c_int(ntrc), C.c_int(nsmp),
                         C.c_int(int(cfg["deconv_mode"])), C.c_double(cfg["delta"]),
                         C.c_double(cfg["t_start"]), C.c_double(cfg["sdep"]), prp, pag, pip,
                         po, C.c_int(obs.shape[1]), pri, C.c_int(nb), pn, C.c_int(nlay_pad),
                         pl, ps, logl.ctypes.data_as(_dp),
                         rft.ctypes.data_as(_dp) if want_rft else None, C.c_int(nthreads),
                         kappa.ctypes.data_as(_dp) if want_kappa else None)
    out = (logl,) + ((rft,) if want_rft else ()) + ((kappa,) if want_kappa else ())
    return out if len(out) > 1 else logl


def max_threads():
    return lib().rfo_max_threads()


def vp_to_rho(vp):
    return lib().rfo_vp_to_rho(C.c_double(vp))


class _ModelCfg(C.Structure):
    _fields_ = [("k_max", C.c_int), ("vp_mode", C.c_int), ("nref", C.c_int),
                ("sdep", C.c_double), ("z_max", C.c_double), ("h_min", C.c_double),
                ("z_ref_min", C.c_double), ("dz_ref", C.c_double),
                ("vp_min", C.c_double), ("vp_max", C.c_double), ("vs_min", C.c_double),
                ("vs_max", C.c_double), ("vpvs_min", C.c_double), ("vpvs_max", C.c_double),
                ("vp_ref", _dp), ("vs_ref", _dp)]


def format_model(mcfg, k, z, dvp, dvs):
    """src/model.f90:175-290.  mcfg: dict with the fields of _ModelCfg (vp_ref,
    vs_ref arrays).  Returns (nlay, alpha, beta, rho, h, is_valid)."""
    vp_ref, pvp = _d(mcfg["vp_ref"]); vs_ref, pvs = _d(mcfg["vs_ref"])
    m = _ModelCfg(int(mcfg["k_max"]), int(mcfg["vp_mode"]), vp_ref.size, mcfg["sdep"],
                  mcfg["z_max"], mcfg["h_min"], mcfg["z_ref_min"], mcfg["dz_ref"],
                  mcfg["vp_min"], mcfg["vp_max"], mcfg["vs_min"], mcfg["vs_max"],
                  mcfg["vpvs_min"], mcfg["vpvs_max"], pvp, pvs)
    kmax = int(mcfg["k_max"])
    z, pz = _d(np.resize(np.asarray(z, dtype=np.float64), max(kmax - 1, 1)) if len(z) != kmax - 1 else z)
    dvp, pdp = _d(dvp); dvs, pds = _d(dvs)
    assert dvp.size == kmax and dvs.size == kmax
    n = kmax + 2
    alpha = np.zeros(n); beta = np.zeros(n); rho = np.zeros(n); h = np.zeros(n)
    valid = C.c_int(0)
    lib().rfo_format_model.restype = C.c_int
    nlay = lib().rfo_format_model(C.byref(m), C.c_int(int(k)), pz, pdp, pds,
                                  alpha.ctypes.data_as(_dp), beta.ctypes.data_as(_dp),
                                  rho.ctypes.data_as(_dp), h.ctypes.data_as(_dp), C.byref(valid))
    return nlay, alpha[:nlay], beta[:nlay], rho[:nlay], h[:nlay], bool(valid.value)


# ----------------------------------------------------------------------------
def build_r_inv(nsmp, a_gus, delta, return_rank=False):
    """src/likelihood.f90:168-222.  Returns r_inv[ntrc, nsmp, nsmp] laid out so
    that r_inv[t].ravel() is Fortran's column-major r_inv(:, :, t), i.e.
    r_inv[t, j, i] == r_inv(i, j, t).  LAPACK dgesvd via scipy (the reference
    links an unpinned LAPACK, Makefile:19)."""
    from scipy.linalg import svd

    a_gus = np.atleast_1d(np.asarray(a_gus, dtype=np.float64))
    out = np.empty((a_gus.size, nsmp, nsmp))
    ranks = []
    idx = np.arange(nsmp)
    e2 = (idx[:, None] - idx[None, :]) ** 2
    for t, a in enumerate(a_gus):
        r = np.exp(-a ** 2 * delta ** 2)                       # :183
        r_mat = r ** e2.astype(np.float64)                     # :185-190  r ** ((i-j)**2)
        u, s, vt = svd(r_mat, full_matrices=True, lapack_driver="gesvd")  # :203
        dinv = np.where(s > 1.0e-3, 1.0 / np.where(s > 1.0e-3, s, 1.0), 0.0)  # :212-219
        ranks.append(int((s > 1.0e-3).sum()))
        m = (vt.T * dinv[None, :]) @ u.T                       # :221 (V*D)*U^T, m[i, j] = r_inv(i, j)
        out[t] = m.T                                           # store column-major
    return (out, ranks) if return_rank else out


def read_sac(path, t_start, t_end):
    """src/params.f90:422-476 read_obs for one file: returns (obs[nsmp], delta, nsmp).
    SAC: 4-byte records, delta @rec 1, b @rec 6, npts @rec 80, data from rec 159.
    t_start / t_end are real(8) (:66), delta4 / t_beg4 default REAL: the mixed expressions of
    :449-450 are evaluated in double on the float32-valued header fields."""
    raw = np.fromfile(path, dtype="<f4")
    delta4 = np.float32(raw[0])
    t_beg4 = np.float32(raw[5])
    it1 = int(_nint((float(t_start) - float(t_beg4)) / float(delta4))) + 1   # :449
    it2 = int(_nint((float(t_end) - float(t_beg4)) / float(delta4))) + 1     # :450
    nsmp = it2 - it1 + 1
    data = raw[158 + it1 - 1: 158 + it1 - 1 + nsmp].astype(np.float64)  # :455-458
    return data, float(delta4), nsmp


def _nint(x):
    return np.floor(x + 0.5) if x >= 0 else -np.floor(0.5 - x)


# ----------------------------------------------------------------------------
def calc_seis_numpy(nfft, delta, rayp, ipha, alpha, beta, rho, h):
    """Independent numpy complex128 restatement of src/forward.f90:212-287 with
    literal dense 4x4 complex matrices (np.matmul), used only to cross-check the
    C restatement (tests/test_oracle_kat.py)."""
    alpha = np.asarray(alpha, float); beta = np.asarray(beta, float)
    rho = np.asarray(rho, float); h = np.asarray(h, float)
    nlay = alpha.size
    nh = nfft // 2 + 1
    pi = 3.1415926535897931
    sea = beta[0] < 0
    ilay0 = 1 if sea else 0
    omg = np.arange(nh) * (2.0 * pi / (nfft * delta))
    omg[0] = float(np.float32(1.0e-5))
    p = rayp

    def e_inv(om, ro, al, be):
        eta = np.sqrt(1 / (be * be) - p * p); xi = np.sqrt(1 / (al * al) - p * p)
        bp = 1 - 2 * be * be * p * p
        e = np.zeros((nh, 4, 4), complex)
        e[:, 0, 0] = be * be * p / al
        e[:, 0, 1] = bp / (2 * al * xi)
        e[:, 0, 2] = -p / (2 * om * ro * al * xi) * 1j
        e[:, 0, 3] = -1 / (2 * om * ro * al) * 1j
        e[:, 1, 0] = bp / (2 * be * eta)
        e[:, 1, 1] = -be * p
        e[:, 1, 2] = -1 / (2 * om * ro * be) * 1j
        e[:, 1, 3] = p / (2 * om * ro * be * eta) * 1j
        e[:, 2, 0] = e[:, 0, 0]; e[:, 2, 1] = -e[:, 0, 1]; e[:, 2, 2] = -e[:, 0, 2]; e[:, 2, 3] = e[:, 0, 3]
        e[:, 3, 0] = e[:, 1, 0]; e[:, 3, 1] = -e[:, 1, 1]; e[:, 3, 2] = -e[:, 1, 2]; e[:, 3, 3] = e[:, 1, 3]
        return e

    def p_sol(om, ro, al, be, z):
        b2 = be * be; p2 = p * p; bp = 1 - 2 * b2 * p2
        eta = np.sqrt(1 / b2 - p2); xi = np.sqrt(1 / (al * al) - p2)
        cx, ce = np.cos(om * xi * z), np.cos(om * eta * z)
        sx, se = np.sin(om * xi * z), np.sin(om * eta * z)
        m = np.zeros((nh, 4, 4), complex)
        m[:, 0, 0] = 2 * b2 * p2 * cx + bp * ce
        m[:, 1, 0] = p * (2 * b2 * xi * sx - bp / eta * se) * 1j
        m[:, 2, 0] = om * ro * (-4 * b2 * b2 * p2 * xi * sx - bp * bp / eta * se)
        m[:, 3, 0] = 2 * om * b2 * ro * p * bp * (cx - ce) * 1j
        m[:, 0, 1] = p * (bp / xi * sx - 2 * b2 * eta * se) * 1j
        m[:, 1, 1] = bp * cx + 2 * b2 * p2 * ce
        m[:, 2, 1] = m[:, 3, 0]
        m[:, 3, 1] = -om * ro * (bp * bp / xi * sx + 4 * b2 * b2 * p2 * eta * se)
        m[:, 0, 2] = (p2 / xi * sx + eta * se) / (om * ro)
        m[:, 1, 2] = p * (-cx + ce) / (om * ro) * 1j
        m[:, 2, 2] = m[:, 0, 0]; m[:, 3, 2] = m[:, 0, 1]; m[:, 0, 3] = m[:, 1, 2]
        m[:, 1, 3] = (xi * sx + p2 / eta * se) / (om * ro)
        m[:, 2, 3] = m[:, 1, 0]; m[:, 3, 3] = m[:, 1, 1]
        return m

    prod = np.tile(np.eye(4, dtype=complex), (nh, 1, 1))
    for il in range(ilay0, nlay - 1):
        prod = np.matmul(p_sol(omg, rho[il], alpha[il], beta[il], h[il]), prod)
    sl = np.matmul(e_inv(omg, rho[-1], alpha[-1], beta[-1]), prod)
    s = lambda i, j: sl[:, i - 1, j - 1]
    if not sea:
        den = s(3, 1) * s(4, 2) - s(3, 2) * s(4, 1)
        if ipha >= 0:
            return s(4, 2) / den, -s(4, 1) / den
        return -s(3, 2) / den, s(3, 1) / den
    xi = np.sqrt(1 / (alpha[0] ** 2) - p * p)
    g = rho[0] * omg / xi
    l11 = np.cos(omg * xi * h[0]); l21 = -g * np.sin(omg * xi * h[0])
    a = s(4, 2) * l11 + s(4, 4) * l21
    b = s(3, 2) * l11 + s(3, 4) * l21
    if ipha >= 0:
        return a / (a * s(3, 1) - b * s(4, 1)), l11 * s(4, 1) / (b * s(4, 1) - a * s(3, 1))
    return -b / (a * s(3, 1) - b * s(4, 1)), -l11 * s(3, 1) / (b * s(4, 1) - a * s(3, 1))
