"""How far are the GPU result and the CPU oracle (= the reference's double-precision arithmetic)
from the exactly evaluated formulas?  Test tooling (uses oracle/): evaluates a sample of the bench
walkers three ways --

  truth  : numpy 80-bit long double for every operation after the inputs all implementations share
           (the layer stack, omega, and the reference's double-rounded phase arguments
           (omega*xi)*z, whose rounding is part of the reference result), O(n^2) inverse DFT;
  oracle : oracle/rf_oracle.c (the reference's arithmetic in double);
  gpu    : the in-tree librfgpu.so, or the build named by --lib.

and prints the distribution of |logL - truth| / |truth| for oracle and gpu, and |gpu - oracle|.

    python tests/tools/truth_check.py --workload c4 --n 24
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LD, CLD = np.longdouble, np.clongdouble


def spectra_ld(nfft, delta, p, ipha, alpha, beta, rho, h):
    """src/forward.f90:212-287 in long double (land and ocean), literal complex 4x4 products."""
    nlay = len(alpha)
    nh = nfft // 2 + 1
    pi = 3.1415926535897931
    sea = beta[0] < 0
    ilay0 = 1 if sea else 0
    omg_d = np.arange(nh) * (2.0 * pi / (nfft * delta))
    omg_d[0] = float(np.float32(1.0e-5))
    om = omg_d.astype(LD)
    p_l = LD(p)
    one = LD(1)

    def slow(v):
        return np.sqrt(one / (LD(v) * LD(v)) - p_l * p_l)

    def args(v, z):   # the reference's double-rounded argument (forward.f90:397-400)
        sd = np.sqrt(1.0 / (v * v) - p * p)
        return ((omg_d * sd) * z).astype(LD)

    def p_sol(ro, al, be, z):
        ro, b2 = LD(ro), LD(be) * LD(be)
        p2 = p_l * p_l
        bp = one - 2 * b2 * p2
        eta, xi = slow(be), slow(al)
        ax, ae = args(al, z), args(be, z)
        cx, ce, sx, se = np.cos(ax), np.cos(ae), np.sin(ax), np.sin(ae)
        m = np.zeros((nh, 4, 4), CLD)
        j = CLD(1j)
        m[:, 0, 0] = 2 * b2 * p2 * cx + bp * ce
        m[:, 1, 0] = p_l * (2 * b2 * xi * sx - bp / eta * se) * j
        m[:, 2, 0] = om * ro * (-4 * b2 * b2 * p2 * xi * sx - bp * bp / eta * se)
        m[:, 3, 0] = 2 * om * b2 * ro * p_l * bp * (cx - ce) * j
        m[:, 0, 1] = p_l * (bp / xi * sx - 2 * b2 * eta * se) * j
        m[:, 1, 1] = bp * cx + 2 * b2 * p2 * ce
        m[:, 2, 1] = m[:, 3, 0]
        m[:, 3, 1] = -om * ro * (bp * bp / xi * sx + 4 * b2 * b2 * p2 * eta * se)
        m[:, 0, 2] = (p2 / xi * sx + eta * se) / (om * ro)
        m[:, 1, 2] = p_l * (-cx + ce) / (om * ro) * j
        m[:, 2, 2] = m[:, 0, 0]; m[:, 3, 2] = m[:, 0, 1]; m[:, 0, 3] = m[:, 1, 2]
        m[:, 1, 3] = (xi * sx + p2 / eta * se) / (om * ro)
        m[:, 2, 3] = m[:, 1, 0]; m[:, 3, 3] = m[:, 1, 1]
        return m

    def e_inv(ro, al, be):
        ro, al_l, be_l = LD(ro), LD(al), LD(be)
        eta, xi = slow(be), slow(al)
        bp = one - 2 * be_l * be_l * p_l * p_l
        e = np.zeros((nh, 4, 4), CLD)
        j = CLD(1j)
        e[:, 0, 0] = be_l * be_l * p_l / al_l
        e[:, 0, 1] = bp / (2 * al_l * xi)
        e[:, 0, 2] = -p_l / (2 * om * ro * al_l * xi) * j
        e[:, 0, 3] = -one / (2 * om * ro * al_l) * j
        e[:, 1, 0] = bp / (2 * be_l * eta)
        e[:, 1, 1] = -be_l * p_l
        e[:, 1, 2] = -one / (2 * om * ro * be_l) * j
        e[:, 1, 3] = p_l / (2 * om * ro * be_l * eta) * j
        e[:, 2, 0] = e[:, 0, 0]; e[:, 2, 1] = -e[:, 0, 1]; e[:, 2, 2] = -e[:, 0, 2]; e[:, 2, 3] = e[:, 0, 3]
        e[:, 3, 0] = e[:, 1, 0]; e[:, 3, 1] = -e[:, 1, 1]; e[:, 3, 2] = -e[:, 1, 2]; e[:, 3, 3] = e[:, 1, 3]
        return e

    prod = np.tile(np.eye(4, dtype=CLD), (nh, 1, 1))
    for il in range(ilay0, nlay - 1):
        prod = np.einsum("kij,kjl->kil", p_sol(rho[il], alpha[il], beta[il], h[il]), prod)
    sl = np.einsum("kij,kjl->kil", e_inv(rho[-1], alpha[-1], beta[-1]), prod)
    s = lambda i, jx: sl[:, i - 1, jx - 1]
    if not sea:
        den = s(3, 1) * s(4, 2) - s(3, 2) * s(4, 1)
        ur, uz = (s(4, 2) / den, -s(4, 1) / den) if ipha >= 0 else (-s(3, 2) / den, s(3, 1) / den)
    else:
        xi = slow(alpha[0])
        aw = args(alpha[0], h[0])
        l11 = np.cos(aw)
        l21 = -(LD(rho[0]) * om / xi) * np.sin(aw)
        a = s(4, 2) * l11 + s(4, 4) * l21
        b = s(3, 2) * l11 + s(3, 4) * l21
        if ipha >= 0:
            ur, uz = a / (a * s(3, 1) - b * s(4, 1)), l11 * s(4, 1) / (b * s(4, 1) - a * s(3, 1))
        else:
            ur, uz = -b / (a * s(3, 1) - b * s(4, 1)), -l11 * s(3, 1) / (b * s(4, 1) - a * s(3, 1))
    return np.conj(ur), -np.conj(uz)          # forward.f90:145-146


def c2r_ld(cx, n, which=None):
    """Unnormalised inverse real DFT from the half spectrum (FFTW c2r: imaginary parts of the DC and
    Nyquist bins dropped), long double, O(n^2); `which`: output indices wanted (default all)."""
    nh = n // 2 + 1
    idx = np.arange(n) if which is None else np.asarray(which)
    k = np.arange(1, nh - 1)
    out = np.zeros(idx.size, LD)
    two_pi = 2 * np.arctan2(LD(0), LD(-1))
    for c0 in range(0, idx.size, 256):
        j = idx[c0:c0 + 256]
        ang = (two_pi / n) * ((j[:, None] * k[None, :]) % n).astype(LD)
        acc = 2 * (np.cos(ang) @ cx[1:nh - 1].real - np.sin(ang) @ cx[1:nh - 1].imag)
        acc += cx[0].real + cx[nh - 1].real * np.where(j % 2 == 0, LD(1), LD(-1))
        out[c0:c0 + 256] = acc
    return out


def logl_truth(p, flt, obs, r_inv, nlay, lay, sig, oracle):
    """calc_rf + calc_likelihood (deconv_mode 0) in long double."""
    n, nsmp = p.nfft, p.nsmp
    a, b, r, h = (lay[i, :nlay] for i in range(4))
    ll = LD(0)
    for t in range(p.ntrc):
        ipha = int(p.ipha[t])
        fr, fv = spectra_ld(n, p.delta, float(p.rayps[t]), ipha, a, b, r, h)
        rff = fr if ipha == 1 else fv
        tp = oracle.direct_arrival(h, a if ipha == 1 else b, float(p.rayps[t]), p.sdep)
        f = flt[t].astype(LD)
        i = np.arange(1, nsmp + 1)
        if ipha == 1:
            npre = int(np.floor((-p.t_start - tp) / p.delta + 0.5))
            j = (n - npre + i) % n
            j[j == 0] = n
            tr = c2r_ld(rff * f, n, j - 1)
        else:
            npre = int(np.floor((-p.t_start + tp) / p.delta + 0.5))
            j = (n + npre - i + 1) % n
            j[j == 0] = n
            tr = -c2r_ld(rff * f, n, j - 1)
        tr = tr / c2r_ld(fv * f, n).max()
        mis = tr - obs[t, :nsmp].astype(LD)
        rinv = r_inv[t].astype(LD)                       # [j, i] == r_inv(i, j)
        phi = mis @ (rinv @ mis)
        s = LD(sig[t])
        ll += -LD(0.5) * phi / (s * s) - LD(nsmp) * np.log(s)
    return ll


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--lib", default=None, help="another build of librfgpu.so")
    args = ap.parse_args()
    import bench
    from oracle import rf_oracle as oracle
    from rf_inv_amd import _lib

    _lib.load(args.lib)
    from rf_inv_amd import RFEngine, format_model, read_ref_model
    from rf_inv_amd.likelihood import init_r_inv

    oracle.build()
    w = dict(bench.WORKLOADS[args.workload])
    p = bench.make_params(w)
    ref = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    nb = min(w["walkers"], 1024)
    nlay, layers = bench.draw_walkers(p, ref, 0, nb)
    sig = np.full((nb, p.ntrc), 0.01)
    r_inv = init_r_inv(p.nsmp, p.a_gus, p.delta)
    zt = np.zeros(max(p.k_max - 1, 1)); dvt = np.zeros(p.k_max); dst = np.zeros(p.k_max)
    zt[:3] = [3.1 + p.sdep, 7.7 + p.sdep, 14.2 + p.sdep]; dst[:3] = [-0.6, 0.2, 0.5]; dst[p.k_max - 1] = 0.9
    nl_t, a_t, b_t, r_t, h_t, ok = format_model(p, ref, 3, zt, dvt, dst)
    kw = dict(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep,
              rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha, nsmp=p.nsmp, nlay_max=p.k_max + 2)
    with RFEngine(obs=np.zeros((p.ntrc, p.nsmp)), r_inv=r_inv, max_walkers=1, **kw) as e0:
        obs = np.ascontiguousarray(e0.calc_rf(nl_t, a_t, b_t, r_t, h_t)[:p.nsmp].T)
    with RFEngine(obs=obs, r_inv=r_inv, max_walkers=nb, **kw) as eng:
        ll_gpu = eng.eval_batch(np.arange(nb, dtype=np.int32), nlay, layers, sig)
    cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep,
               rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha)
    ll_cpu = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, p.nsmp, nthreads=oracle.max_threads())
    d = np.abs(ll_gpu - ll_cpu) / np.abs(ll_cpu)
    # sample: the walkers where gpu and oracle differ most, plus a spread over layer counts
    pick = list(np.argsort(-d)[:args.n // 2]) + list(np.argsort(nlay)[::max(1, nb // (args.n - args.n // 2))][:args.n - args.n // 2])
    flt = oracle.init_filter(p.nfft, p.delta, p.a_gus)
    rows = []
    for i in pick:
        t = logl_truth(p, flt, obs, r_inv, int(nlay[i]), layers[i], sig[i], oracle)
        eg, ec = abs(LD(ll_gpu[i]) - t) / abs(t), abs(LD(ll_cpu[i]) - t) / abs(t)
        rows.append((int(i), int(nlay[i]), float(t), float(eg), float(ec), float(d[i])))
        print("walker %5d nlay %2d logL %.6e  |gpu-truth| %.2e  |oracle-truth| %.2e  |gpu-oracle| %.2e (relative)"
              % rows[-1], flush=True)
    r = np.array(rows)
    print("lib", args.lib or "in-tree", args.workload, "n", len(rows))
    print("max / median rel. error vs truth:  gpu %.2e / %.2e   oracle %.2e / %.2e   gpu-vs-oracle (all %d walkers) max %.2e"
          % (r[:, 3].max(), np.median(r[:, 3]), r[:, 4].max(), np.median(r[:, 4]), nb, d.max()))


if __name__ == "__main__":
    main()
