"""Shared synthetic-input builders for the parity tests (seeded, deterministic)."""
import os

import numpy as np

DELTA = float(np.float32(0.05))


def load_true_model(golden_dir):
    return np.loadtxt(os.path.join(golden_dir, "sample_syn", "true", "true.velmod")).T


def random_stack(rng, nlay, ocean=False, sdep=2.0):
    """A physically valid layer stack (alpha, beta, rho, h); ocean prepends the
    reference's water layer (model.f90:203-206)."""
    alpha = rng.uniform(4.0, 7.5, nlay)
    beta = alpha / rng.uniform(1.6, 1.9, nlay)
    rho = 0.77 + 0.32 * alpha
    h = rng.uniform(0.3, 4.0, nlay)
    h[-1] = 999.0
    if ocean:
        alpha[0], beta[0], rho[0], h[0] = 1.5, -999.0, 1.0, sdep
    return alpha, beta, rho, h


def pack_layers(stacks, nlay_pad):
    """list of (alpha, beta, rho, h) -> (nlay[nb], layers[nb, 4, nlay_pad])."""
    nb = len(stacks)
    layers = np.ones((nb, 4, nlay_pad))
    nlay = np.zeros(nb, dtype=np.int32)
    for i, st in enumerate(stacks):
        n = len(st[0])
        nlay[i] = n
        for r in range(4):
            layers[i, r, :n] = st[r]
    return nlay, layers


def make_cfg(nfft=256, deconv_mode=0, t_start=0.0, sdep=0.0, rayps=(0.06,), a_gus=None, ipha=None):
    rayps = np.asarray(rayps, dtype=np.float64)
    n = rayps.size
    return dict(nfft=nfft, deconv_mode=deconv_mode, delta=DELTA, t_start=t_start, sdep=sdep,
                rayps=rayps, a_gus=np.full(n, 4.0) if a_gus is None else np.asarray(a_gus, float),
                ipha=np.ones(n, dtype=np.int32) if ipha is None else np.asarray(ipha, dtype=np.int32))


def synth_obs(oracle, cfg, stack, nsmp):
    """Noise-free observed traces of a fixed model, through the oracle."""
    rft = oracle.calc_rf(cfg, *stack)
    return np.ascontiguousarray(rft[:, :nsmp])


def logl_tol(ref):
    """|dlogL| bound: 1e-9 absolute (north star, stated at sigma = 0.01 and |logL| <~ 1e3),
    relaxed to 1e-12 relative for large |logL|."""
    return np.maximum(1e-9, 1e-12 * np.abs(ref))


KAPPA_MIN, KAPPA_SCALE = 1000.0, 1000.0      # backed by profiles/r06_kappa_reference_spread.json, see assert_logl_parity


def assert_logl_parity(got, ref, kappa, what=""):
    """|dlogL| <= logl_tol for every item; an item may exceed it only under the conditioning rule of
    tests/test_gpu_configs.py: kappa = max|rx| / |maxval(rx)| of the oracle's own vertical trace
    >= 1000, and then within tolerance * kappa / 1000.  Items below kappa = 1000 get no allowance.
    Why that scale (round 6, profiles/r06_kappa_reference_spread.json: all 103 items that needed an allowance in three
    randomised sweeps, re-evaluated on the CPU): the divisor maxval(rx) carries an absolute rounding error of c * 1.1e-16 *
    max|rx|, i.e. a relative error c * 1.1e-16 * kappa, twice that in logL = 2.2e-4 * c * kappa in units of the 1e-12
    relative tolerance.  Measured: the HIP path against the reference's own -O0 build <= 2.7e-4 kappa; the oracle with its
    FFT against the oracle with the O(n^2) definition of the SAME transform <= 2.6e-4 kappa (86 of the 103 items beyond the
    plain tolerance: two correct inverse transforms -- the reference links an unpinned FFTW -- already disagree by that
    much); two builds of the reference itself (-O0 / -O2, same MKL transform) <= 1.5e-5 kappa (10 items beyond the plain
    tolerance).  kappa / 1000 leaves c <= 4.5; rounds 3-5 used kappa / 10, a hundred times looser than anything seen."""
    got, ref, kappa = np.asarray(got), np.asarray(ref), np.asarray(kappa)
    d = np.abs(got - ref)
    tol = logl_tol(ref)
    for i in np.nonzero(~(d <= tol))[0]:
        assert kappa[i] >= KAPPA_MIN, (what, int(i), "well-conditioned item off tolerance", got[i], ref[i], kappa[i])
        assert d[i] <= tol[i] * kappa[i] / KAPPA_SCALE, (what, int(i), got[i], ref[i], kappa[i])


def mkl_fftw3_c2r(spec, nfft):
    """The c2r plan of the reference's CPU build (oracle/Makefile.ref): FFTW3's Fortran interface as the image's Intel MKL
    exports it -- dfftw_plan_dft_c2r_1d_ / dfftw_execute_, the calls of src/fftw.f90:44 and src/forward.f90:172 -- on
    cx(1:nfft) = (spec, 0 ...).  None when that MKL is not there."""
    import ctypes as C

    path = "/opt/conda/lib/libmkl_rt.so"
    if not os.path.exists(path):
        return None
    mkl = C.CDLL(path)
    cx = np.zeros(nfft, dtype=np.complex128)
    cx[:len(spec)] = spec
    rx = np.zeros(nfft)
    plan, n, flags = C.c_int64(0), C.c_int(nfft), C.c_int(64)          # FFTW_ESTIMATE
    mkl.dfftw_plan_dft_c2r_1d_(C.byref(plan), C.byref(n), cx.ctypes.data_as(C.c_void_p), rx.ctypes.data_as(C.c_void_p),
                               C.byref(flags))
    mkl.dfftw_execute_(C.byref(plan))
    return rx
