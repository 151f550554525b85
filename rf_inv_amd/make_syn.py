"""Synthetic-data generator in the reference's wire format: mirror of `program make_syn`
(src/make_syn.f90).

  write_sac        the direct-access records make_syn writes (src/make_syn.f90:121-137: delta @ record 1, b @ 6,
                   e @ 7, npts @ 80, plus the constant words 77, 86, 106; float32 samples from record 159;
                   little-endian, 4-byte records), i.e. exactly what read_obs (src/params.f90:422-476) reads back
  reference_noise  the "Add noise" block (src/make_syn.f90:80-115) draw for draw: sigma from ONE grnd(), nfft gauss()
                   values per series from the MT19937 stream in the reference's order, r2c -> times flt(:, itrc)
                   -> c2r (FFTW's unnormalised pair); when the rays are common ONE white series is drawn and the
                   loop of :90-96 is reproduced as written: trace 1 is that series through flt(:, 1), traces 2 ..
                   are trace 1's OUTPUT through their own filter (the loop re-reads noise(:, 1) after overwriting it)
  make_syn         traces of a given layer stack + that noise -> the two SAC files per trace
  make_syn_program the whole program: sgrnd(iseed), init_model, init_likelihood (their draws come first in the
                   stream), format_model of chain 1, test_vel, noise, SAC files

The forward model runs on the GPU engine, and so does the transform pair of the noise filter: FFTW's r2c / c2r plans
(src/fftw.f90:44-46) are librfgpu's rf_fft_r2c / rf_fft_c2r (include/rfgpu_ext.h) -- the same entry points the drop-in
Fortran `module fftw` executes, so `program make_syn` of the reference linked against the drop-in modules and this
mirror write byte-identical files.
File names: the reference formats them with `'(A10,I2.2,A2)'` from the 11-character literal "test_trace."
(src/make_syn.f90:120,140); A10 keeps the leftmost ten characters, so the files it actually creates are
`test_traceNNwn` and `test_traceNN` -- no dot.  `dotted=True` writes the names the literal suggests instead.
"""
from __future__ import annotations

import os

import numpy as np

from .mt19937 import MT19937
from .params import Params


def write_sac(path: str, samples, delta: float, t_start: float, t_end: float):
    """One trace in the layout of src/make_syn.f90:121-137 (records are 1-based, 4 bytes)."""
    x = np.asarray(samples, dtype=np.float64)
    nsmp = x.size
    rec = np.zeros(158 + nsmp, dtype="<f4")
    ints = rec.view("<i4")
    rec[0] = np.float32(delta)        # rec 1
    rec[5] = np.float32(t_start)      # rec 6
    rec[6] = np.float32(t_end)        # rec 7
    ints[76] = 6                      # rec 77
    ints[85] = 1                      # rec 86
    ints[79] = nsmp                   # rec 80
    ints[105] = 1                     # rec 106
    rec[158:] = x.astype(np.float32)  # rec 159...
    rec.tofile(path)


def _filter_series(white, flt_col, nfft: int, plans=None):
    """rx = white; dfftw_execute(ifft2) [r2c]; cx(1:nh) *= flt(1:nh, itrc); dfftw_execute(ifft) [c2r]
    (src/make_syn.f90:91-95 / :108-112).  Both FFTW transforms are unnormalised.  plans: (r2c(x), c2r(spec, n)), default
    librfgpu's (GPU; there is no host fall-back -- the CPU tests of the draw order hand in a host pair)."""
    if plans is None:
        from .engine import fft_c2r, fft_r2c

        plans = (fft_r2c, fft_c2r)
    spec = plans[0](white)                                 # r2c: sum_j x_j exp(-2 pi i j k / n)
    return plans[1](spec * flt_col, nfft)                  # c2r: unnormalised, Im of DC / Nyquist ignored


def reference_noise(rng: MT19937, p: Params, flt, is_ray_common: bool, plans=None):
    """src/make_syn.f90:80-115.  flt[nh, ntrc] (`flt` of module forward).  Returns
    (noise[nfft, ntrc] filtered, noise_sigma[ntrc], white[nfft, ntrc or 1] before the filter).
    plans: see _filter_series."""
    from .mcmc import gauss

    nfft, ntrc = p.nfft, p.ntrc
    noise = np.zeros((nfft, ntrc))
    sigma = np.zeros(ntrc)
    if is_ray_common:
        sigma[0] = rng.grnd() * (p.sig_max[0] - p.sig_min[0]) + p.sig_min[0]            # :85
        white = np.array([gauss(rng) * sigma[0] for _ in range(nfft)])[:, None]         # :86-88
        # :90-96, literally: the loop reads its input from noise(:, 1) and writes trace itrc's result to
        # noise(:, itrc) -- so iteration 1 REPLACES the white series by its own filtered (unnormalised: x nfft) output,
        # and traces 2 .. ntrc are that already filtered column through their own filter: flt_t(flt_1(white)) x nfft^2
        noise[:, 0] = white[:, 0]
        for t in range(ntrc):
            noise[:, t] = _filter_series(noise[:, 0], flt[:, t], nfft, plans)
        sigma[1:] = sigma[0]          # (the reference reports and uses noise_sigma(1) only)
    else:
        white = np.zeros((nfft, ntrc))
        for t in range(ntrc):
            sigma[t] = rng.grnd() * (p.sig_max[t] - p.sig_min[t]) + p.sig_min[t]        # :102-103
            white[:, t] = [gauss(rng) * sigma[t] for _ in range(nfft)]                  # :104-106
            noise[:, t] = _filter_series(white[:, t], flt[:, t], nfft, plans)           # :108-112
    return noise, sigma, white


def _names(itrc: int, dotted: bool):
    stem = f"test_trace.{itrc:02d}" if dotted else f"test_trace{itrc:02d}"
    return stem, stem + "wn"


def _write_pair(p: Params, out_dir: str, rft, noise, dotted: bool):
    os.makedirs(out_dir, exist_ok=True)
    for t in range(p.ntrc):
        clean_name, noisy_name = _names(t + 1, dotted)
        clean = rft[:p.nsmp, t]
        write_sac(os.path.join(out_dir, noisy_name), clean + noise[:p.nsmp, t], p.delta, p.t_start, p.t_end)   # :117-137
        write_sac(os.path.join(out_dir, clean_name), clean, p.delta, p.t_start, p.t_end)                        # :139-158


def make_syn(p: Params, engine, stack, out_dir: str, rng: MT19937 | None = None, seed: int | None = None,
             dotted: bool = False):
    """Traces of the layer stack (alpha, beta, rho, h) through the GPU engine + the reference's noise block on the
    stream `rng` (default: a fresh MT19937 seeded with `seed`, or p.iseed).  Writes test_traceNN (noise-free) and
    test_traceNNwn for every trace.  Returns dict(rft[nfft, ntrc], noise, noise_sigma, white)."""
    alpha, beta, rho, h = stack
    rft = engine.calc_rf(len(alpha), alpha, beta, rho, h)
    if rng is None:
        rng = MT19937(p.iseed if seed is None else seed)
    noise, sigma, white = reference_noise(rng, p, engine.flt, engine.is_ray_common)
    _write_pair(p, out_dir, rft, noise, dotted)
    return {"rft": rft, "noise": noise, "noise_sigma": sigma, "white": white}


def make_syn_program(p: Params, ref, engine, out_dir: str, dotted: bool = False):
    """`program make_syn` (src/make_syn.f90:44-160) from sgrnd(iseed) on: init_model and init_likelihood consume
    the stream first (random initial models of all nchains chains; sigma draws where sigma is solved), chain 1's
    model becomes the "true" model (format_model, :66-67; written to test_vel, :69-77), then the noise block and the
    SAC files of chain 1's traces.  `engine`: an RFEngine for p with max_walkers >= p.nchains."""
    from .mcmc import EngineEvaluator, RJMCMC
    from .model import format_model

    rng = MT19937(p.iseed)                                                  # :53
    s = RJMCMC(p, ref, EngineEvaluator(engine, p.k_max + 2), rng)
    s.init_model()                                                          # :57
    s.init_likelihood()                                                     # :65 (init_sig, init_rft)
    nlay, alpha, beta, rho, h, _ = format_model(p, ref, int(s.k[0]), s.z[0], s.dvp[0], s.dvs[0])   # :66-67
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "test_vel"), "w") as fh:                # :69-77 (list-directed reals)
        for i in range(nlay):
            fh.write(f" {float(alpha[i])!r} {float(beta[i])!r} {float(rho[i])!r} {float(h[i])!r}\n")
    rft = engine.get_rft(0, which=0)                                        # rft(:, :, 1)
    noise, sigma, white = reference_noise(rng, p, engine.flt, engine.is_ray_common)
    _write_pair(p, out_dir, rft, noise, dotted)
    return {"rft": rft, "noise": noise, "noise_sigma": sigma, "white": white, "stack": (alpha, beta, rho, h),
            "k": int(s.k[0]), "rng": rng}
