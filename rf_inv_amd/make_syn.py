"""Synthetic-data generator in the reference's wire format: mirror of src/make_syn.f90.

`write_sac` reproduces the direct-access records make_syn writes (src/make_syn.f90:121-137:
delta @ record 1, b @ 6, e @ 7, npts @ 80, plus the constant words 77, 86, 106; float32
samples from record 159; little-endian, 4-byte records), i.e. exactly what read_obs
(src/params.f90:422-476) reads back.  `make_syn` runs the forward model of a given or random
model through the GPU engine and adds Gaussian noise filtered like the reference's
(r2c -> flt -> c2r, src/make_syn.f90:91-95, with numpy's FFT standing in for FFTW on the host).
"""
from __future__ import annotations

import os

import numpy as np

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


def filtered_noise(rng, nfft: int, flt_col, sigma: float):
    """src/make_syn.f90:88-95: white Gaussian noise of std sigma, r2c, times flt, c2r
    (FFTW's unnormalised pair)."""
    w = rng.standard_normal(nfft) * sigma
    spec = np.fft.rfft(w)                       # dfftw r2c (forward, unnormalised)
    return np.fft.irfft(spec * flt_col, nfft) * nfft   # dfftw c2r is unnormalised


def make_syn(p: Params, engine, stack, out_dir: str, noise_sigma=None, seed: int = 0):
    """Writes test_trace.NN (noise-free) and test_trace.NNwn (with filtered noise) for every
    trace of `p`, from the layer stack (alpha, beta, rho, h).  Returns the noise-free traces
    rft[nfft, ntrc]."""
    alpha, beta, rho, h = stack
    rft = engine.calc_rf(len(alpha), alpha, beta, rho, h)
    flt = engine.flt
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    for t in range(p.ntrc):
        sig = p.sig_min[t] if noise_sigma is None else noise_sigma
        noise = filtered_noise(rng, p.nfft, flt[:, t], sig)
        clean = rft[:p.nsmp, t]
        write_sac(os.path.join(out_dir, f"test_trace.{t + 1:02d}"), clean, p.delta, p.t_start, p.t_end)
        write_sac(os.path.join(out_dir, f"test_trace.{t + 1:02d}wn"), clean + noise[:p.nsmp], p.delta, p.t_start,
                  p.t_end)
    return rft
