"""Independent known answers for plane P / SV waves in a stack of solid layers over a half-space.

TEST INFRASTRUCTURE.  This is NOT a restatement of the reference's propagator-matrix code
(src/forward.f90:212-442): it is the reflectivity formulation (Kennett 1983, "Seismic wave propagation in
stratified media", ch. 5-6) built from first principles --

  * plane-wave displacement / traction vectors straight from Hooke's law (no eigenvector-matrix formula is
    copied: polarisations are the slowness direction for P and its normal for SV, tractions come from
    differentiating the plane wave);
  * welded-interface and free-surface scattering matrices by SOLVING the boundary conditions numerically;
  * the stack response by Kennett's addition rules and the reverberation operator (I - R_D R_F)^-1,
    instead of a product of layer propagators.

The receiver-function processing on top is the textbook definition (Langston 1979; Clayton & Wiggins 1976
water-level deconvolution; unit-height Gaussian pulse), written against physical time, with the reference's
integer quirks listed explicitly in `REFERENCE_QUIRKS` -- each one cites the reference line it comes from.

Conventions here: z positive DOWN, time dependence exp(-i w t) while solving; at the end the spectra are
conjugated (numpy / FFTW synthesise with exp(+i w t)) and the vertical component is flipped to positive UP,
the receiver-function convention (direct P positive on both components).
"""
import numpy as np

PI = 3.1415926535897931

# integer / literal quirks of the reference that a physical formulation cannot know (parity checklist,
# SURVEY.md appendix A); everything else below is physics
REFERENCE_QUIRKS = {
    "dc_omega": float(np.float32(1.0e-5)),   # DC bin evaluated at w = 1.0e-5 (single literal), forward.f90:245-248
    "s_lag_samples": 1,                      # S-RF: the 1-based sample i holds the lag t_start + i*delta, not
                                             # t_start + (i-1)*delta: j = mod(nfft + npre - i + 1, nfft), forward.f90:188
    "water_level": 0.001,                    # forward.f90:149,152
}


def _nint(x):
    return int(np.floor(x + 0.5)) if x >= 0 else -int(np.floor(0.5 - x))


def _wave_vectors(w, p, alpha, beta, rho):
    """F[nw, 4, 4]: columns = (u_x, u_z, t_xz, t_zz) of unit-displacement-amplitude plane waves
    [upgoing P, upgoing SV, downgoing P, downgoing SV] in a medium, horizontal slowness p."""
    mu = rho * beta * beta
    lam = rho * alpha * alpha - 2.0 * mu
    xi = np.sqrt(1.0 / alpha ** 2 - p * p + 0j)
    eta = np.sqrt(1.0 / beta ** 2 - p * p + 0j)
    cols = []
    for kind, s in (("P", -xi), ("S", -eta), ("P", xi), ("S", eta)):       # upgoing: vertical slowness < 0 (z down)
        # P: along the slowness vector.  SV: normal to it, signed like Aki & Richards' convention (upgoing SV =
        # (cos j, 0, sin j) with z down) -- the sign matters only for the S-RF without deconvolution, which is
        # normalised by a SIGNED maximum; the opposite sign is rejected by tests/test_analytic_pins.py
        d = (alpha * p, alpha * s) if kind == "P" else (-beta * s, beta * p)
        dx, dz = d
        txz = 1j * w * mu * (s * dx + p * dz)
        tzz = 1j * w * (lam * (p * dx + s * dz) + 2.0 * mu * s * dz)
        cols.append(np.stack([dx + 0 * w, dz + 0 * w, txz, tzz], axis=-1))
    return np.stack(cols, axis=-1), xi, eta


def surface_response(w, p, ipha, alpha, beta, rho, h):
    """(u_x, u_z)[nw] at the free surface (z down, exp(-iwt)) for a unit upgoing P (ipha = 1) or SV (ipha = -1)
    wave incident from the half-space = last entry of alpha/beta/rho; h[:-1] are the layer thicknesses.
    The incident wave's phase is zero at the deepest interface."""
    n = len(alpha)
    F = [_wave_vectors(w, p, alpha[i], beta[i], rho[i]) for i in range(n)]
    nw = w.size
    eye = np.tile(np.eye(2, dtype=complex), (nw, 1, 1))
    R_D = np.zeros((nw, 2, 2), complex)
    T_U = eye.copy()
    for a in range(n - 2, -1, -1):                  # interface between layer a (above) and a + 1 (below)
        Fa, Fb = F[a][0], F[a + 1][0]
        A = np.concatenate([Fa[:, :, :2], -Fb[:, :, 2:]], axis=2)          # unknowns: (u_a, d_b)
        sol_u = np.linalg.solve(A, Fb[:, :, :2])                           # incoming u_b = I
        sol_d = np.linalg.solve(A, -Fa[:, :, 2:])                          # incoming d_a = I
        t_u, r_u = sol_u[:, :2, :], sol_u[:, 2:, :]
        r_d, t_d = sol_d[:, :2, :], sol_d[:, 2:, :]
        R_new = r_d + t_u @ R_D @ np.linalg.solve(eye - r_u @ R_D, t_d)
        T_U = t_u @ np.linalg.solve(eye - R_D @ r_u, T_U)
        R_D = R_new
        # up through layer a: phase delays of P and SV
        E = np.zeros((nw, 2, 2), complex)
        E[:, 0, 0] = np.exp(1j * w * F[a][1] * h[a])
        E[:, 1, 1] = np.exp(1j * w * F[a][2] * h[a])
        R_D = E @ R_D @ E
        T_U = E @ T_U
    F1 = F[0][0]
    R_F = -np.linalg.solve(F1[:, 2:, 2:], F1[:, 2:, :2])                   # traction-free surface
    e = np.zeros((nw, 2, 1), complex)
    e[:, 0 if ipha == 1 else 1, 0] = 1.0
    v_up = np.linalg.solve(eye - R_D @ R_F, T_U @ e)
    u = (F1[:, :2, :2] + F1[:, :2, 2:] @ R_F) @ v_up
    return u[:, 0, 0], u[:, 1, 0]


def receiver_function(nfft, delta, t_start, a_gus, rayp, ipha, deconv, alpha, beta, rho, h, s_polarity=1.0):
    """The receiver function on the reference's time axis (nfft samples from t_start), from the reflectivity
    response above.  s_polarity: sign convention of the incident SV wave's displacement (matters only for the
    S-RF without deconvolution, which is normalised by a SIGNED maximum)."""
    alpha, beta, rho, h = (np.asarray(x, float) for x in (alpha, beta, rho, h))
    nh = nfft // 2 + 1
    w = np.arange(nh) * (2.0 * PI / (nfft * delta))
    w[0] = REFERENCE_QUIRKS["dc_omega"]
    ux, uz = surface_response(w, rayp, ipha, alpha, beta, rho, h)
    if ipha == -1:
        ux, uz = s_polarity * ux, s_polarity * uz
    R = np.conj(ux)            # radial, exp(+iwt) synthesis
    V = -np.conj(uz)           # vertical, positive up
    # unit-height Gaussian pulse exp(-a^2 t^2) under an UNNORMALISED inverse DFT of n points
    g = np.exp(-(w / (2.0 * a_gus)) ** 2) * np.sqrt(PI) / (a_gus * nfft * delta)

    def synth(spec):           # unnormalised real inverse DFT (imaginary parts of DC / Nyquist have no effect)
        return np.fft.irfft(spec, nfft) * nfft

    xi = np.sqrt(1.0 / alpha ** 2 - rayp ** 2)
    eta = np.sqrt(1.0 / beta ** 2 - rayp ** 2)
    if deconv:
        num, den = (R, V) if ipha == 1 else (V, R)
        amp = np.abs(den) ** 2
        rf_spec = num * np.conj(den) / np.maximum(amp, REFERENCE_QUIRKS["water_level"] * amp.max())
        t_direct = 0.0         # spectral division removes the direct arrival's delay
    else:
        rf_spec = R if ipha == 1 else V
        t_direct = float(np.sum(h[:-1] * (xi if ipha == 1 else eta)[:-1]))
    x = synth(rf_spec * g)
    i = np.arange(nfft)
    if ipha == 1:
        # sample i <-> lag t_start + i delta after the direct arrival (rounded once to the sample grid)
        src = (i + _nint((t_start + t_direct) / delta)) % nfft
        out = x[src]
    else:
        # S: time runs backwards from the direct S (precursors at positive lag), polarity flipped
        lag0 = REFERENCE_QUIRKS["s_lag_samples"]
        src = (_nint((t_direct - t_start) / delta) - i - lag0) % nfft
        out = -x[src]
    if not deconv:
        out = out / synth(V * g).max()    # normalised by the (signed) maximum of the filtered vertical trace
    return out
