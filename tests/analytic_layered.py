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

An ocean (the reference marks it with beta(1) < 0: src/forward.f90:229, model.f90:203-206) is a FLUID layer on top
of the solid stack with the receiver on the sea floor: the traction-free surface is replaced by the sea floor's
own boundary conditions -- no shear traction, continuous vertical displacement and normal traction -- solved
numerically together with the acoustic up- and down-going waves of the water column under a pressure-free sea
surface.  (The reference instead multiplies a 2x2 liquid-layer propagator into rows of the solid chain,
src/forward.f90:276-287, 424-442.)

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


def _sea_floor_reflection(w, p, F1, alpha_w, rho_w, depth):
    """R[nw, 2, 2]: downgoing (P, SV) amplitudes in the top solid layer per unit upgoing (P, SV) wave hitting
    the sea floor from below, under a water column of the given depth with a pressure-free surface.
    Acoustic plane waves of unit displacement amplitude along their slowness vector: u_z = alpha_w s,
    t_zz = i w lambda_w (p u_x + s u_z) with lambda_w = rho_w alpha_w^2; the sea floor is z = 0, the sea surface
    z = -depth.  Unknowns per incident wave: the two reflected solid waves and the upgoing acoustic amplitude."""
    nw = w.size
    xi_w = np.sqrt(1.0 / alpha_w ** 2 - p * p + 0j)
    lam_w = rho_w * alpha_w ** 2
    uz_up, uz_dn = alpha_w * (-xi_w), alpha_w * xi_w
    tzz = 1j * w * lam_w * alpha_w * (p * p + xi_w * xi_w)          # the same for both directions
    # pressure-free sea surface: up exp(i w xi_w depth) + down exp(-i w xi_w depth) = 0
    down_per_up = -np.exp(2j * w * xi_w * depth)
    wz = uz_up + uz_dn * down_per_up                                 # water column at the sea floor, per unit U
    wt = tzz * (1.0 + down_per_up)
    A = np.zeros((nw, 3, 3), complex)
    A[:, 0, :2] = F1[:, 2, 2:]                                       # t_xz of the reflected waves = -t_xz incident
    A[:, 1, :2] = F1[:, 1, 2:]                                       # u_z continuous
    A[:, 1, 2] = -wz
    A[:, 2, :2] = F1[:, 3, 2:]                                       # t_zz continuous
    A[:, 2, 2] = -wt
    rhs = -np.stack([F1[:, 2, :2], F1[:, 1, :2], F1[:, 3, :2]], axis=1)
    return np.linalg.solve(A, rhs)[:, :2, :]


def surface_response(w, p, ipha, alpha, beta, rho, h):
    """(u_x, u_z)[nw] at the free surface -- under an ocean (beta[0] < 0): of the solid at the sea floor -- (z down,
    exp(-iwt)) for a unit upgoing P (ipha = 1) or SV (ipha = -1) wave incident from the half-space = last entry of
    alpha/beta/rho; h[:-1] are the layer thicknesses.  The incident wave's phase is zero at the deepest interface."""
    water = None
    if beta[0] < 0:                                  # ocean: layer 0 is water, the receiver sits on the sea floor
        water = (alpha[0], rho[0], h[0])
        alpha, beta, rho, h = alpha[1:], beta[1:], rho[1:], h[1:]
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
    if water is None:
        R_F = -np.linalg.solve(F1[:, 2:, 2:], F1[:, 2:, :2])               # traction-free surface
    else:
        R_F = _sea_floor_reflection(w, p, F1, *water)
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

    solid = slice(1, None) if beta[0] < 0 else slice(None)     # the direct wave reaches the sea floor, not the sea surface
    alpha, beta, h = alpha[solid], beta[solid], h[solid]
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
