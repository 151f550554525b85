"""Host-side mirror of the layer-stack part of the reference's `module model`
(src/model.f90): read_ref_model, format_model, vp_to_rho and the 3-array
quick_sort of src/sort.f90.  In the drop-in Fortran build these stay in the
reference's own model.f90; this numpy version feeds the Python host and is checked
bit-for-bit against the oracle in tests/test_host_model.py.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .params import Params


@dataclass
class RefModel:
    """vp_ref, vs_ref, dz_ref, z_ref_min, z_ref_max of module model (src/model.f90:35-36)."""
    vp_ref: np.ndarray
    vs_ref: np.ndarray
    dz_ref: float
    z_ref_min: float
    z_ref_max: float


def read_ref_model(path: str) -> RefModel:
    """subroutine read_ref_model (src/model.f90:104-171)."""
    rows = []
    with open(path) as f:
        for line in f:
            t = line.replace(",", " ").split()
            if len(t) < 3:
                break
            try:
                rows.append((float(t[0]), float(t[1]), float(t[2])))
            except ValueError:
                break
    a = np.asarray(rows)
    z = a[:, 0]
    if len(z) >= 3:
        # src/model.f90:129-137: constant depth increment (tolerance is a single literal)
        dz = np.diff(z)
        if np.any(np.abs(dz[1:] - dz[:-1]) > float(np.float32(1.0e-5))):
            raise ValueError(f"ERROR: Depth increment must be constant in {path}")
    return RefModel(vp_ref=a[:, 1].copy(), vs_ref=a[:, 2].copy(), dz_ref=float(z[-1] - z[-2]),
                    z_ref_min=float(z[0]), z_ref_max=float(z[-1]))


_C1, _C2, _C3, _C4, _C5 = (float(np.float32(c)) for c in (1.6612, 0.4721, 0.0671, 0.0043, 0.000106))


def vp_to_rho(a1: float) -> float:
    """function vp_to_rho (src/model.f90:298-314), Brocher (2005); the coefficients
    are default-REAL literals, i.e. float32 values promoted to double."""
    a2 = a1 * a1
    a3 = a2 * a1
    a4 = a3 * a1
    a5 = a4 * a1
    return _C1 * a1 - _C2 * a2 + _C3 * a3 - _C4 * a4 + _C5 * a5


def _quick_sort(a, il, ir, b, c):
    """recursive subroutine quick_sort (src/sort.f90:34-68), 1-based inclusive."""
    if ir - il <= 0:
        return
    ipiv = (il + ir) // 2
    piv = a[ipiv - 1]
    for x in (a, b, c):
        x[ipiv - 1], x[ir - 1] = x[ir - 1], x[ipiv - 1]
    i = il
    for j in range(il, ir + 1):
        if a[j - 1] < piv:
            for x in (a, b, c):
                x[i - 1], x[j - 1] = x[j - 1], x[i - 1]
            i += 1
    for x in (a, b, c):
        x[i - 1], x[ir - 1] = x[ir - 1], x[i - 1]
    _quick_sort(a, il, i, b, c)
    _quick_sort(a, i + 1, ir, b, c)


def _nint(x: float) -> int:
    return int(np.floor(x + 0.5)) if x >= 0 else -int(np.floor(0.5 - x))


_TOP_FAC = float(np.float32(0.125))


def format_model(p: Params, ref: RefModel, prop_k, prop_z, prop_dvp, prop_dvs):
    """subroutine format_model (src/model.f90:175-290).

    Returns (nlay, alpha, beta, rho, h, is_valid); arrays have length nlay."""
    k = int(prop_k)
    tz = [float(x) for x in prop_z[:max(p.k_max - 1, 1)]]
    tvp = [float(x) for x in prop_dvp[:p.k_max]]
    tvs = [float(x) for x in prop_dvs[:p.k_max]]
    _quick_sort(tz, 1, k, tvp, tvs)

    alpha, beta, rho, h = [], [], [], []
    valid = True

    def push(iz, dvp, dvs, thick):
        nonlocal valid
        b = ref.vs_ref[iz - 1] + dvs
        a = ref.vp_ref[iz - 1] + dvp if p.vp_mode == 1 else ref.vp_ref[iz - 1]
        if (a < p.vp_min or a > p.vp_max or b < p.vs_min or b > p.vs_max
                or a / b < p.vpvs_min or a / b > p.vpvs_max):
            valid = False
        alpha.append(float(a)); beta.append(float(b)); rho.append(vp_to_rho(float(a))); h.append(thick)
        return float(a)

    if p.sdep > 0.0:                                     # :201-207
        alpha.append(1.5); beta.append(-999.0); rho.append(1.0); h.append(p.sdep)
    zc = 0.5 * (p.sdep + tz[0])                          # :211
    iz = _nint((zc - ref.z_ref_min) / ref.dz_ref) + 1
    a = push(iz, tvp[0], tvs[0], tz[0] - p.sdep)
    if h[-1] < _TOP_FAC * a:                             # :229 (not h_min)
        valid = False
    for j in range(2, k + 1):                            # :235-262
        zc = 0.5 * (tz[j - 1] + tz[j - 2])
        iz = _nint((zc - ref.z_ref_min) / ref.dz_ref) + 1
        push(iz, tvp[j - 1], tvs[j - 1], tz[j - 1] - tz[j - 2])
        if h[-1] < p.h_min:
            valid = False
    zc = 0.5 * (p.z_max + tz[k - 1])                     # :267
    iz = _nint((zc - ref.z_ref_min) / ref.dz_ref) + 1
    push(iz, tvp[p.k_max - 1], tvs[p.k_max - 1], 999.0)
    return (len(alpha), np.asarray(alpha), np.asarray(beta), np.asarray(rho), np.asarray(h), valid)
