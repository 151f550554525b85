"""Mirror of the reference's `module likelihood` public interface
(src/likelihood.f90:28-37): state `sig`, `rft`, `log_likelihood`; `init_likelihood`,
`calc_likelihood`.  The noise-covariance pseudo-inverse is built on the host with
LAPACK dgesvd exactly like init_r_inv; everything per-evaluation runs on the GPU.
"""
from __future__ import annotations

import numpy as np

from .engine import RFEngine
from .model import RefModel, format_model
from .params import Params


def init_r_inv(nsmp: int, a_gus, delta: float):
    """subroutine init_r_inv (src/likelihood.f90:168-241) with LAPACK dgesvd (scipy).
    Returns r_inv[ntrc, nsmp, nsmp], r_inv[t].ravel() == column-major r_inv(:, :, t)."""
    from scipy.linalg import svd

    a_gus = np.atleast_1d(np.asarray(a_gus, dtype=np.float64))
    idx = np.arange(nsmp)
    e2 = ((idx[:, None] - idx[None, :]) ** 2).astype(np.float64)
    out = np.empty((a_gus.size, nsmp, nsmp))
    for t, a in enumerate(a_gus):
        r = np.exp(-a ** 2 * delta ** 2)
        u, s, vt = svd(r ** e2, full_matrices=True, lapack_driver="gesvd")
        keep = s > 1.0e-3                                   # src/likelihood.f90:214
        dinv = np.zeros_like(s)
        dinv[keep] = 1.0 / s[keep]
        out[t] = ((vt.T * dinv[None, :]) @ u.T).T           # (V D) U^T, stored column-major
    return out


class Likelihood:
    def __init__(self, params: Params, ref: RefModel | None = None, engine: RFEngine | None = None,
                 device: int = 0):
        self.p = params
        self.ref = ref
        self.engine = engine
        self._device = device
        self.sig = None             # sig(ntrc, nchains)
        self.log_likelihood = None  # log_likelihood(nchains)

    # -- state -----------------------------------------------------------------
    @property
    def rft(self):
        """rft(nfft, ntrc, nchains): device-resident; materialised on access."""
        n = self.p.nchains
        out = np.empty((self.p.nfft, self.p.ntrc, n), order="F")
        for c in range(n):
            out[:, :, c] = self.engine.get_rft(c, which=0)
        return out

    def init_likelihood(self, verb=False, k=None, z=None, dvp=None, dvs=None, rng=None):
        """subroutine init_likelihood(verb) (src/likelihood.f90:42-52): init_sig,
        init_r_inv, init_rft.  k/z/dvp/dvs are module model's chain state
        (k(nchains), z(k_max-1, nchains), dvp/dvs(k_max, nchains)); rng() -> grnd()."""
        p = self.p
        # init_sig (src/likelihood.f90:107-139)
        self.sig = np.zeros((p.ntrc, p.nchains), order="F")
        for c in range(p.nchains):
            for t in range(p.ntrc):
                if p.sig_mode[t] == 1:
                    self.sig[t, c] = p.sig_min[t] + rng() * (p.sig_max[t] - p.sig_min[t])
                else:
                    self.sig[t, c] = p.sig_min[t]
        if self.engine is None:
            self.engine = RFEngine.from_params(p, r_inv=init_r_inv(p.nsmp, p.a_gus, p.delta),
                                               device=self._device)
        # init_rft (src/likelihood.f90:143-163): first evaluation of every chain
        self.log_likelihood = np.zeros(p.nchains)
        if k is not None:
            for c in range(p.nchains):
                ll, _ = self.calc_likelihood(c + 1, True, k[c], z[:, c], dvp[:, c], dvs[:, c], self.sig[:, c],
                                             want_rft=False)
                self.log_likelihood[c] = ll
                self.engine.commit([c], [1])

    def calc_likelihood(self, chain_id, fwd_flag, prop_k, prop_z, prop_dvp, prop_dvs, sig, want_rft=True):
        """subroutine calc_likelihood(chain_id, fwd_flag, prop_k, prop_z, prop_dvp,
        prop_dvs, sig, prop_log_likelihood, prop_rft) (src/likelihood.f90:56-101).
        chain_id is 1-based like the reference.  Returns (prop_log_likelihood, prop_rft)."""
        if fwd_flag:
            nlay, alpha, beta, rho, h, _ = format_model(self.p, self.ref, prop_k, prop_z, prop_dvp, prop_dvs)
            return self.engine.calc_likelihood(chain_id - 1, True, nlay, alpha, beta, rho, h, sig, want_rft)
        # sigma-only proposal: the stored trace of the chain is re-used (src/likelihood.f90:81)
        one = np.ones(2)
        return self.engine.calc_likelihood(chain_id - 1, False, 2, one, one, one, one, sig, want_rft)
