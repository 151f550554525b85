"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's "record sampled model"
block (src/pt_mcmc.f90:204-286) and of the accumulator set-up of init_pt_mcmc
(src/pt_mcmc.f90:394-430).  Pure-Python loops: for small cases in tests/ only; the product
path (rf_inv_amd/posterior.py -> rf_post_* in librfgpu) never imports this.

Pinned by the reference's own code: tests/test_fortran_shim.py runs the reference's unmodified
pt_mcmc.f90 + mcmc_out.f90 (compiled into oracle/_ref) and compares what they write with the
device accumulators on the same trajectory; this module is the checker for cases that have no
Fortran run (synthetic models, ocean layer, out-of-range amplitudes).

Arrays use C order with the reference's first index LAST, so `.ravel()` is the Fortran
array's memory: nsig[ntrc, nbin_sig] == nsig(nbin_sig, ntrc) etc.
"""
from __future__ import annotations

import numpy as np

from . import rf_oracle as ro


def _fint(x: float) -> int:
    """Fortran int() as compiled for x86-64 (cvttsd2si): truncation; NaN / overflow -> INT_MIN."""
    if not (-2147483649.0 < x < 2147483648.0):
        return -2147483648
    return int(x)


class PosteriorOracle:
    def __init__(self, *, mcfg, ntrc, nsmp, nbin_z, nbin_vs, nbin_vp, nbin_vpvs, nbin_sig, nbin_amp,
                 amp_min, amp_max, z_min, sig_min, sig_max, sig_mode, max_models):
        """mcfg: the dict oracle.rf_oracle.format_model takes (k_max, limits, reference table, sdep)."""
        self.m = mcfg
        self.ntrc, self.nsmp = ntrc, nsmp
        self.nbin_z, self.nbin_vs, self.nbin_vp = nbin_z, nbin_vs, nbin_vp
        self.nbin_vpvs, self.nbin_sig, self.nbin_amp = nbin_vpvs, nbin_sig, nbin_amp
        self.amp_min, self.z_min = amp_min, z_min
        self.sig_min, self.sig_mode = list(sig_min), list(sig_mode)
        # :423-430
        self.dbin_sig = [(sig_max[t] - sig_min[t]) / nbin_sig for t in range(ntrc)]
        self.dbin_amp = (amp_max - amp_min) / nbin_amp
        self.dbin_vp = (mcfg["vp_max"] - mcfg["vp_min"]) / nbin_vp
        self.dbin_vs = (mcfg["vs_max"] - mcfg["vs_min"]) / nbin_vs
        self.dbin_z = (mcfg["z_max"] - 0.0) / nbin_z
        self.dbin_vpvs = (mcfg["vpvs_max"] - mcfg["vpvs_min"]) / nbin_vpvs
        k_max = mcfg["k_max"]
        self.nmod = 0
        self.nk = np.zeros(k_max, dtype=np.int32)
        self.nz = np.zeros(nbin_z, dtype=np.int32)
        self.nsig = np.zeros((ntrc, nbin_sig), dtype=np.int32)
        self.namp = np.zeros((ntrc, nsmp, nbin_amp), dtype=np.int32)
        self.nvpz = np.zeros((nbin_vp, nbin_z), dtype=np.int32)
        self.nvsz = np.zeros((nbin_vs, nbin_z), dtype=np.int32)
        self.nvpvsz = np.zeros((nbin_vpvs, nbin_z), dtype=np.int32)
        self.vp_mean = np.zeros(nbin_z)
        self.vs_mean = np.zeros(nbin_z)
        self.vpvs_mean = np.zeros(nbin_z)
        self.vp_model = np.zeros((max_models, nbin_z))
        self.vs_model = np.zeros((max_models, nbin_z))
        self.vs_model[:, 0] = -999.9                                           # :419
        self.all_likelihood = np.zeros(max_models)
        self.amp_out_of_range = 0

    def record(self, k, z, dvp, dvs, sig, logl, trace, temp=1.0):
        """One pass through :204-286 for one chain.  trace[ntrc, >= nsmp] is the chain's rft."""
        if not (temp <= 1.0 + float(np.float32(1.0e-6))):                      # :204
            return
        m = self.m
        self.nmod += 1
        im = self.nmod - 1
        self.all_likelihood[im] = logl                                         # :210
        self.nk[k - 1] += 1                                                    # :213
        for t in range(self.ntrc):                                             # :216-223
            if self.sig_mode[t] == 1:
                ibin = _fint((sig[t] - self.sig_min[t]) / self.dbin_sig[t]) + 1
                self.nsig[t, ibin - 1] += 1
        for il in range(k - 1):                                                # :226-229
            ibin = _fint((z[il] - self.z_min) / self.dbin_z) + 1
            self.nz[ibin - 1] += 1
        nlay, alpha, beta, _rho, h, _ok = ro.format_model(m, k, z, dvp, dvs)   # :240-242
        tmpz = 0.0
        for il in range(nlay):                                                 # :244-270
            iz1 = _fint(tmpz / self.dbin_z) + 1
            iz2 = _fint((tmpz + h[il]) / self.dbin_z) + 1 if il < nlay - 1 else self.nbin_z + 1
            a, b = float(alpha[il]), float(beta[il])
            ivp = _fint((a - m["vp_min"]) / self.dbin_vp) + 1
            ivs = max(1, _fint((b - m["vs_min"]) / self.dbin_vs) + 1)
            ivpvs = _fint(((a / b) - m["vpvs_min"]) / self.dbin_vpvs) + 1
            ivpvs = min(max(1, ivpvs), self.nbin_vpvs)
            for iz in range(iz1, iz2):
                self.nvpz[ivp - 1, iz - 1] += 1
                self.vp_mean[iz - 1] = self.vp_mean[iz - 1] + a
                self.nvsz[ivs - 1, iz - 1] += 1
                self.nvpvsz[ivpvs - 1, iz - 1] += 1
                if b > 0.0:
                    self.vpvs_mean[iz - 1] = self.vpvs_mean[iz - 1] + a / b
                    self.vs_mean[iz - 1] = self.vs_mean[iz - 1] + b
                    self.vs_model[im, iz - 1] = b
                else:                                                          # :263-266 (assignments)
                    self.vpvs_mean[iz - 1] = m["vpvs_min"]
                    self.vs_mean[iz - 1] = m["vs_min"]
                    self.vs_model[im, iz - 1] = m["vs_min"]
                self.vp_model[im, iz - 1] = a
            tmpz = tmpz + h[il]
        for t in range(self.ntrc):                                             # :273-285
            for it in range(self.nsmp):
                ibin = _fint((trace[t][it] - self.amp_min) / self.dbin_amp) + 1
                if ibin < 1:
                    ibin = 1
                    self.amp_out_of_range += 1
                elif ibin > self.nbin_amp:
                    ibin = self.nbin_amp
                    self.amp_out_of_range += 1
                self.namp[t, it, ibin - 1] += 1
