"""Mirror of the reference's `module forward` public interface
(src/forward.f90:28-41): `flt`, `is_ray_common`, `init_forward`, `calc_rf`.
All evaluation goes through the HIP engine.
"""
from __future__ import annotations

import numpy as np

from .engine import RFEngine
from .params import Params


class Forward:
    def __init__(self, params: Params, engine: RFEngine | None = None, device: int = 0):
        self.p = params
        self.engine = engine
        self._device = device
        self.flt = None
        self.is_ray_common = None

    def init_forward(self, verb: bool = False):
        """subroutine init_forward(verb) (src/forward.f90:47-55): init_filter + check_ray."""
        if self.engine is None:
            self.engine = RFEngine.from_params(self.p, device=self._device)
        self.flt = self.engine.flt                       # src/forward.f90:95-119
        self.is_ray_common = self.engine.is_ray_common   # src/forward.f90:59-91
        if verb and self.p.ntrc > 1:
            print("--- check ray parameters ---")
            if self.is_ray_common:
                print("Ray geometries are common among traces\n-> Single FWD mode\n")
            else:
                print("Ray geometries are not common among traces\n-> Multiple FWD mode\n")

    def calc_rf(self, chain_id, nlay, n, ntrc, rayps, alpha, beta, rho, h):
        """subroutine calc_rf(chain_id, nlay, n, ntrc, rayps, alpha, beta, rho, h, rft)
        (src/forward.f90:123-132).  Returns rft(n, ntrc).  n / ntrc / rayps must be the
        context's (the reference always passes nfft, ntrc, rayps of module params)."""
        if n != self.p.nfft or ntrc != self.p.ntrc or not np.array_equal(np.asarray(rayps), self.p.rayps):
            raise ValueError("calc_rf: n, ntrc, rayps must equal the params the engine was built with")
        return self.engine.calc_rf(nlay, alpha[:nlay], beta[:nlay], rho[:nlay], h[:nlay])
