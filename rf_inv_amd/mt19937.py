"""MT19937 exactly as the reference's `module mt19937` (src/mt19937.f90): the 1997
Matsumoto-Nishimura generator with the 69069 LCG seeding (`sgrnd`, :76-88) and
`grnd()` returning y / 2^32 in [0, 1) (:90-126).  The Fortran code works on signed
32-bit integers; the arithmetic below is the same modulo 2^32.
"""
from __future__ import annotations

import numpy as np

_N, _M = 624, 397
_MATA = np.uint32(0x9908B0DF)      # MATA = -1727483681
_UMASK = np.uint32(0x80000000)
_LMASK = np.uint32(0x7FFFFFFF)
_TMASKB = np.uint32(0x9D2C5680)    # -1658038656
_TMASKC = np.uint32(0xEFC60000)    # -272236544


class MT19937:
    def __init__(self, seed: int | None = None):
        self.mt = np.zeros(_N, dtype=np.uint32)
        self.mti = _N + 1          # "sgrnd has not been called" (:73)
        self._out = np.zeros(_N)
        if seed is not None:
            self.sgrnd(seed)

    def sgrnd(self, seed: int):
        """subroutine sgrnd(seed) (:76-88): mt(0) = seed, mt(i) = 69069 * mt(i-1) mod 2^32."""
        x = int(seed) & 0xFFFFFFFF
        for i in range(_N):
            self.mt[i] = x
            x = (69069 * x) & 0xFFFFFFFF
        self.mti = _N

    def _generate(self):
        mt = self.mt
        if self.mti == _N + 1:
            self.sgrnd(4357)       # default seed (:96-100)
        # the recurrence reads words that were already updated in this sweep, so it is
        # evaluated in the reference's three sequential ranges (:102-113)
        for kk in range(_N - _M):
            y = (mt[kk] & _UMASK) | (mt[kk + 1] & _LMASK)
            mt[kk] = mt[kk + _M] ^ (y >> np.uint32(1)) ^ (_MATA if (y & np.uint32(1)) else np.uint32(0))
        for kk in range(_N - _M, _N - 1):
            y = (mt[kk] & _UMASK) | (mt[kk + 1] & _LMASK)
            mt[kk] = mt[kk + (_M - _N)] ^ (y >> np.uint32(1)) ^ (_MATA if (y & np.uint32(1)) else np.uint32(0))
        y = (mt[_N - 1] & _UMASK) | (mt[0] & _LMASK)
        mt[_N - 1] = mt[_M - 1] ^ (y >> np.uint32(1)) ^ (_MATA if (y & np.uint32(1)) else np.uint32(0))
        # tempering (:117-121), vectorised over the block
        y = mt.copy()
        y ^= y >> np.uint32(11)
        y ^= (y << np.uint32(7)) & _TMASKB
        y ^= (y << np.uint32(15)) & _TMASKC
        y ^= y >> np.uint32(18)
        self._out = y.astype(np.float64) / 4294967296.0     # (:123-127) -> [0, 1)
        self.mti = 0

    def grnd(self) -> float:
        """real(8) function grnd() (:90-128)."""
        if self.mti >= _N:
            self._generate()
        v = self._out[self.mti]
        self.mti += 1
        return float(v)
