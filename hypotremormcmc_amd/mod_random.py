"""Host mirror of the reference's process-global RNG (src/mod_random.f90) used for the set-up draws only
(initial models, initial temperatures, src/hypo_tremor_mcmc.f90:120-211).  The per-iteration draws are
made on the device from the state handed over at chain creation."""
from __future__ import annotations

import math

_M = 0xFFFFFFFF


def _seed_word(i: int, j1: int) -> int:
    j2 = (j1 * j1) & _M
    j4 = (j2 * j2) & _M
    return (i * j4 + 1000 * i * j2 + i) & _M  # wrapping int32 arithmetic, src/mod_random.f90:49-52


class Xorshift128:
    PI2 = 2.0 * math.acos(-1.0)

    def __init__(self, rank: int = 0, seeds=(5551111, 453222, 4444431, 6765)):
        j1 = (rank + 1) & _M
        self.x, self.y, self.z, self.w = (_seed_word(s, j1) for s in seeds)

    @property
    def state(self):
        return (self.x, self.y, self.z, self.w)

    def _next(self) -> int:
        t = (self.x ^ (self.x << 11)) & _M
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = ((self.w ^ (self.w >> 19)) ^ (t ^ (t >> 8))) & _M
        return self.w

    @staticmethod
    def _signed(w: int) -> int:
        return w - 0x100000000 if w & 0x80000000 else w

    def rand_u(self) -> float:  # [0, 1)  :72
        return (float(self._signed(self._next())) + 2147483648.0) / 4294967296.0

    def rand_u2(self) -> float:  # (0, 1)  :90
        return (float(self._signed(self._next())) + 2147483648.0 + 0.5) / 4294967296.0

    def rand_g(self) -> float:  # :98-100
        v1 = self.rand_u2()
        v2 = self.rand_u2()
        return math.sqrt(-2.0 * math.log(v1)) * math.cos(self.PI2 * v2)

    def rand_r(self) -> float:  # :109-110
        return math.sqrt(-2.0 * math.log(self.rand_u2()))
