"""Host mirror of `type model` (reference src/cls_model.f90:5-27): the plugin surface for chain state.
Same fields (nx, prior_type, x, mu, sigma, step_size), same setters, same prior sampling."""
from __future__ import annotations

import numpy as np

from .mod_random import Xorshift128


class Model:
    def __init__(self, nx: int):
        self.nx = nx
        self.prior_type = np.zeros(nx, dtype=np.int32)  # 0 Gaussian, 1 Rayleigh
        self.x = np.zeros(nx)
        self.mu = np.zeros(nx)
        self.sigma = np.ones(nx)
        self.step_size = np.zeros(nx)

    # indices are 1-based like the reference's
    def set_prior(self, i: int, mu: float, sigma: float, prior_type: int = 0):
        self.mu[i - 1] = mu
        self.sigma[i - 1] = sigma
        self.prior_type[i - 1] = prior_type

    def set_perturb(self, i: int, step_size: float):
        self.step_size[i - 1] = step_size

    def get_nx(self) -> int:
        return self.nx

    def get_x(self, i: int) -> float:
        return float(self.x[i - 1])

    def get_all_x(self) -> np.ndarray:
        return self.x.copy()

    def set_x(self, i: int, x: float):
        self.x[i - 1] = x

    def generate_model(self, rng: Xorshift128):  # src/cls_model.f90:139-158
        for i in range(self.nx):
            if self.prior_type[i] == 0:
                self.x[i] = self.mu[i] + rng.rand_g() * self.sigma[i]
            elif self.prior_type[i] == 1:
                self.x[i] = self.mu[i] + rng.rand_r() * self.sigma[i]
            else:
                raise ValueError(f"element {i + 1}: prior type {self.prior_type[i]} is neither 0 (Gaussian) nor 1 (Rayleigh)")
