"""Matern-nu SDE in companion form (reference: pssgp/kernels/matern/common.py:10-52).

For nu = d - 1/2 the spectral density factorises as (lambda + i w)^-d, giving
  F = shift matrix with last row  -C(d,k) lambda^(d-k), k = 0..d-1      (common.py:10-18)
  L = e_d,  H = e_1^T
  q = (2 lambda)^(2d-1) sigma^2 ((d-1)!)^2 / (2d-2)!                    (common.py:21-23)
with lambda = sqrt(2d-1) / lengthscale                                   (common.py:45).
"""
import math

import numpy as np


def get_matern_sde(variance, lengthscales, d):
    lam = math.sqrt(2 * d - 1) / float(lengthscales)
    F = np.diag(np.ones(d - 1), k=1)
    for k in range(d):
        F[d - 1, k] -= math.comb(d, k) * lam ** (d - k)
    L = np.zeros((d, 1))
    L[d - 1, 0] = 1.0
    H = np.zeros((1, d))
    H[0, 0] = 1.0
    q = (2.0 * lam) ** (2 * d - 1) * float(variance) * math.factorial(d - 1) ** 2 \
        / math.factorial(2 * d - 2)
    return F, L, H, np.array([[q]])
