#!/usr/bin/env python3
"""Coefficients of the transcendental-free GELU / GELU' of the bf16 kernels (csrc/fw_common.h: gelu_poly, gelu_grad_poly).

  Phi(x) - 1/2        = x P(x^2)      GELU(x)  = x (1/2 + xc P(xc^2)),   xc = clamp(x, -4, 4)
  GELU'(x) - 1/2      = x R(x^2)      GELU'(x) = 1/2 + xc R(xc^2)

P, R: degree 7 in x^2, fitted on [0, 4] by Lawson-reweighted least squares (-> minimax) of the error in the FINAL quantity
(weight x^2 for P: error of GELU; weight x for R).  Prints the coefficients and the f32-evaluated maximum errors."""
import numpy as np
from scipy.special import erf

C, DEG = 4.0, 7


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def phi(x):
    return np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)


def fit(fun, weight):
    x = np.cos(np.linspace(0, np.pi, 4001)) * 0.5 * C + 0.5 * C
    x = x[x > 1e-6]
    u, y, w = x * x, fun(x), np.ones_like(x)
    for _ in range(60):
        A = np.vander(u / C ** 2, DEG + 1, increasing=True)
        sw = np.sqrt(w)
        coef = np.linalg.lstsq(A * sw[:, None], y * sw, rcond=None)[0]
        err = np.abs(A @ coef - y) * weight(x)
        w = w * (err / err.max() + 1e-3)
        w /= w.sum()
    return coef / (C ** 2) ** np.arange(DEG + 1)


def horner(co, u):
    p = np.zeros_like(u)
    for a in co[::-1].astype(np.float32):
        p = p * u + a
    return p


P = fit(lambda x: (Phi(x) - 0.5) / x, lambda x: x * x)
R = fit(lambda x: (Phi(x) + x * phi(x) - 0.5) / x, lambda x: x)
print('P (highest degree last):', ', '.join(f'{v:.9e}f' for v in P))
print('R (highest degree last):', ', '.join(f'{v:.9e}f' for v in R))
xs = np.linspace(-8, 8, 800001).astype(np.float32)
x64 = xs.astype(np.float64)
xc = np.clip(xs, -4, 4)
u = xc * xc
g = xs * (np.float32(0.5) + xc * horner(P, u))
d = np.float32(0.5) + xc * horner(R, u)
inside = np.abs(xs) <= 4
eg = np.abs(g - x64 * Phi(x64))
print(f'GELU : max abs error {eg[inside].max():.2e} on [-4, 4], {eg.max():.2e} on [-8, 8]')
print(f"GELU': max abs error {np.abs(d - (Phi(x64) + x64 * phi(x64))).max():.2e} on [-8, 8]")
