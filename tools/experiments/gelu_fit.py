#!/usr/bin/env python
"""Coefficients of common.h's gelu_erf_f: gelu(x) = max(x, 0) - |x| * 2^P(|x|), P = weighted-minimax polynomial fit of
log2(1 - Phi(a)) on [0, 6.5]; prints the max abs error of the fp32 evaluation for degrees 4..6."""
import numpy as np
from scipy.special import erf, log_ndtr
from numpy.polynomial import chebyshev as C, polynomial as Pn

A = 6.5


def fit(deg):
    n = 8000
    k = np.arange(n); a = (np.cos(np.pi * (k + 0.5) / n) * 0.5 + 0.5) * A
    y = log_ndtr(-a) / np.log(2)
    V = C.chebvander(2 * a / A - 1, deg)
    wt = a * np.exp(log_ndtr(-a)) * np.log(2) + 1e-9      # d gelu / d P
    w = np.ones(n)
    coef = np.linalg.lstsq(V * wt[:, None], y * wt, rcond=None)[0]
    for _ in range(80):                                    # Lawson iterations towards the minimax fit
        e = np.abs(V @ coef - y) * wt; w = w * (e / e.max() + 1e-3); w /= w.max()
        coef = np.linalg.lstsq(V * (w * wt)[:, None], y * w * wt, rcond=None)[0]
    pa = np.zeros(1)
    for k_, c in enumerate(C.cheb2poly(coef)):
        pa = Pn.polyadd(pa, c * Pn.polypow([-1, 2 / A], k_))
    return pa


for deg in (4, 5, 6):
    pa = fit(deg)
    g = np.linspace(-10, 10, 800001).astype(np.float32)
    a = np.minimum(np.abs(g), np.float32(A)).astype(np.float32)
    L = np.float32(pa[-1]) * np.ones_like(a)
    for c in pa[-2::-1]:
        L = (L * a + np.float32(c)).astype(np.float32)
    out = (np.maximum(g, 0) - a * np.exp2(L).astype(np.float32)).astype(np.float32)
    ref = 0.5 * g.astype(np.float64) * (1 + erf(g.astype(np.float64) / np.sqrt(2)))
    print(f"degree {deg}: max abs error {np.abs(out - ref).max():.2e};  P =", ", ".join(f"{c:.9e}f" for c in pa))
