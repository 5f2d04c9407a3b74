#!/usr/bin/env python3
"""Coefficients of the polynomial erf-GELU used where the result is rounded to the 16-bit type next (csrc/common.h
gelu_lp2): Phi(x) - 1/2 = x Q(x^2) on |x| <= 4, weighted least-squares Chebyshev fit of degree 7 in x^2, rescaled so that
Phi(+-4) is exactly 1 / 0 (beyond +-4 the argument is clamped).  Prints the coefficients and the error against math.erf."""
import math
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P

X0, DEG = 4.0, 7
erf = np.vectorize(math.erf)
n = 4000
t = (np.cos(np.pi * (np.arange(n) + 0.5) / n) + 1) / 2 * X0 * X0
x = np.sqrt(t)
q = np.where(x > 1e-9, 0.5 * erf(x / np.sqrt(2)) / np.maximum(x, 1e-9), 1 / np.sqrt(2 * np.pi))
c = Ch.chebfit(2 * t / (X0 * X0) - 1, q, DEG, w=t + 1e-3)
p = Ch.cheb2poly(c)
u = np.array([-1.0, 2 / (X0 * X0)])
pt = np.zeros(1)
for k, ck in enumerate(p):
    pt = P.polyadd(pt, ck * P.polypow(u, k))
pt *= 0.5 / (X0 * P.polyval(X0 * X0, pt))  # Phi(X0) = 1 exactly
print("coefficients (ascending powers of x^2):")
print(", ".join(f"{v:.10e}f" for v in pt))
xs = np.linspace(-12, 12, 600001)
xf = xs.astype(np.float32)
xc = np.clip(xf, -X0, X0)
tt = xc * xc
acc = np.zeros_like(xf) + np.float32(pt[-1])
for cc in pt[-2::-1]:
    acc = acc * tt + np.float32(cc)
g = xf * (np.float32(0.5) + xc * acc)
ref = 0.5 * xs * (1 + erf(xs / np.sqrt(2)))
err = np.abs(g - ref)
print(f"max |error| = {err.max():.3e} at x = {xs[err.argmax()]:.3f};  max |error| / max(|x|, 1) = "
      f"{(err / np.maximum(np.abs(xs), 1)).max():.3e};  on |x| <= 4: {err[np.abs(xs) <= 4].max():.3e}")
