"""Fit and check the two-range erf of common.h (erf_2range): Chebyshev fits, then the fp32 evaluation (fused multiply-adds emulated
with one rounding per step) against scipy's erf over [-6, 6]."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf, erfc


def cheb_fit(f, a, b, deg, n=400):
    k = np.arange(n); t = np.cos(np.pi * (k + 0.5) / n); x = 0.5 * (b - a) * t + 0.5 * (b + a)
    p = C.cheb2poly(C.chebfit(t, f(x), deg))
    s = np.poly1d([2 / (b - a), -(a + b) / (b - a)]); out = np.poly1d([0.0])
    for i, ci in enumerate(p):
        out = out + ci * (s ** i)
    return out.coeffs[::-1]          # ascending powers


def horner32(c, x):
    acc = np.full_like(x, np.float32(c[-1]), dtype=np.float32)
    for ci in c[-2::-1]:
        acc = (acc.astype(np.float64) * x.astype(np.float64) + np.float64(np.float32(ci))).astype(np.float32)
    return acc


cA = cheb_fit(lambda u: np.where(u > 0, erf(np.sqrt(u)) / np.sqrt(np.maximum(u, 1e-300)), 2 / np.sqrt(np.pi)), 0, 1, 5)
cB = cheb_fit(lambda z: np.log2(erfc(z)), 1, 4, 7)
print("A (ascending in u = z^2):", ", ".join("%.9ef" % c for c in cA))
print("B (ascending in |z|):   ", ", ".join("%.9ef" % c for c in cB))
x = np.linspace(-6, 6, 4000001).astype(np.float32)
z = (x.astype(np.float64) * np.float64(np.float32(0.70710678118654752440))).astype(np.float32)
az = np.abs(z); u = (az.astype(np.float64) * az).astype(np.float32)
ea = (az.astype(np.float64) * horner32(cA, u)).astype(np.float32)
eb = (1.0 - np.exp2(horner32(cB, np.minimum(az, np.float32(4))).astype(np.float64)).astype(np.float32).astype(np.float64)).astype(np.float32)
e = np.copysign(np.where(az < 1, ea, eb), z)
ref_e = erf(z.astype(np.float64)); ref_g = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
g = 0.5 * x.astype(np.float64) * (1.0 + e.astype(np.float64))
g32 = 0.5 * x.astype(np.float64) * (1.0 + ref_e.astype(np.float32).astype(np.float64))
print("max |erf error| %.2e   max |GELU error| %.2e   (GELU with a correctly rounded erff: %.2e)" % (np.abs(e - ref_e).max(), np.abs(g - ref_g).max(), np.abs(g32 - ref_g).max()))
