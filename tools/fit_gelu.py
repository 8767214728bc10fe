#!/usr/bin/env python3
"""Fits the erfc approximation behind pt_gelu_erf (posetraj_amd/csrc/pt_common.h):
    erfc(z) ~= 2^(-z (c1 + c2 z + ... + c5 z^4)),  z in [0, 4.2]
by Lawson-iterated weighted least squares on -log2(erfc(z)), and reports the fp32-evaluated error of erf and GELU."""
import numpy as np
from scipy.special import erf, erfc

zmax, deg = 4.2, 5
z = np.linspace(0, zmax, 20001)
target = -np.log2(erfc(z))
A = np.vander(z, deg, increasing=True) * z[:, None]
w = erfc(z) * np.log(2) + 1e-9
c, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
for _ in range(200):
    e = np.abs((1 - np.exp2(-(A @ c))) - erf(z))
    w = w * (0.3 + e / e.max()); w /= w.max()
    c, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
c32 = c.astype(np.float32)
zz = np.linspace(0, 8, 400001).astype(np.float32)
P = np.zeros_like(zz)
for k in range(deg - 1, -1, -1):
    P = P * zz + c32[k]
ap = 1 - np.exp2(-(P * zz).astype(np.float64))
x = zz.astype(np.float64) * np.sqrt(2)
print("coefficients c1..c5:", [f"{v:.9g}" for v in c])
print(f"max |erf error| {np.abs(ap - erf(zz.astype(np.float64))).max():.3e}   "
      f"max |gelu error| {np.abs(0.5 * x * (ap - erf(zz.astype(np.float64)))).max():.3e}")
