"""Fit of the erf-GELU used by the kernels (csrc/common.h gelu_erf_f): Phi(x) ~ sigmoid(x (a0 + a1 x^2 + a2 x^4)).
Iteratively re-weighted least squares towards the minimax fit on [-8, 8]; prints the coefficients (also pre-multiplied by
-log2 e, the form the kernel uses) and the error of an fp32 evaluation over [-12, 12] and at extreme arguments."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf

x = np.linspace(-8, 8, 400001)
g = x * 0.5 * (1 + erf(x / np.sqrt(2)))


def model(c, x):
    t = x * x
    return x / (1 + np.exp(-((c[2] * t + c[1]) * t + c[0]) * x))


c, w = np.array([1.5957691, 0.0713548, 0.0]), np.ones_like(x)
for _ in range(30):
    c = least_squares(lambda c: w * (model(c, x) - g), c, method="lm").x
    e = np.abs(model(c, x) - g)
    w = w * (1 + e / e.max()) ** 2
    w /= w.mean()
print("a0 a1 a2 =", c, " max |err| on [-8, 8] =", np.abs(model(c, x) - g).max())
k = [np.float32(-v * 1.4426950408889634) for v in c]
print("kernel constants (-log2(e) * a):", k)
xs = np.concatenate([np.linspace(-12, 12, 2000001), [-1e4, -100, 100, 1e4, 0.0]]).astype(np.float32)
xc = np.clip(xs, np.float32(-8), np.float32(8))
t = xc * xc
q = ((k[2] * t + k[1]) * t + k[0]) * xc
with np.errstate(over="ignore"):
    y = xs * (np.float32(1) / (np.float32(1) + np.exp2(q).astype(np.float32)))
ref = xs.astype(np.float64) * 0.5 * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
print("fp32 evaluation: max |err| =", np.abs(y - ref).max(), " NaN:", bool(np.isnan(y).any()))
