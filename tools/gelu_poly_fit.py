"""Coefficients of the polynomial normal CDF that `gelu_both_poly4` (open_clip_amd/csrc/ocn_common.h, developer-build alternative of the GELU
epilogue's arithmetic) evaluates:  Phi(x) - 1/2 ~ x Q(x^2) for |x| <= X0, x clamped to [-X0, X0] beyond (Phi(4.25) = 1 - 1.07e-5).
Weighted minimax fit (Lawson iteration on a dense grid; error weight x, i.e. the absolute error of Phi), then the fp32 Horner evaluation the
kernel performs (one rounding per fma) is emulated and its worst |Phi error| printed.  usage: python tools/gelu_poly_fit.py [X0 [n]]  (developer tool)"""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C
from numpy.polynomial import polynomial as P
from scipy.special import erf


def fit(X0, n, iters=60):
    x = np.linspace(1e-6, X0, 40001)
    u = x * x
    f = 0.5 * erf(x / np.sqrt(2.0)) / x
    t = 2 * u / (X0 * X0) - 1
    V = C.chebvander(t, n - 1)
    w = np.ones_like(x)
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V * (w * x)[:, None], f * w * x, rcond=None)
        err = np.abs((V @ c - f) * x)
        w = w * (0.5 + err / err.max())
        w /= w.mean()
    mono, tp = np.zeros(1), np.ones(1)
    for ck in C.cheb2poly(c):
        mono = P.polyadd(mono, ck * tp)
        tp = P.polymul(tp, np.array([-1.0, 2 / (X0 * X0)]))
    return mono.astype(np.float32)


def phi_fp32(x, X0, q):
    """what the kernel computes, in fp32 with one rounding per fma (the product is exact in float64)"""
    x = x.astype(np.float32)
    xc = np.clip(x, np.float32(-X0), np.float32(X0))
    u = (xc * xc).astype(np.float32)
    acc = np.full_like(u, q[-1])
    for ck in q[-2::-1]:
        acc = (acc.astype(np.float64) * u.astype(np.float64) + np.float64(ck)).astype(np.float32)
    return (xc.astype(np.float64) * acc.astype(np.float64) + 0.5).astype(np.float32)


if __name__ == "__main__":
    X0 = float(sys.argv[1]) if len(sys.argv) > 1 else 4.25
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    q = fit(X0, n)
    x = np.linspace(-X0 - 4, X0 + 4, 2000001)
    true = 0.5 * (1 + erf(x / np.sqrt(2)))
    got = phi_fp32(x, X0, q)
    print(f"X0 = {X0}, {n} coefficients: max |Phi error| = {np.abs(got - true).max():.3e} (at x = {x[np.abs(got - true).argmax()]:.3f})")
    print("Q (ascending powers of x^2): " + ", ".join(f"{float(c):.9e}f" for c in q))
