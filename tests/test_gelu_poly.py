"""The polynomial normal CDF of the GELU epilogue (``gelu_both_poly4``, open_clip_amd/csrc/ocn_common.h -- the product library's GELU arithmetic
since round 5; reference arithmetic: nn.GELU(), src/open_clip/transformer.py:295-299): the coefficients IN THE HEADER, evaluated the way the kernel
does (fp32, one rounding per fma, argument clamped to +-4.25), against erf in float64.  CPU only -- the kernels that use it are checked against
torch on the GPU (tests/test_kernels_gpu.py, tests/test_gemm_bench_shapes_gpu.py)."""
import os
import re

import numpy as np
from scipy.special import erf

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open_clip_amd", "csrc", "ocn_common.h")


def _header_constants():
    src = open(HDR).read()
    body = src[src.index("void gelu_both_poly4"):]
    body = body[:body.index("\n}\n")]
    clamp = float(re.search(r"fmed3f\(x\[i\], -([0-9.]+)f, ([0-9.]+)f\)", body).group(2))
    first = re.search(r"q = u \* ([-0-9.e+]+)f \+ ([-0-9.e+]+)f;", body)
    rest = re.findall(r"q = q \* u \+ ([-0-9.e+]+)f;", body)
    coeffs = [float(first.group(1)), float(first.group(2))] + [float(c) for c in rest]  # highest power first
    return clamp, np.array(coeffs, dtype=np.float32)


def _fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)


def test_polynomial_cdf_of_the_header_is_within_its_stated_error():
    clamp, q = _header_constants()
    assert clamp == 4.25 and len(q) == 9
    x = np.linspace(-12.0, 12.0, 1200001).astype(np.float32)
    xc = np.clip(x, np.float32(-clamp), np.float32(clamp))
    u = (xc * xc).astype(np.float32)
    acc = _fma32(u, np.full_like(u, q[0]), q[1])
    for c in q[2:]:
        acc = _fma32(acc, u, c)
    cdf = _fma32(xc, acc, 0.5)
    e = np.exp2(((x * x).astype(np.float32) * np.float32(-0.72134752044448170)).astype(np.float32).astype(np.float64)).astype(np.float32)
    g = (x * cdf).astype(np.float32)
    dg = _fma32((xc * np.float32(0.39894228040143268)).astype(np.float32), e, cdf)
    x64 = x.astype(np.float64)
    phi_true = 0.5 * (1.0 + erf(x64 / np.sqrt(2.0)))
    g_true = x64 * phi_true
    dg_true = phi_true + x64 * np.exp(-0.5 * x64 * x64) / np.sqrt(2.0 * np.pi)
    assert np.abs(cdf - phi_true).max() <= 1.3e-5                     # header: <= 1.24e-5 (the Abramowitz-Stegun form it replaced: 2.5e-5)
    assert np.abs(g - g_true).max() <= 1.3e-5 * 12.0 and np.abs(g - g_true)[np.abs(x) <= 4.25].max() <= 6e-5
    assert np.abs(dg - dg_true).max() <= 5e-5                          # the saved derivative is then quantised in steps of 5e-3
    assert cdf.min() >= -1.3e-5 and cdf.max() <= 1.0 + 1.3e-5


def test_scalar_form_carries_the_same_coefficients():
    """``gelu_both`` (one element; the general fallback GEMM's epilogue) restates the polynomial in scalar fmas: same coefficients, same clamp"""
    src = open(HDR).read()
    body = src[src.index("OCN_DEV void gelu_both(float x"):]
    body = body[:body.index("\n}\n")]
    first = re.search(r"fmaf\(u, ([-0-9.e+]+)f, ([-0-9.e+]+)f\)", body)
    rest = re.findall(r"q = fmaf\(q, u, ([-0-9.e+]+)f\);", body)
    coeffs = np.array([float(first.group(1)), float(first.group(2))] + [float(c) for c in rest], dtype=np.float32)
    clamp, q = _header_constants()
    assert np.array_equal(coeffs, q) and f"-{clamp}f, {clamp}f" in body
