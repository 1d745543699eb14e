"""Developer-build kernel alternatives (libopenclip_hip_dev.so, -DOCN_DEV_BUILD; the product library compiles none of them) against the same
references as the shipped forms.  One subprocess per check: a process loads ONE library (OCN_LIB_PATH)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVLIB = os.path.join(ROOT, "open_clip_amd", "libopenclip_hip_dev.so")

_AS_GELU = r'''
import sys, torch
sys.path.insert(0, %r)
from open_clip_amd import _lib, ops
from tests.test_kernels_gpu import bf, check, check_saved_derivative
dev = torch.device("cuda:0")
_lib.call("ocn_set_gemm_variant", (0x400000 << 8) | 5)  # persistent NT kernel, the Abramowitz-Stegun GELU arithmetic that shipped until round 4
for M, N, K in [(2500, 768, 1024), (4096, 3072, 768), (1000, 640, 320)]:
    g = torch.Generator().manual_seed(M + N)
    a = bf(torch.randn(M, K, generator=g)).to(dev)
    b = bf(torch.randn(N, K, generator=g) * K ** -0.5 * 2.0).to(dev)   # pre-activations out to |x| ~ 9: both sides of the clamp
    bias = torch.randn(N, generator=g).to(dev)
    aux = torch.full((M, N), 255, dtype=torch.uint8, device=dev)
    out = ops.gemm_nt(ops.EPI_BIAS_GELU, a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev), bias=bias, aux=aux)
    pre = (a.float() @ b.float().t() + bias).requires_grad_(True)
    act = torch.nn.functional.gelu(pre)
    act.backward(torch.ones_like(act))
    check(f"dev gemm_nt[{M}x{N}x{K}] gelu(A&S).out", out, act.detach(), bf16_out=True, abs_tol=1e-3)
    check_saved_derivative(f"dev gemm_nt[{M}x{N}x{K}] gelu(A&S).saved_derivative", aux, pre.grad)
print("AS_GELU_OK")
'''


@pytest.mark.gpu
def test_abramowitz_stegun_gelu_epilogue_of_the_developer_build():
    """knob 0x400000 of the persistent NT GEMM in the developer build (csrc/ocn_common.h::gelu_both_as: the erfc form that shipped until round 4,
    kept for A/B timing against the polynomial-CDF form the product library now uses; reference nn.GELU(), transformer.py:295-299): output within
    the bf16 bound and the saved 8-bit derivative within half a step of torch's, like the shipped form (whose own checks are
    tests/test_kernels_gpu.py / tests/test_gemm_bench_shapes_gpu.py on the product library)"""
    if not os.path.exists(DEVLIB):
        pytest.skip("developer library not built (python -m open_clip_amd.build --dev)")
    env = dict(os.environ, OCN_LIB_PATH=DEVLIB)
    r = subprocess.run([sys.executable, "-c", _AS_GELU % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "AS_GELU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
