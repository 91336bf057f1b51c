"""The 256 x 128-tile form of the 4-wave hand-scheduled GEMM (gemm_w4.hip NJ = 4, gemm_w4n_loop_*.inc): taken for outputs whose 256 x 256
tiling fills at most half a wave of workgroups (micro-batches of 4 sequences: o_net, ff2, the data gradients).  All three operand
layouts, bf16 / fp32 outputs, bias, accumulation onto C, against the strided fp32-MFMA kernel on the same bf16 operands."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
DEV = "cuda"


def _operands(layout, M, N, K, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    r = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    if layout == "nt":
        return r(M, K), r(N, K).t()
    if layout == "nn":
        return r(M, K), r(K, N)
    return r(K, M).t(), r(K, N)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", [(4096, 2048, 2048), (4096, 2048, 4096), (4096, 2048, 8192), (3584, 2176, 384)])
def test_w4n_matches_the_strided_kernel(layout, M, N, K):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bdm_db1_amd import ops, lib
    lib.set_knob("gemm_halfwave", 1 << 20)      # (keep the half-wave split-K out of the way: this test is about the 256 x 128 kernel)
    try:
        a, b = _operands(layout, M, N, K, 3)
        for odt, beta, with_bias in ((torch.bfloat16, 0.0, True), (torch.bfloat16, 0.75, False), (torch.float32, 0.0, False), (torch.float32, 1.0, True)):
            bias = (torch.randn(N, device=DEV) * 0.3).to(torch.bfloat16) if with_bias else None
            c0 = (torch.randn(M, N, device=DEV) * 0.2).to(odt)
            got, want = c0.clone(), c0.clone()
            name, sk, _ = ops.gemm_kernel_choice(a, b, got, beta=beta)
            assert name == "w4n" and not sk, (name, sk, layout, M, N, K, odt, beta)
            ops.gemm(a, b, got, bias=bias, beta=beta)
            ops.gemm_force_generic(True)
            ops.gemm(a, b, want, bias=bias, beta=beta)
            ops.gemm_force_generic(False)
            err = float((got.double() - want.double()).abs().max() / want.double().abs().max())
            assert err <= (6e-3 if odt == torch.bfloat16 else 2e-5), (layout, M, N, K, odt, beta, err)
    finally:
        ops.gemm_force_generic(False)
        lib.load().db1_test_clear_knobs()
