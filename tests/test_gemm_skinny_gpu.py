"""GPU parity: skinny (weight-streaming) GEMM vs an fp32 computation on the same bf16 inputs with the same rounding points
(linear output rounded to bf16, activation on that value rounded again, then the residual add).  Tolerance 1e-4 relative L2 like the tile GEMM; element-wise the two
kernels may differ by one bf16 ulp (different fp32 summation order over K)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16, f32 = torch.bfloat16, torch.float32


def _ref(x, w, bias, act, res, out_f32):
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    y = y.to(bf16).float()
    if act == "gelu_tanh":
        y = torch.nn.functional.gelu(y, approximate="tanh").to(bf16).float()
    elif act == "silu":
        y = torch.nn.functional.silu(y).to(bf16).float()
    if res is not None:
        y = y + res.float()
    return y if out_f32 else y.to(bf16).float()


@pytest.mark.parametrize("M,N,K,bias,act,res,out_f32", [
    (40, 4096, 4096, False, None, None, False),       # UMT5 q/k/v/o
    (40, 10240, 4096, False, "gelu_tanh", None, False),   # wi_0
    (37, 4096, 10240, False, None, "f32", True),       # wo with the fp32 residual stream
    (1, 1536, 1536, True, "silu", None, False),        # time embedding
    (64, 1000, 1024, True, None, "bf16", False),       # ragged N (last strip partial)
    (8, 512, 512 * 3, True, None, None, True),
    (97, 4096, 4096, False, None, "f32", True),         # 65..128 rows: the 128-row variant
    (128, 2048, 1024, True, "gelu_tanh", None, False),
])
def test_skinny_matches_fp32(hip_lib, M, N, K, bias, act, res, out_f32):
    from vist3a_amd import lib as L
    from vist3a_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g).to(bf16)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(bf16)
    b = torch.randn(N, device="cuda", generator=g) if bias else None
    r = None
    if res:
        r = torch.randn(M, N, device="cuda", generator=g).to(f32 if res == "f32" else bf16)
    a = {None: L.ACT_NONE, "gelu_tanh": L.ACT_GELU_TANH, "silu": L.ACT_SILU}[act]
    calls = []
    orig = ops._gemm_skinny
    ops._gemm_skinny = lambda *aa, **kk: (calls.append(1), orig(*aa, **kk))[1]
    try:
        out = ops.gemm(x, w, b, act=a, residual=r, out_f32=out_f32)
    finally:
        ops._gemm_skinny = orig
    assert calls, "shape should dispatch to the skinny kernel"
    ref = _ref(x, w, b, act, r, out_f32)
    rel = ((out.float() - ref).norm() / ref.norm()).item()
    assert out.dtype == (f32 if out_f32 else bf16) and rel < 1e-4, rel
    tile = ops.gemm(x, w, b, act=a, residual=r, out_f32=out_f32, tile=5)   # the tile GEMM on the same problem
    assert ((tile.float() - out.float()).norm() / ref.norm()).item() < 2e-4


def test_skinny_transposed_store(hip_lib):
    """V^T = Wv . X^T form: the weight is the left operand, output [d, tokens], bias per weight row."""
    from vist3a_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    Wv = (torch.randn(4096, 2048, device="cuda", generator=g) / 45).to(bf16)
    X = torch.randn(40, 2048, device="cuda", generator=g).to(bf16)
    bias = torch.randn(4096, device="cuda", generator=g)
    vt = torch.zeros(4096, 64, device="cuda", dtype=bf16)
    ops.gemm(Wv, X, bias, out=vt[:, :40], bias_row=True)
    ref = (Wv.float() @ X.float().t() + bias[:, None]).to(bf16).float()
    assert ((vt[:, :40].float() - ref).norm() / ref.norm()).item() < 1e-4
    assert float(vt[:, 40:].abs().max()) == 0  # nothing written past the logical width


def test_not_dispatched_when_epilogue_features_are_needed(hip_lib):
    from vist3a_amd import ops
    x = torch.randn(16, 1024, device="cuda").to(bf16)
    w = torch.randn(1024, 1024, device="cuda").to(bf16)
    scale = torch.ones(1024, device="cuda")
    calls = []
    orig = ops._gemm_skinny
    ops._gemm_skinny = lambda *a, **k: calls.append(1)
    try:
        ops.gemm(x, w, scale=scale)
    finally:
        ops._gemm_skinny = orig
    assert not calls
