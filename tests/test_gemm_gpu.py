"""GPU parity of csam_gemm_f16 against a plain fp32 matmul of the same fp16-rounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, w, bias, act, colscale, residual):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if act == 1:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = torch.relu(y)
    if colscale is not None:
        y = y * colscale
    if residual is not None:
        y = y + residual.float()
    return y


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (4096, 1024, 1024),
                                   (5330, 3072, 1024), (77, 256, 192), (4900, 128, 4096),
                                   (5330, 1024, 1024), (5376, 1024, 4096), (4000, 1536, 256)])   # 96-row tile dispatch
def test_gemm_shapes(cuda, M, N, K):
    from crowdsam_amd import hip
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    # asymmetric, non-symmetric data (transpose-detecting, guide rule 16)
    a = (torch.randn(M, K, generator=g) * 0.5).to(cuda).half()
    w = (torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).view(N, 1) * 1e-4).to(cuda).half()
    out = hip.gemm_f16(a, w, out_dtype=torch.float32)
    ref = _ref(a, w, None, 0, None, None)
    err = (out - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("out_dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("res_dtype", [None, torch.float16, torch.float32])
def test_gemm_epilogue(cuda, act, out_dtype, res_dtype):
    from crowdsam_amd import hip
    M, N, K = 300, 256, 256
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.randn(M, K, generator=g).to(cuda).half()
    w = (torch.randn(N, K, generator=g) * 0.06).to(cuda).half()
    bias = torch.randn(N, generator=g).to(cuda)
    cs = (torch.rand(N, generator=g) + 0.5).to(cuda)
    res = None if res_dtype is None else torch.randn(M, N, generator=g).to(cuda).to(res_dtype)
    out = hip.gemm_f16(a, w, bias=bias, act=act, residual=res, colscale=cs, out_dtype=out_dtype)
    ref = _ref(a, w, bias, act, cs, res)
    tol = 2e-2 if out_dtype == torch.float16 else 2e-3
    assert (out.float() - ref).abs().max().item() < tol


def test_gemm_inplace_residual(cuda):
    from crowdsam_amd import hip
    M, N, K = 4096, 1024, 1024
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randn(M, K, generator=g).to(cuda).half()
    w = (torch.randn(N, K, generator=g) * 0.03).to(cuda).half()
    x = torch.randn(M, N, generator=g).to(cuda)
    ref = x + a.float() @ w.float().t()
    hip.gemm_f16(a, w, out=x, residual=x)
    assert (x - ref).abs().max().item() < 5e-3


def test_gemm_rejects_bad_shapes(cuda):
    from crowdsam_amd import hip
    a = torch.zeros(16, 60, device=cuda, dtype=torch.float16)
    w = torch.zeros(128, 60, device=cuda, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        hip.gemm_f16(a, w)


@pytest.mark.parametrize("M,N,K,act", [(256, 2048, 64, 0), (300, 2048, 128, 1), (4096, 3072, 1024, 0), (5330, 3072, 1024, 0),
                                       (4096, 4096, 1024, 1), (1000, 2304, 4096, 2)])
def test_gemm_pingpong_256(cuda, M, N, K, act):
    """Shapes that dispatch to the 256x256 ping-pong kernel (fp16 out, no residual, N % 256 == 0, one round of tiles):
    ragged M, the shortest K the prologue/tail logic allows (2 stages), GELU / ReLU epilogues."""
    from crowdsam_amd import hip
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(cuda).half()
    w = (torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).view(N, 1) * 1e-4).to(cuda).half()
    bias = torch.randn(N, generator=g).to(cuda)
    out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float16)
    hip.gemm_f16(a, w, out=out, bias=bias, act=act)
    ref = _ref(a, w, bias, act, None, None)
    assert torch.isfinite(out).all()
    err = (out.float() - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K", [(32, 1, 256), (128, 4, 256), (16384, 1, 256), (50, 3, 512), (37, 5, 256), (200, 32, 256),
                                   (9, 70, 96)])
def test_linear_f32_small_heads(cuda, M, N, K):
    """csam_linear_f32: the one-wave-per-row form (N <= 4: IoU / parallel-IoU / one-class classifier heads) and the tiled
    form, with bias, ReLU, residual and a strided A, against fp64 torch."""
    from crowdsam_amd import hip
    g = torch.Generator().manual_seed(M * 131 + N)
    a_full = torch.randn(M, 2 * K, generator=g).to(cuda)
    a = a_full[:, :K]                                                  # row stride 2K
    w, b = (torch.randn(N, K, generator=g) * 0.1).to(cuda), torch.randn(N, generator=g).to(cuda)
    r = torch.randn(M, N, generator=g).to(cuda)
    for act, res in ((hip.ACT_NONE, None), (hip.ACT_RELU, r)):
        out = hip.linear_f32(a, w, b, act=act, residual=res, M=M, lda=a.stride(0))
        ref = a.double() @ w.double().t() + b.double()
        if act == hip.ACT_RELU:
            ref = ref.clamp(min=0)
        if res is not None:
            ref = ref + res.double()
        err = (out.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (M, N, K, err)
