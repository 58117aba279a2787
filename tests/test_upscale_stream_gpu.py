"""GPU parity of the persistent upscaler (csam_upscale_stream) against the tile-per-workgroup kernel it replaces
(csam_upscale_fused, itself checked against the oracle's mask_decoder.py:172-181 restatement in test_decoder_gpu.py):
same weights, same key state, odd batch sizes (ragged prompt ranges per workgroup), logits and per-plane maxima."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [1, 3, 530])
def test_upscale_stream_matches_fused(cuda, B):
    from crowdsam_amd import hip
    gen = torch.Generator().manual_seed(10 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    nX = min(B, 4)
    X = r(nX * 4096, 256, sc=0.8).half()
    if B > nX:
        X = X.view(nX, -1)[torch.arange(B, device=cuda) % nX].contiguous().view(B * 4096, 256)
    W1, b1 = r(256, 256, sc=0.06).half(), r(256, sc=0.3)
    g, be = (torch.rand(64, generator=gen) + 0.5).to(cuda), r(64, sc=0.2)
    W2, b2 = r(128, 64, sc=0.15).half(), r(128, sc=0.3)
    hy = r(B, 4, 32, sc=0.7)
    m1 = torch.empty(B, 4, 256, 256, device=cuda)
    m2 = torch.full((B, 4, 256, 256), float("nan"), device=cuda)
    s1 = torch.empty(B * 4, 2, device=cuda)
    s2 = torch.full((B * 4, 2), float("nan"), device=cuda)
    hip.upscale_fused(X, W1, b1, g, be, 1e-6, W2, b2, hy, m1, B, stats=s1)
    hip.upscale_stream(X, W1, b1, g, be, 1e-6, W2, b2, hy, m2, B, stats=s2)
    assert torch.isfinite(m2).all()
    err = (m1 - m2).abs()
    scale = m1.abs().mean().item()
    assert err.max().item() < 2e-2 * max(1.0, scale) and err.mean().item() < 2e-4 * max(1.0, scale), \
        (err.max().item(), err.mean().item(), scale)
    assert torch.allclose(s2[:, 0], m2.view(B * 4, -1).max(1).values)          # its own plane maxima, exactly
    assert (s2[:, 1] == 0).all()
    m3 = torch.empty_like(m2)
    hip.upscale_stream(X, W1, b1, g, be, 1e-6, W2, b2, hy, m3, B, stats=None)
    assert torch.equal(m2.view(torch.int32), m3.view(torch.int32))             # bitwise repeatable
