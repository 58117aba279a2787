"""GPU parity of the persistent image->token kernel (csam_i2t_stream) against a plain PyTorch fp32 statement of
TwoWayAttentionBlock's image->token half (segment_anything_cs/modeling/transformer.py:186-190):
    keys = LayerNorm4(keys + out_proj(softmax(q(keys + pe) k(tokens)^T / sqrt(16)) v(tokens)))
for both forms (projection fused in / hoisted layer-0 Q with a shared source), odd batch sizes (workgroups whose tile
range starts and ends mid-prompt), and against the tile-per-workgroup kernel it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SC = 0.25 * 1.4426950408889634


def _ref(X, q, k, v, Wo, bo, g, be, eps):
    """X [B,T,256] f32 residual, q [B,T,128], k/v [B,7,128] -> [B,T,256]"""
    B, T, _ = X.shape
    qh = q.view(B, T, 8, 16).transpose(1, 2)
    kh = k.view(B, 7, 8, 16).transpose(1, 2)
    vh = v.view(B, 7, 8, 16).transpose(1, 2)
    a = torch.softmax(qh @ kh.transpose(-1, -2) * 0.25, -1) @ vh
    o = a.transpose(1, 2).reshape(B, T, 128) @ Wo.t() + bo
    return torch.nn.functional.layer_norm(X + o, (256,), g, be, eps)


@pytest.mark.parametrize("B", [1, 3, 37])
def test_i2t_stream_projected(cuda, B):
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    X = r(B * T, 256, sc=0.7).half()
    k, v = r(B * 7, 128, sc=0.8), r(B * 7, 128, sc=0.8).half()
    Wq, qpe = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5)
    Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
    g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
    out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
    k_s = (k * SC).half()
    hip.i2t_stream(X, T * 256, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, Wq=Wq, qpe=qpe)
    Xf = X.float().view(B, T, 256)
    q = Xf @ Wq.float().t() + qpe
    ref = _ref(Xf, q, (k_s.float() / SC).view(B, 7, 128), v.float().view(B, 7, 128), Wo.float(), bo, g, be, 1e-5)
    err = (out.float().view(B, T, 256) - ref).abs()
    assert err.max().item() < 2e-2 and err.mean().item() < 1.5e-3, (err.max().item(), err.mean().item())
    # the kernel it replaces (same math, k unscaled, Wo columns permuted for its register chaining)
    from crowdsam_amd.decoder import _kperm
    out2 = torch.zeros_like(out)
    hip.i2t_fused(X, T * 256, k.half(), v, Wo[:, _kperm(128)].contiguous(), bo, g, be, 1e-5, out2, B, T, Wq=Wq, qpe=qpe)
    assert (out.float() - out2.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("B", [2, 19])
def test_i2t_stream_hoisted_q(cuda, B):
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(100 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    X = r(T, 256, sc=0.7).half()                         # shared source, prompt stride 0
    Q = r(T, 128, sc=0.9).half()
    k, v = r(B * 7, 128, sc=0.8), r(B * 7, 128, sc=0.8).half()
    Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
    g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
    out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
    k_s = (k * SC).half()
    hip.i2t_stream(X, 0, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, Q=Q, q_bstride=0)
    Xf = X.float().view(1, T, 256).expand(B, T, 256)
    q = Q.float().view(1, T, 128).expand(B, T, 128)
    ref = _ref(Xf, q, (k_s.float() / SC).view(B, 7, 128), v.float().view(B, 7, 128), Wo.float(), bo, g, be, 1e-5)
    err = (out.float().view(B, T, 256) - ref).abs()
    assert err.max().item() < 2e-2 and err.mean().item() < 1.5e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("B", [1, 2, 19, 530])
def test_i2t_rank_hoisted_q(cuda, B):
    """csam_i2t_rank (rank-56, wave-local form of the hoisted-Q layer) against the same fp32 reference and against
    csam_i2t_stream; bitwise repeatable."""
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(200 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    X = r(T, 256, sc=0.7).half()
    Q = r(T, 128, sc=0.9).half()
    k, v = r(B * 7, 128, sc=0.8), r(B * 7, 128, sc=0.8).half()
    Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
    g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
    k_s = (k * SC).half()
    ws = torch.empty(hip.i2t_rank_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
    out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
    hip.i2t_rank(X, 0, Q, 0, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, ws)
    nref = min(B, 6)                                     # reference on the first prompts and the last one
    idx = list(range(nref - 1)) + [B - 1] if B > 1 else [0]
    sel = torch.tensor(idx, device=cuda)
    Xf = X.float().view(1, T, 256).expand(len(idx), T, 256)
    q = Q.float().view(1, T, 128).expand(len(idx), T, 128)
    ref = _ref(Xf, q, (k_s.float() / SC).view(B, 7, 128)[sel], v.float().view(B, 7, 128)[sel], Wo.float(), bo, g, be, 1e-5)
    err = (out.float().view(B, T, 256)[sel] - ref).abs()
    assert err.max().item() < 2e-2 and err.mean().item() < 1.5e-3, (err.max().item(), err.mean().item())
    out2 = torch.zeros_like(out)
    hip.i2t_stream(X, 0, k_s, v, Wo, bo, g, be, 1e-5, out2, B, T, Q=Q, q_bstride=0)
    d = (out.float() - out2.float()).abs()
    assert d.max().item() < 2e-2 and d.mean().item() < 1e-3, (d.max().item(), d.mean().item())
    out3 = torch.zeros_like(out)
    hip.i2t_rank(X, 0, Q, 0, k_s, v, Wo, bo, g, be, 1e-5, out3, B, T, ws)
    assert torch.equal(out.view(torch.int16), out3.view(torch.int16))


@pytest.mark.parametrize("B", [1, 3, 37, 530])
def test_i2t_rank_projected(cuda, B):
    """csam_i2t_rank_proj (layer-1 form: per-prompt keys, the q projection folded onto 56 back-projected token keys) against
    the fp32 statement and against csam_i2t_stream's projected form; bitwise repeatable."""
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(300 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    nX = min(B, 5)
    Xs = r(nX * T, 256, sc=0.7).half()
    idx = torch.arange(B, device=cuda) % nX
    X = Xs.view(nX, T * 256)[idx].contiguous().view(B * T, 256) if B > nX else Xs
    k, v = r(B * 7, 128, sc=0.8), r(B * 7, 128, sc=0.8).half()
    Wq, qpe = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5)
    Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
    g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
    k_s = (k * SC).half()
    qpe16 = qpe.half()
    ws = torch.empty(hip.i2t_rank_proj_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
    out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
    hip.i2t_rank_proj(X, T * 256, qpe16, Wq, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, ws)
    sel = torch.tensor(sorted(set(list(range(min(B, 4))) + [B - 1])), device=cuda)
    Xf = X.float().view(B, T, 256)[sel]
    q = Xf @ Wq.float().t() + qpe16.float()
    ref = _ref(Xf, q, (k_s.float() / SC).view(B, 7, 128)[sel], v.float().view(B, 7, 128)[sel], Wo.float(), bo, g, be, 1e-5)
    err = (out.float().view(B, T, 256)[sel] - ref).abs()
    assert err.max().item() < 2e-2 and err.mean().item() < 1.5e-3, (err.max().item(), err.mean().item())
    out2 = torch.zeros_like(out)
    hip.i2t_stream(X, T * 256, k_s, v, Wo, bo, g, be, 1e-5, out2, B, T, Wq=Wq, qpe=qpe)
    d = (out.float() - out2.float()).abs()
    assert d.max().item() < 2.5e-2 and d.mean().item() < 1e-3, (d.max().item(), d.mean().item())
    out3 = torch.zeros_like(out)
    hip.i2t_rank_proj(X, T * 256, qpe16, Wq, k_s, v, Wo, bo, g, be, 1e-5, out3, B, T, ws)
    assert torch.equal(out.view(torch.int16), out3.view(torch.int16))


@pytest.mark.parametrize("B", [1, 3, 37, 530])
@pytest.mark.parametrize("proj", [False, True])
def test_i2t_t2i_is_bitwise_the_two_kernels(cuda, B, proj):
    """csam_i2t_t2i (image->token producers + the next block's token->image readers in one pass) must write the keys of
    csam_i2t_rank[_proj] and the Y of csam_t2i_rank over those keys BIT-EXACTLY: the reader half does the same arithmetic
    on the same fp16 key rows in the same order, only from LDS instead of HBM.  Bitwise repeatable."""
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(400 + B + int(proj))
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    k, v = r(B * 7, 128, sc=0.8), r(B * 7, 128, sc=0.8).half()
    Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
    g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
    k_s = (k * SC).half()
    # reader operands (csam_t2i_rank)
    Wk = r(128, 256, sc=0.06).half()
    kpe16 = r(T, 128, sc=0.5).half()
    qs = (r(B * 7, 128, sc=1.2) * SC).half()
    qp = torch.empty(B * 64, 256, dtype=torch.float16, device=cuda)
    out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
    ws = torch.empty(hip.i2t_rank_proj_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
    if proj:
        nX = min(B, 5)
        Xs = r(nX * T, 256, sc=0.7).half()
        idx = torch.arange(B, device=cuda) % nX
        X = Xs.view(nX, T * 256)[idx].contiguous().view(B * T, 256) if B > nX else Xs
        Wq, Q = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5).half()
        hip.i2t_rank_proj(X, T * 256, Q, Wq, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, ws)
        xs = T * 256
    else:
        X, Q, Wq = r(T, 256, sc=0.7).half(), r(T, 128, sc=0.9).half(), None
        hip.i2t_rank(X, 0, Q, 0, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, ws)
        xs = 0
    Y = torch.zeros(B * 7, 2048, dtype=torch.float16, device=cuda)
    hip.t2i_rank(out, Wk, kpe16, qs, qp, Y, B, T)
    wsf = torch.empty(hip.i2t_t2i_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
    for _ in range(2):
        out2 = torch.full_like(out, float("nan"))
        Y2 = torch.full_like(Y, float("nan"))
        hip.i2t_t2i(X, xs, Q, 0, Wq, k_s, v, Wo, bo, g, be, 1e-5, out2, Wk, kpe16, qs, Y2, B, T, wsf)
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), out2.view(torch.int16)), \
            (out.float() - out2.float()).abs().max().item()
        bad = (Y.view(torch.int16) != Y2.view(torch.int16))
        assert not bad.any(), (int(bad.sum()), (Y.float() - Y2.float()).abs().max().item(),
                               bad.view(B, 7, 8, 256).any(-1).nonzero()[:8].tolist())
