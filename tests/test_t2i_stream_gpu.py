"""GPU parity of the persistent token->image kernel (csam_t2i_stream) against a plain PyTorch fp32 statement of
Attention.forward with the K/V projections of the key state (segment_anything_cs/modeling/transformer.py:173-177,
105-112, 185-232):  out = softmax(q_h (keys Wk^T + pe Wk^T + bk)_h^T / 4) (keys Wv^T + bv)_h  per head, and against the
tile-per-workgroup kernel + merge it replaces.  Odd batch sizes exercise ragged prompt ranges per workgroup."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [1, 5, 600, 1031])
def test_t2i_stream(cuda, B):
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    nX = min(B, 8)                                        # distinct key states (reference stays small)
    X = r(nX * T, 256, sc=0.7).half()
    if B > nX:
        Xall = X.view(nX, T * 256)[torch.arange(B, device=cuda) % nX].contiguous().view(B * T, 256)
    else:
        Xall = X
    Wkv = r(256, 256, sc=0.06).half()
    kpe, bv = r(T, 128, sc=0.5), r(128, sc=0.3)
    q = r(B * 7, 128, sc=1.2).half()
    out = torch.zeros(B * 7, 128, dtype=torch.float16, device=cuda)
    hip.t2i_stream(q, out, B, Xall, Wkv, kpe, bv, T)
    Xf = X.float().view(nX, T, 256)
    K = (Xf @ Wkv[:128].float().t() + kpe).view(nX, T, 8, 16).transpose(1, 2)          # [nX,8,T,16]
    V = (Xf @ Wkv[128:].float().t() + bv).view(nX, T, 8, 16).transpose(1, 2)
    idx = torch.arange(B, device=cuda) % nX
    qh = q.float().view(B, 7, 8, 16).transpose(1, 2)                                    # [B,8,7,16]
    ref = torch.empty(B, 7, 128, device=cuda)
    for i in range(nX):
        sel = (idx == i).nonzero().flatten()
        if sel.numel() == 0:
            continue
        a = torch.softmax(qh[sel] @ K[i].transpose(-1, -2) * 0.25, -1) @ V[i]          # [n,8,7,16]
        ref[sel] = a.transpose(1, 2).reshape(-1, 7, 128)
    err = (out.float().view(B, 7, 128) - ref).abs()
    assert err.max().item() < 6e-3 and err.mean().item() < 5e-4, (err.max().item(), err.mean().item())
    ws = torch.empty(hip.attn_t2i_workspace_bytes(B, 8) // 4 + B * 32 * 56 * 18, dtype=torch.float32, device=cuda)
    out2 = torch.zeros_like(out)
    hip.t2i_fused(q, out2, B, ws, X=Xall, Wkv=Wkv, kpe=kpe, bv=bv)
    assert (out.float() - out2.float()).abs().max().item() < 6e-3
    # bitwise repeatable
    out3 = torch.zeros_like(out)
    hip.t2i_stream(q, out3, B, Xall, Wkv, kpe, bv, T)
    assert torch.equal(out.view(torch.int16), out3.view(torch.int16))


@pytest.mark.parametrize("B", [1, 5, 300, 1031])
def test_t2i_rank_matches_attention_statement(cuda, B):
    """csam_t2i_rank (rank-56 form: back-projected queries, weighted sums of the RAW keys, Wv + out_proj afterwards as one
    GEMM with host-folded weights) against the same fp32 statement of Attention.forward (transformer.py:228-254) INCLUDING
    the output projection, which the rank form cannot be separated from."""
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(100 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    nX = min(B, 6)
    X = r(nX * T, 256, sc=0.7).half()
    idx = torch.arange(B, device=cuda) % nX
    Xall = X.view(nX, T * 256)[idx].contiguous().view(B * T, 256) if B > nX else X
    Wk, Wv = r(128, 256, sc=0.06).half(), r(128, 256, sc=0.06).half()
    bk, bv = r(128, sc=0.3), r(128, sc=0.3)
    Wo, bo = r(256, 128, sc=0.1).half(), r(256, sc=0.2)
    pe_k = r(T, 128, sc=0.5)                              # pe Wk^T (fp32 table of the stream kernel, without bk)
    q = r(B * 7, 128, sc=1.2)
    sc = 0.25 * 1.4426950408889634
    qs = (q * sc).half()
    qp = torch.empty(B * 64, 256, dtype=torch.float16, device=cuda)
    Y = torch.full((B * 7, 2048), float("nan"), dtype=torch.float16, device=cuda)
    hip.t2i_rank(Xall, Wk, pe_k.half(), qs, qp, Y, B, T)
    wc = torch.einsum("ohd,hdk->ohk", Wo.float().view(256, 8, 16), Wv.float().view(8, 16, 256)).reshape(256, 2048)
    got = Y.float() @ wc.t() + (bo + Wo.float() @ bv)
    assert torch.isfinite(got).all()
    Xf = X.float().view(nX, T, 256)
    K = (Xf @ Wk.float().t() + pe_k.half().float() + bk).view(nX, T, 8, 16).transpose(1, 2)
    V = (Xf @ Wv.float().t() + bv).view(nX, T, 8, 16).transpose(1, 2)
    qh = (qs.float() / sc).view(B, 7, 8, 16).transpose(1, 2)
    ref = torch.empty(B, 7, 256, device=cuda)
    for i in range(nX):
        sel = (idx == i).nonzero().flatten()
        if sel.numel():
            a = torch.softmax(qh[sel] @ K[i].transpose(-1, -2) * 0.25, -1) @ V[i]
            ref[sel] = a.transpose(1, 2).reshape(-1, 7, 128) @ Wo.float().t() + bo
    err = (got.view(B, 7, 256) - ref).abs()
    scale = ref.abs().mean().item()
    assert err.max().item() < 0.02 * max(scale, 1.0) and err.mean().item() < 2e-3 * max(scale, 1.0), \
        (err.max().item(), err.mean().item(), scale)
    Y2 = torch.empty_like(Y)
    hip.t2i_rank(Xall, Wk, pe_k.half(), qs, qp, Y2, B, T)
    assert torch.equal(Y.view(torch.int16), Y2.view(torch.int16))             # bitwise repeatable
