"""Size-independent properties of the attention kernels: transformations that provably leave softmax attention unchanged
(a per-query constant added to all scores, a permutation of the keys) must leave the kernels' outputs unchanged up to fp16
rounding.  They catch what a random-data comparison within fp16 tolerances can miss: a slip confined to one register, one
lane group or one key position (tests/test_encoder_gpu.py::test_flash_attn_constant_bias_is_a_no_op is the one that found such
a slip in round 2)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SC = 0.25 * 1.4426950408889634


def test_win_attn_position_independent_tables_are_a_no_op(cuda):
    """rel_pos tables whose rows are all the same vector give Th[q, kh] = q . r_h for every kh (and likewise Tw): a per-query
    constant, so the windowed attention must equal the one with zero tables."""
    from crowdsam_amd import hip
    nH, D = 2, 128
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(4096, 3 * D, generator=g).to(cuda).half()
    bias = torch.randn(3 * D, generator=g).to(cuda)
    outs = []
    for scale_tab in (0.0, 1.0, -2.0):
        rh = (torch.randn(1, 64, generator=g) * 0.3 * scale_tab).expand(27, 64).contiguous().to(cuda)
        rw = (torch.randn(1, 64, generator=g) * 0.3 * scale_tab).expand(27, 64).contiguous().to(cuda)
        out = torch.zeros(4096, D, device=cuda, dtype=torch.float16)
        hip.win_attn(qkv, bias, hip.relcat_window(rh, rw), out, D, nH, 0.125)
        outs.append(out.float())
    for o in outs[1:]:
        assert (o - outs[0]).abs().max().item() < 3e-3, (o - outs[0]).abs().max().item()


@pytest.mark.parametrize("T", [4096, 5330])
def test_flash_attn_key_permutation(cuda, T):
    """Keys (K and V rows together) in another order: other tiles, other lanes, other ring slots -- same attention."""
    from crowdsam_amd import hip
    nH, D = 2, 128
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(T, 3 * D, generator=g).to(cuda).half()
    qkv[:, D:2 * D] *= 1.5                                     # a peaked softmax: single keys matter
    perm = torch.randperm(T, generator=g).to(cuda)
    qkv_p = qkv.clone()
    qkv_p[:, D:] = qkv[perm][:, D:]                            # q rows stay, k / v rows are permuted together
    a, b = (torch.zeros(T, D, device=cuda, dtype=torch.float16) for _ in range(2))
    hip.flash_attn(qkv, a, T, nH, 0.125, D)
    hip.flash_attn(qkv_p, b, T, nH, 0.125, D)
    d = (a.float() - b.float()).abs()
    assert d.max().item() < 4e-3 and d.mean().item() < 2e-4, (d.max().item(), d.mean().item())


@pytest.mark.parametrize("B", [3, 300])
def test_t2i_rank_score_shift_and_key_permutation(cuda, B):
    """csam_t2i_rank: (a) the same vector added to every key_pe row shifts all scores of a (query, head) by q . v -- no-op;
    (b) keys (X rows with their key_pe rows) permuted -- no-op."""
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    X = r(B * T, 256, sc=0.7).half()
    Wk = r(128, 256, sc=0.06).half()
    kpe = r(T, 128, sc=0.5)
    qs = (r(B * 7, 128, sc=1.2) * SC).half()
    qp = torch.empty(B * 64, 256, dtype=torch.float16, device=cuda)

    def run(Xin, kp):
        Y = torch.empty(B * 7, 2048, dtype=torch.float16, device=cuda)
        hip.t2i_rank(Xin, Wk, kp.half().contiguous(), qs, qp, Y, B, T)
        return Y.float()

    y0 = run(X, kpe)
    y1 = run(X, kpe + r(1, 128, sc=0.7))
    perm = torch.randperm(T, generator=gen).to(cuda)
    y2 = run(X.view(B, T, 256)[:, perm].reshape(B * T, 256).contiguous(), kpe[perm])
    for y in (y1, y2):
        d = (y - y0).abs()
        assert d.max().item() < 6e-3 and d.mean().item() < 3e-4, (d.max().item(), d.mean().item())


@pytest.mark.parametrize("B", [2, 270])
def test_i2t_rank_token_key_permutation(cuda, B):
    """csam_i2t_rank / csam_i2t_rank_proj: the 7 token keys and values of a prompt in another order -- other slots of the
    paired-head MFMA operands, other rows of M_b / Kp_b -- same softmax, same output."""
    from crowdsam_amd import hip
    T = 4096
    gen = torch.Generator().manual_seed(40 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
    X0, Q0 = r(T, 256, sc=0.7).half(), r(T, 128, sc=0.9).half()
    X = r(B * T, 256, sc=0.7).half()
    Wq, qpe16 = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5).half()
    k, v = (r(B, 7, 128, sc=0.8) * SC).half(), r(B, 7, 128, sc=0.8).half()
    Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
    g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
    p7 = torch.tensor([3, 0, 6, 1, 5, 2, 4], device=cuda)
    ws = torch.empty(hip.i2t_rank_proj_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
    outs = []
    for kk, vv in ((k, v), (k[:, p7].contiguous(), v[:, p7].contiguous())):
        o0 = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
        o1 = torch.zeros_like(o0)
        hip.i2t_rank(X0, 0, Q0, 0, kk.view(B * 7, 128), vv.view(B * 7, 128), Wo, bo, g, be, 1e-5, o0, B, T, ws)
        hip.i2t_rank_proj(X, T * 256, qpe16, Wq, kk.view(B * 7, 128), vv.view(B * 7, 128), Wo, bo, g, be, 1e-5, o1, B, T, ws)
        outs.append((o0.float(), o1.float()))
    for a, b in zip(outs[0], outs[1]):
        d = (a - b).abs()
        assert d.max().item() < 1.6e-2 and d.mean().item() < 3e-4, (d.max().item(), d.mean().item())
