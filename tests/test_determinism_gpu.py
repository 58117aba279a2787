"""Bitwise repeatability of the LDS-ring kernels.  A counted `s_waitcnt vmcnt(N)` that retires an LDS-DMA stage
which other waves read in the same phase was found racy in round 1 (non-repeatable V rows in the token->image
kernel, all within the parity tolerances): every ring now retires a stage one iteration before its first read.
These tests fail on such a race even when the numerical error stays inside the tolerances."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same(a, b):
    it = torch.int16 if a.element_size() == 2 else torch.int32
    return bool((a.contiguous().view(-1).view(it) == b.contiguous().view(-1).view(it)).all())


def _repeat(fn, n=4):
    outs = []
    for _ in range(n):
        outs.append(fn().clone())
        torch.cuda.synchronize()
    return all(_same(outs[0], o) for o in outs[1:])


def test_fused_decoder_kernels_repeatable(cuda):
    from crowdsam_amd import hip
    torch.manual_seed(0)
    B = 384                                      # 805 MB of key state: never cache resident, loads stay "cold"
    X = (torch.randn(B * 4096, 256, device=cuda) * 0.5).half()
    Wkv = (torch.randn(256, 256, device=cuda) * 0.05).half()
    kpe, bv = torch.randn(4096, 128, device=cuda), torch.randn(128, device=cuda)
    q = (torch.randn(B * 7, 128, device=cuda) * 0.5).half()
    ws = torch.empty(hip.attn_t2i_workspace_bytes(B, 8) // 4, dtype=torch.float32, device=cuda)
    o1 = torch.zeros(B * 7, 128, dtype=torch.float16, device=cuda)
    assert _repeat(lambda: hip.t2i_fused(q, o1, B, ws, X=X, Wkv=Wkv, kpe=kpe, bv=bv))
    k, v = (torch.randn(B * 7, 128, device=cuda) * 0.5).half(), (torch.randn(B * 7, 128, device=cuda) * 0.5).half()
    Wq, qpe = (torch.randn(128, 256, device=cuda) * 0.05).half(), torch.randn(4096, 128, device=cuda)
    Wo, bo = (torch.randn(256, 128, device=cuda) * 0.05).half(), torch.randn(256, device=cuda)
    g, be = torch.ones(256, device=cuda), torch.zeros(256, device=cuda)
    o2 = torch.zeros(B * 4096, 256, dtype=torch.float16, device=cuda)
    assert _repeat(lambda: hip.i2t_fused(X, 4096 * 256, k, v, Wo, bo, g, be, 1e-5, o2, B, 4096, Wq=Wq, qpe=qpe))
    assert _repeat(lambda: hip.i2t_stream(X, 4096 * 256, k, v, Wo, bo, g, be, 1e-5, o2, B, 4096, Wq=Wq, qpe=qpe))
    W1, b1 = (torch.randn(256, 256, device=cuda) * 0.05).half(), torch.randn(256, device=cuda)
    W2, b2 = (torch.randn(128, 64, device=cuda) * 0.1).half(), torch.randn(128, device=cuda)
    hy, masks = torch.randn(B, 4, 32, device=cuda), torch.empty(B, 4, 256, 256, device=cuda)
    stats = torch.empty(B * 4, 2, device=cuda)
    assert _repeat(lambda: (hip.upscale_fused(X, W1, b1, torch.ones(64, device=cuda), torch.zeros(64, device=cuda), 1e-6,
                                              W2, b2, hy, masks, B, stats=stats), masks)[1])


@pytest.mark.parametrize("M,N,K", [(4096, 3072, 1024), (5330, 4096, 1024), (4096, 1024, 4096), (8192, 4096, 4096),
                                   (5330, 1024, 4096)])
def test_gemm_repeatable(cuda, M, N, K):
    from crowdsam_amd import hip
    torch.manual_seed(1)
    a = torch.randn(M, K, device=cuda).half()
    w = (torch.randn(N, K, device=cuda) * 0.05).half()
    out = torch.empty(M, N, device=cuda, dtype=torch.float16)
    assert _repeat(lambda: hip.gemm_f16(a, w, out=out))


def test_flash_repeatable(cuda):
    from crowdsam_amd import hip
    torch.manual_seed(2)
    T, nH = 5330, 16
    qkv = torch.randn(T, 3 * nH * 64, device=cuda).half()
    out = torch.empty(T, nH * 64, device=cuda, dtype=torch.float16)
    assert _repeat(lambda: hip.flash_attn(qkv, out, T, nH, 0.125, nH * 64))


def test_flash_relpos_repeatable_many_launches(cuda):
    """Round 3: the rel-pos variant lost one Tw seed term in one wave's lane group 3 at key tile 0 in 1-8 % of the
    launches (all inside the parity tolerances).  250 launches on the SAM global-block shape must be bit-identical."""
    from crowdsam_amd import hip
    torch.manual_seed(4)
    T, nH = 4096, 16
    qkv = torch.randn(T, 3 * nH * 64, device=cuda).half()
    traw = torch.randn(nH, T, 256, device=cuda) * 0.3
    outs = []
    for _ in range(250):
        o = torch.empty(T, nH * 64, device=cuda, dtype=torch.float16)
        hip.flash_attn(qkv, o, T, nH, 0.125, nH * 64, relpos=traw, q_prescaled=True)
        outs.append(o)
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not _same(outs[0], o)]
    assert not bad, "launches that differ from the first: %s" % bad[:10]


@pytest.mark.parametrize("hd", [64, 80])
def test_win_attn_repeatable_many_launches(cuda, hd):
    from crowdsam_amd import hip
    torch.manual_seed(5)
    nH = 16
    D = nH * hd
    qkv = torch.randn(4096, 3 * D, device=cuda).half()
    b = torch.randn(3 * D, device=cuda)
    rc = hip.relcat_window(torch.randn(27, hd, device=cuda), torch.randn(27, hd, device=cuda))
    outs = []
    for _ in range(250):
        o = torch.empty(4096, D, device=cuda, dtype=torch.float16)
        hip.win_attn(qkv, b, rc, o, D, nH, hd ** -0.5)
        outs.append(o)
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not _same(outs[0], o)]
    assert not bad, "launches that differ from the first: %s" % bad[:10]


def test_decoder_batch_repeatable(cuda):
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    B = 512
    sd = synth.make_state_dict(list(synth.sam_param_specs(128, 4, 2, (1, 3))), 0)
    plan = DecoderPlan(sd, cuda, 1, B)
    torch.manual_seed(0)
    feat = torch.randn(4096, 256, device=cuda)
    dtok = torch.zeros(5376, 1024, dtype=torch.float16, device=cuda)
    dtok[:5329] = torch.randn(5329, 1024, device=cuda).half()
    plan.set_image(feat, dtok)
    coords = torch.rand(B, 2, device=cuda) * 1023
    ref = None
    for _ in range(4):
        plan.run_batch(coords)
        torch.cuda.synchronize()
        cur = {n: plan.ws[n][: (B if n in ("masks", "iou") else B * 4)].clone() for n in ("masks", "iou", "cls")}
        if ref is None:
            ref = cur
        else:
            for n in cur:
                assert _same(ref[n], cur[n]), n


def test_decoder_batch_4096_soak_250_launches(cuda):
    """250 launches of the FULL production decoder batch (4096 prompts -- csam_i2t_t2i, csam_upscale_stream,
    the pooled PWD-Net heads; 8.6 GB of key state, never cache resident) must agree bit for bit.  Every launch is reduced to
    integer checksums of the bit patterns of the low-res logits, IoU and class outputs on the device (an exact comparison
    of 1 GB per launch without keeping 250 copies)."""
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    B = 4096
    sd = synth.make_state_dict(list(synth.sam_param_specs(128, 4, 2, (1, 3))), 0)
    plan = DecoderPlan(sd, cuda, 1, B)
    torch.manual_seed(2)
    feat = torch.randn(4096, 256, device=cuda)
    dtok = torch.zeros(5376, 1024, dtype=torch.float16, device=cuda)
    dtok[:5329] = torch.randn(5329, 1024, device=cuda).half()
    plan.set_image(feat, dtok)
    coords = torch.rand(B, 2, device=cuda) * 1023
    w = torch.arange(1, 1 + 4 * 256 * 256, device=cuda, dtype=torch.int64) % 1021 + 1       # position-dependent weights

    def checksum():
        out = []
        for n in ("masks", "iou", "cls"):
            t = plan.ws[n][: (B if n in ("masks", "iou") else B * 4)].contiguous().view(-1).view(torch.int32).to(torch.int64)
            k = t.numel()
            out.append((t * w[:k] if k <= w.numel() else t.view(B, -1) * w[None, : t.numel() // B]).sum())
        return torch.stack(out)

    sums = []
    for _ in range(250):
        plan.run_batch(coords)
        sums.append(checksum())
    torch.cuda.synchronize()
    sums = torch.stack(sums).cpu()
    bad = [i for i in range(len(sums)) if not torch.equal(sums[i], sums[0])]
    assert not bad, "launches whose outputs differ from the first: %s" % bad[:10]


def test_flash_attn80_relpos_repeatable_many_launches(cuda):
    """head_dim 80 (ViT-H global blocks): 250 launches of the rel-pos variant on one set of operands must agree bit for bit
    (the head_dim-64 kernel's intermittent seed error, HISTORY.md 4.2b, was only visible this way)."""
    from crowdsam_amd import hip
    torch.manual_seed(9)
    nH, D, T = 16, 1280, 4096
    qkv = torch.randn(T, 3 * D, device=cuda).half()
    rc = hip.relcat_global80(torch.randn(127, 80, device=cuda) * 0.3, torch.randn(127, 80, device=cuda) * 0.3)
    traw = torch.empty(nH, T, 256, device=cuda)
    hip.relpos_raw80(qkv, rc, traw, nH)
    outs = []
    for _ in range(250):
        o = torch.empty(T, D, device=cuda, dtype=torch.float16)
        hip.flash_attn80(qkv, o, T, nH, 80 ** -0.5, D, relpos=traw)
        outs.append(o)
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not _same(outs[0], o)]
    assert not bad, "launches that differ from the first: %s" % bad[:10]
