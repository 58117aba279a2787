"""GPU parity of the HIP prompt-encoder + two-way decoder + PWD-Net heads against the reference's own
outputs (tests/golden/decoder_test128.npz, captured by oracle/make_goldens.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _inputs():
    rs = np.random.RandomState(11)
    emb = torch.from_numpy(rs.standard_normal((1, 256, 64, 64)).astype(np.float32))
    dino = torch.from_numpy(rs.standard_normal((1, 73, 73, 1024)).astype(np.float32))
    pts = rs.randint(0, 1024, size=(5, 1, 2)).astype(np.float64)
    return emb, dino, pts


@pytest.fixture(scope="module")
def plan(cuda):
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    sd = synth.make_sam_state_dict("vit_test128")
    return DecoderPlan(sd, cuda, n_class=1, max_batch=8)


def _set_image(plan, cuda):
    emb, dino, pts = _inputs()
    feat = emb[0].permute(1, 2, 0).reshape(4096, 256).contiguous().to(cuda)
    dtok = torch.zeros(5376, 1024, dtype=torch.float16, device=cuda)
    dtok[:5329] = dino.reshape(5329, 1024).to(cuda).half()
    plan.set_image(feat, dtok)
    return torch.from_numpy(pts[:, 0, :].astype(np.float32)).to(cuda).contiguous()


def test_decoder_vs_reference_golden(plan, cuda):
    g = np.load(os.path.join(G, "decoder_test128.npz"))
    coords = _set_image(plan, cuda)
    masks, iou, cls = plan.run_batch(coords)
    torch.cuda.synchronize()
    low = masks.cpu().numpy()
    ref = g["low_sample"]
    err = np.abs(low[:, :, ::8, ::8] - ref)
    scale = np.abs(ref).mean()
    print("low-res logits: mean|ref|=%.3f max err=%.4f mean err=%.5f" % (scale, err.max(), err.mean()))
    # fp16 operands / fp32 accumulate against the fp32 reference: mean < 0.5 %, max < 5 % of the mean |logit|
    assert err.mean() < 0.005 * scale and err.max() < 0.05 * scale
    np.testing.assert_allclose(low.astype(np.float64).sum((2, 3)), g["low_sum"], rtol=0, atol=0.005 * scale * 65536)
    e_iou = np.abs(iou.cpu().numpy() - g["iou"]).max()
    e_cls = np.abs(cls.cpu().numpy() - g["cls"]).max()
    print("iou err %.5f  cls err %.5f" % (e_iou, e_cls))
    assert e_iou < 5e-3 and e_cls < 5e-3
    # dense PE and point tokens are fp32 kernels: tight tolerance
    pe = plan.pe.view(64, 64, 256).permute(2, 0, 1)[None].cpu().numpy()
    np.testing.assert_allclose(pe[:, ::8, ::4, ::4], g["dense_pe_sample"], rtol=0, atol=2e-5)
    tok = plan.ws["tokens0"][:35].view(5, 7, 256)[:, 5:7].cpu().numpy()
    np.testing.assert_allclose(tok, g["sparse"], rtol=0, atol=2e-5)


def _boxes():
    rs = np.random.RandomState(21)
    a = rs.randint(0, 700, size=(6, 2)).astype(np.float64)
    wh = rs.randint(40, 320, size=(6, 2)).astype(np.float64)
    return np.concatenate([a, np.minimum(a + wh, 1023.0)], 1)


def test_box_prompts_vs_reference_golden(plan, cuda):
    """Box prompts (predictor.py:214-292 `boxes`; prompt_encoder.py:95-102: corner tokens PE + point_embeddings[2 / 3], no
    padding point -- seven tokens per prompt like the one-point form) through the same fused decoder, against the REFERENCE's
    prompt encoder + mask decoder on six seeded boxes (tests/golden/decoder_box_test128.npz, oracle/make_goldens.py)."""
    g = np.load(os.path.join(G, "decoder_box_test128.npz"))
    _set_image(plan, cuda)
    boxes = torch.from_numpy(_boxes().astype(np.float32)).to(cuda).contiguous()
    masks, iou, cls = plan.run_batch(None, boxes_f32=boxes)
    torch.cuda.synchronize()
    tok = plan.ws["tokens0"][:42].view(6, 7, 256)[:, 5:7].cpu().numpy()
    np.testing.assert_allclose(tok, g["sparse"], rtol=0, atol=2e-5)               # the two corner tokens (fp32 kernel)
    low = masks.cpu().numpy()
    ref = g["low_sample"]
    err = np.abs(low[:, :, ::8, ::8] - ref)
    scale = np.abs(ref).mean()
    print("box prompts, low-res logits: mean|ref|=%.3f max err=%.4f mean err=%.5f" % (scale, err.max(), err.mean()))
    assert err.mean() < 0.005 * scale and err.max() < 0.05 * scale
    np.testing.assert_allclose(low.astype(np.float64).sum((2, 3)), g["low_sum"], rtol=0, atol=0.005 * scale * 65536)
    assert np.abs(iou.cpu().numpy() - g["iou"]).max() < 5e-3 and np.abs(cls.cpu().numpy() - g["cls"]).max() < 5e-3
    # a point batch after a box batch of the same size replays ITS graph (the graph key carries the prompt kind)
    coords = torch.from_numpy(np.random.RandomState(3).randint(0, 1024, size=(6, 2)).astype(np.float32)).to(cuda)
    m_pt = plan.run_batch(coords)[0].clone()
    m_bx = plan.run_batch(None, boxes_f32=boxes)[0].clone()
    assert torch.equal(m_bx, masks) and not torch.equal(m_pt, m_bx)


def test_labelled_point_tokens_vs_oracle(plan, cuda):
    """Point prompts with labels 1 / 0 / -1 (prompt_encoder.py:88-92: foreground, background, not-a-point): the token kernel
    against the oracle's embed_points (pinned to the reference by test_oracle_vs_golden), and a background prompt decodes to
    something else than the foreground prompt at the same pixel."""
    from crowdsam_amd import synth
    from oracle import sam_oracle as so
    sd = synth.make_sam_state_dict("vit_test128")
    coords = _set_image(plan, cuda)
    labels = torch.tensor([1, 0, -1, 0, 1], dtype=torch.int32)
    m_lab = plan.run_batch(coords, labels_i32=labels.to(cuda))[0].clone()
    tok = plan.ws["tokens0"][:35].view(5, 7, 256)[:, 5:7].cpu().numpy()
    ref = so.embed_points(sd, coords.cpu().double()[:, None, :], labels[:, None]).numpy()
    np.testing.assert_allclose(tok, ref, rtol=0, atol=2e-5)
    m_fg = plan.run_batch(coords)[0].clone()
    assert torch.equal(m_lab[0], m_fg[0]) and torch.equal(m_lab[4], m_fg[4])          # label-1 prompts: the same decode
    assert not torch.equal(m_lab[1], m_fg[1])


def test_fg_prior_vs_reference_golden(plan, cuda):
    g = np.load(os.path.join(G, "decoder_test128.npz"))
    _set_image(plan, cuda)
    logits = plan.fg_logits()                                    # [5329, 1]
    fg = torch.nn.functional.interpolate(logits.view(1, 73, 73, 1).permute(0, 3, 1, 2).cpu(), (256, 256),
                                         mode="bilinear")      # test-side resample of the reference's 2nd step
    err = np.abs(fg[:, :, ::8, ::8].numpy() - g["fg_sample"]).max()
    assert err < 2e-2, err


def test_batch_invariance(plan, cuda):
    """The same prompt must decode identically alone and inside a larger batch (no cross-prompt leakage)."""
    coords = _set_image(plan, cuda)
    m_all, iou_all, cls_all = [t.clone() for t in plan.run_batch(coords)]
    m_one, iou_one, cls_one = plan.run_batch(coords[2:3].contiguous())
    assert (m_all[2] - m_one[0]).abs().max().item() < 1e-3
    assert (iou_all[2] - iou_one[0]).abs().max().item() < 1e-4


def test_large_batch_stream_kernels_match_small_batch_path(cuda):
    """One 320-prompt batch runs the persistent token->image / upscaler kernels (>= 256 prompts); the same prompts in
    chunks of 64 run the tile-per-workgroup kernels the reference golden above pins.  Both must give the same
    low-res logits, IoU and class outputs (different kernels, same arithmetic up to fp16 / accumulation order)."""
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    sd = synth.make_sam_state_dict("vit_test128")
    big = DecoderPlan(sd, cuda, n_class=1, max_batch=320)
    _set_image(big, cuda)
    assert big.t2i_stream and big.up_stream and big.i2t_stream
    rs = np.random.RandomState(5)
    coords = torch.from_numpy(rs.uniform(0, 1023, size=(320, 2)).astype(np.float32)).to(cuda)
    m_big, iou_big, cls_big = [t.clone() for t in big.run_batch(coords)]
    for c0 in range(0, 320, 64):
        m, iou, cls = big.run_batch(coords[c0:c0 + 64].contiguous())
        scale = m.abs().mean().item()
        err = (m - m_big[c0:c0 + 64]).abs()
        assert err.max().item() < 0.05 * max(scale, 1.0) and err.mean().item() < 2e-3 * max(scale, 1.0), \
            (c0, err.max().item(), err.mean().item(), scale)
        assert (iou - iou_big[c0:c0 + 64]).abs().max().item() < 5e-3
        assert (cls - cls_big[c0:c0 + 64]).abs().max().item() < 5e-3


def test_decoder_production_batch_vs_reference_golden(cuda):
    """The kernels the benchmark runs (>= 256 prompts: csam_t2i_stream, csam_upscale_stream, csam_i2t_rank, csam_i2t_stream)
    against the REFERENCE's own mask decoder on 320 prompts (tests/golden/decoder_big_test128.npz, produced by
    oracle/make_goldens.py::golden_decoder_big from /root/reference/segment_anything_cs/modeling/mask_decoder.py)."""
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    g = np.load(os.path.join(G, "decoder_big_test128.npz"))
    sd = synth.make_sam_state_dict("vit_test128")
    plan = DecoderPlan(sd, cuda, n_class=1, max_batch=320)
    assert plan.t2i_stream and plan.up_stream and plan.i2t_stream and plan.i2t_rank
    _set_image(plan, cuda)
    pts = np.random.RandomState(12).randint(0, 1024, size=(320, 1, 2)).astype(np.float64)
    coords = torch.from_numpy(pts[:, 0, :].astype(np.float32)).to(cuda).contiguous()
    masks, iou, cls = plan.run_batch(coords)
    torch.cuda.synchronize()
    low = masks.cpu().numpy()
    ref = g["low_sample"]
    err = np.abs(low[:, :, 5::32, 9::32] - ref)
    scale = np.abs(ref).mean()
    print("B=320 low-res logits: mean|ref|=%.3f max err=%.4f mean err=%.5f" % (scale, err.max(), err.mean()))
    assert err.mean() < 0.005 * scale and err.max() < 0.05 * scale
    np.testing.assert_allclose(low.astype(np.float64).sum((2, 3)), g["low_sum"], rtol=0, atol=0.005 * scale * 65536)
    np.testing.assert_allclose(np.abs(low.astype(np.float64)).sum((2, 3)), g["low_abs_sum"], rtol=0, atol=0.005 * scale * 65536)
    np.testing.assert_allclose(low.max((2, 3)), g["low_max"], rtol=0, atol=0.05 * scale)
    e_iou = np.abs(iou.cpu().numpy() - g["iou"]).max()
    e_cls = np.abs(cls.cpu().numpy() - g["cls"]).max()
    print("B=320 iou err %.5f  cls err %.5f" % (e_iou, e_cls))
    assert e_iou < 5e-3 and e_cls < 5e-3
    # PWD-Net selection (argmax of clamp(iou,0)*sigmoid(cls), first max wins) must agree wherever the reference's
    # margin between its best two candidates exceeds the value tolerance
    s_ref = np.clip(g["iou"], 0, None) / (1 + np.exp(-g["cls"][..., 0]))
    s_got = np.clip(iou.cpu().numpy(), 0, None) / (1 + np.exp(-cls.cpu().numpy()[..., 0]))
    top2 = np.sort(s_ref, 1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    clear = margin > 1e-2
    flips = s_ref.argmax(1) != s_got.argmax(1)
    # say how many prompts sit INSIDE the margin and how many of those actually flip.  A flip can only
    # happen where the reference's own margin is smaller than twice the score error; the score error is bounded above.
    e_s = np.abs(s_got - s_ref).max()
    print("B=320 PWD-Net selection: %d of 320 prompts inside the 1e-2 margin, %d inside 2 x the score error (%.2e), %d flips "
          "(all inside: %s)" % ((~clear).sum(), (margin < 2 * e_s).sum(), e_s, flips.sum(), bool(np.all(margin[flips] < 2 * e_s))))
    assert clear.sum() > 100 and not flips[clear].any()
    assert e_s < 5e-3 and np.all(margin[flips] < 2 * e_s)          # every flip is explained by the score tolerance
    assert flips.sum() <= max(3, int(0.02 * 320))                   # and they are rare: <= 2 % of the prompts


def test_decoder_i2t_t2i_fusion_is_bitwise_the_unfused_sweep(cuda):
    """csam_i2t_t2i (image->token pass + the next block's token->image attention in one kernel; the next layer's token
    self-attention is issued ahead of it) must not move a single bit of the decoder's outputs: same arithmetic on the
    same fp16 key rows, only their route (LDS instead of HBM) and the launch order of independent token ops differ."""
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    sd = synth.make_sam_state_dict("vit_test128")
    plan = DecoderPlan(sd, cuda, n_class=1, max_batch=320)
    assert plan.i2t_t2i and plan.t2i_rank and plan.i2t_rank and plan.i2t_rank_l1
    plan.i2t_fold = False        # the folded constants of round 4 change the arithmetic (next test); this one pins the ROUTE
    _set_image(plan, cuda)
    pts = np.random.RandomState(5).randint(0, 1024, size=(320, 2)).astype(np.float32)
    coords = torch.from_numpy(pts).to(cuda).contiguous()
    outs = {}
    for on in (True, False, True):
        plan.i2t_t2i = on
        plan.batch_graphs.clear()
        m, iou, cls = plan.run_batch(coords)
        torch.cuda.synchronize()
        cur = (m.clone(), iou.clone(), cls.clone())
        if on in outs:
            assert all(torch.equal(a, b) for a, b in zip(outs[on], cur))       # repeatable
        outs[on] = cur
    for a, b, name in zip(outs[True], outs[False], ("masks", "iou", "cls")):
        assert torch.equal(a, b), (name, (a - b).abs().max().item())


@pytest.mark.parametrize("B", [32, 7, 1])
def test_token_block_kernels_match_the_launch_sequence_they_replace(cuda, B):
    """csam_token_block_a / _b (round 4): the token side of a decoder block for small batches in two launches.  Same operands,
    same rounding points, fp32 accumulation in the same K order as the GEMM / LayerNorm / attention launches they replace (the
    split-K form of two of those GEMMs is switched off for the comparison): IoU scores bit-identical, hyper-network weights
    equal to the last fp32 bit (see below), for even, odd and single-prompt batches."""
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    sd = synth.make_sam_state_dict("vit_test128")
    plan = DecoderPlan(sd, cuda, n_class=1, max_batch=64)
    assert plan.token_block
    _set_image(plan, cuda)
    plan.splitk = False
    pts = np.random.RandomState(21).randint(0, 1024, size=(B, 2)).astype(np.float32)
    coords = torch.from_numpy(pts).to(cuda).contiguous()
    outs = {}
    for on in (True, False, True):
        plan.token_block = on
        plan.batch_graphs.clear()
        m, iou, cls = plan.run_batch(coords)
        torch.cuda.synchronize()
        cur = (m.clone(), iou.clone(), cls.clone())
        if on in outs:
            assert all(torch.equal(a, b) for a, b in zip(outs[on], cur))       # repeatable
        outs[on] = cur
    scale = outs[False][0].abs().mean().item()
    d = [(a - b).abs().max().item() for a, b in zip(outs[True], outs[False])]
    print("B=%d token blocks vs separate launches: max |diff| masks %.3e (mean |logit| %.3f), iou %.3e, cls %.3e"
          % (B, d[0], scale, d[1], d[2]))
    # the two block kernels are bit-identical to their launch sequences; csam_token_heads' fp32 hyper-network output layer
    # differs from csam_linear_f32_batched in the LAST BIT for rows 6, 7 of every 8 (that kernel's compiler-scheduled row
    # pairs do not all contract to fma; tools/debug/heads_diff.py), which shows as ~2e-6 in the low-res logits
    assert d[0] < 1e-5 and d[1] == 0.0 and d[2] < 1e-5


def test_decoder_folded_constants_stay_inside_the_fp16_noise(cuda):
    """csam_i2t_t2i_fold (round 4): out-projection bias carried by M_b, layer 1's norm4 gamma / beta folded into the first
    conv of the upscaler and the final attention.  Exact in real arithmetic; in fp16 it moves roundings.  The decoder with
    the fold must agree with the decoder without it as closely as either agrees with the reference (production-batch
    golden: mean 1.0e-3 of a mean |logit| of 1.45), and be bit-repeatable."""
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    sd = synth.make_sam_state_dict("vit_test128")
    plan = DecoderPlan(sd, cuda, n_class=1, max_batch=320)
    assert plan.i2t_fold and plan.i2t_t2i
    _set_image(plan, cuda)
    pts = np.random.RandomState(5).randint(0, 1024, size=(320, 2)).astype(np.float32)
    coords = torch.from_numpy(pts).to(cuda).contiguous()
    outs = {}
    for on in (True, False, True):
        plan.i2t_fold = on
        plan.batch_graphs.clear()
        m, iou, cls = plan.run_batch(coords)
        torch.cuda.synchronize()
        cur = (m.clone(), iou.clone(), cls.clone())
        if on in outs:
            assert all(torch.equal(a, b) for a, b in zip(outs[on], cur))       # repeatable
        outs[on] = cur
    scale = outs[False][0].abs().mean().item()
    dm = (outs[True][0] - outs[False][0]).abs()
    print("fold on vs off: low-res logits mean |diff| %.3e max %.3e (mean |logit| %.3f); iou max %.3e; cls max %.3e"
          % (dm.mean().item(), dm.max().item(), scale, (outs[True][1] - outs[False][1]).abs().max().item(),
             (outs[True][2] - outs[False][2]).abs().max().item()))
    assert dm.mean().item() < 1.5e-3 * scale and dm.max().item() < 3e-2 * scale
    assert (outs[True][1] - outs[False][1]).abs().max().item() < 5e-3
    assert (outs[True][2] - outs[False][2]).abs().max().item() < 1e-3


@pytest.mark.parametrize("B", [3, 300])
def test_upscale_stream_vs_oracle_fp32(cuda, B):
    """csam_upscale_stream -- the largest kernel of the benchmark -- directly against the oracle's fp32 restatement of
    mask_decoder.py:172-181 (erf GELU, LayerNorm2d, both transposed convolutions, hyper product), with the plan's own
    weight permutations; not against another HIP kernel."""
    from crowdsam_amd import hip, synth
    from crowdsam_amd.decoder import DecoderPlan
    from oracle import sam_oracle as so
    sd = synth.make_sam_state_dict("vit_test128")
    plan = DecoderPlan(sd, cuda, n_class=1, max_batch=8)
    gen = torch.Generator().manual_seed(40 + B)
    nX = min(B, 6)
    X = (torch.randn(nX, 4096, 256, generator=gen) * 0.9).half()
    idx = torch.arange(B) % nX
    hyper = torch.randn(B, 4, 32, generator=gen) * 0.6
    masks = torch.full((B, 4, 256, 256), float("nan"), device=cuda)
    stats = torch.full((B * 4, 2), float("nan"), device=cuda)
    Xd = X[idx].contiguous().view(B * 4096, 256).to(cuda)
    hip.upscale_stream(Xd, plan.up1_w, plan.up1_b, plan.up_ln_g, plan.up_ln_b, 1e-6, plan.up2_w_perm, plan.up2_b,
                       hyper.to(cuda).contiguous(), masks, B, stats=stats)
    torch.cuda.synchronize()
    with torch.no_grad():
        up_src = X.float().transpose(1, 2).reshape(nX, 256, 64, 64)
        for b in range(0, B, 50):
            sl = slice(b, min(b + 50, B))
            ref = so.upscale_hyper(sd, up_src[idx[sl]], hyper[sl])
            got = masks[sl].cpu()
            err = (got - ref).abs()
            scale = ref.abs().mean().item()
            assert err.mean().item() < 0.003 * scale and err.max().item() < 0.03 * scale, \
                (b, err.mean().item(), err.max().item(), scale)
            assert torch.allclose(stats[sl.start * 4:sl.stop * 4, 0].cpu(), got.reshape(-1, 65536).max(1).values)


def test_set_image_twice_with_fresh_tensors_is_not_stale(plan, cuda):
    """the per-image hipGraph must read the plan's own operand buffers.  Two images in FRESH tensors (new
    addresses, different contents): the second call's state must equal what a plan that never saw the first computes."""
    from crowdsam_amd import synth
    from crowdsam_amd.decoder import DecoderPlan
    rs = np.random.RandomState(5)
    imgs = []
    for _ in range(2):
        feat = torch.from_numpy(rs.standard_normal((4096, 256)).astype(np.float32)).to(cuda)
        dtok = torch.zeros(5376, 1024, dtype=torch.float16, device=cuda)
        dtok[:5329] = torch.from_numpy(rs.standard_normal((5329, 1024)).astype(np.float32)).to(cuda).half()
        imgs.append((feat, dtok))
    coords = torch.tensor([[100.0, 200.0], [900.5, 31.0], [512.0, 512.0]], device=cuda)
    plan.set_image(*imgs[0])
    first = [t.clone() for t in plan.run_batch(coords)] + [plan.fg_logits().clone()]
    plan.set_image(imgs[1][0].clone(), imgs[1][1].clone())                       # second image, fresh addresses
    second = [t.clone() for t in plan.run_batch(coords)] + [plan.fg_logits().clone()]
    fresh = DecoderPlan(synth.make_sam_state_dict("vit_test128"), cuda, n_class=1, max_batch=8)
    fresh.set_image(*imgs[1])
    want = list(fresh.run_batch(coords)) + [fresh.fg_logits()]
    for a, b, c in zip(second, want, first):
        assert torch.equal(a, b)
        assert not torch.equal(a, c)
