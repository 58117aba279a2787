"""Pin the oracle (oracle/*.py CPU restatement) against golden vectors captured from the reference
itself (oracle/make_goldens.py, authoring container).  CPU-only; runs under -m "not gpu"."""
import os

import numpy as np
import pytest
import torch

from crowdsam_amd import synth
from oracle import pipeline_oracle as po
from oracle import sam_oracle as so

G = os.path.join(os.path.dirname(__file__), "golden")
ARCH = "vit_test128"


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=True)


@pytest.fixture(scope="module")
def sd():
    return synth.make_sam_state_dict(ARCH)


def _decoder_inputs():
    rs = np.random.RandomState(11)
    emb = torch.from_numpy(rs.standard_normal((1, 256, 64, 64)).astype(np.float32))
    dino = torch.from_numpy(rs.standard_normal((1, 73, 73, 1024)).astype(np.float32))
    pts = rs.randint(0, 1024, size=(5, 1, 2)).astype(np.float64)
    return emb, dino, pts


def test_amg_utils_match_reference():
    g = _load("amg.npz")
    rs = np.random.RandomState(5)
    logits = torch.from_numpy((rs.standard_normal((6, 40, 56)) * 2).astype(np.float32))
    logits = torch.nn.functional.avg_pool2d(logits[None], 5, 1, 2)[0] * 3
    logits[4] = -5.0
    logits[5] = 5.0
    stab = po.calculate_stability_score(logits, 0.0, 1.0).numpy()
    np.testing.assert_array_equal(np.nan_to_num(stab, nan=-1), np.nan_to_num(g["stab"], nan=-1))
    masks = logits > 0
    np.testing.assert_array_equal(po.batched_mask_to_box(masks).numpy(), g["boxes"])
    rles = po.mask_to_rle(masks)
    for r, c in zip(rles, g["rle_counts"]):
        assert r["counts"] == list(c)
        assert r["size"] == [40, 56]
    cb, cl = po.generate_crop_boxes((445, 640), 2, 0.341)
    np.testing.assert_array_equal(np.array(cb), g["crop_boxes"])
    np.testing.assert_array_equal(np.array(cl), g["crop_layers"])


def test_prompt_encoder_and_decoder_match_reference(sd):
    g = _load("decoder_test128.npz")
    emb, dino, pts = _decoder_inputs()
    with torch.no_grad():
        sparse = so.embed_points(sd, torch.as_tensor(pts), torch.ones(5, 1, dtype=torch.int))
        pe = so.dense_pe(sd)
        low, iou, cls = so.mask_decoder(sd, emb, pe, sparse, dino)
        m1 = so.postprocess_masks(low, (1024, 768), (1024, 768))
        m2 = so.postprocess_masks(low, (683, 1024), (682, 1023))
        fg = so.predict_fg_map(sd, dino)
    np.testing.assert_allclose(sparse.numpy(), g["sparse"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pe[:, ::8, ::4, ::4].numpy(), g["dense_pe_sample"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(low[:, :, ::8, ::8].numpy(), g["low_sample"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(low.double().sum((2, 3)).numpy(), g["low_sum"], rtol=1e-4, atol=0.5)
    np.testing.assert_allclose(iou.numpy(), g["iou"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cls.numpy(), g["cls"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m1[:, :, ::32, ::32].numpy(), g["post1_sample"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(m2[:, :, ::31, ::31].numpy(), g["post2_sample"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(fg[:, :, ::8, ::8].numpy(), g["fg_sample"], rtol=1e-4, atol=1e-5)


def test_box_prompts_match_reference(sd):
    """oracle box prompts (prompt_encoder.py:95-102) + decoder against the reference's own run on six seeded boxes."""
    from oracle.make_goldens import decoder_box_inputs
    g = _load("decoder_box_test128.npz")
    emb, dino, _ = _decoder_inputs()
    with torch.no_grad():
        sparse = so.embed_boxes(sd, torch.as_tensor(decoder_box_inputs()))
        low, iou, cls = so.mask_decoder(sd, emb, so.dense_pe(sd), sparse, dino)
    np.testing.assert_allclose(sparse.numpy(), g["sparse"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(low[:, :, ::8, ::8].numpy(), g["low_sample"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(iou.numpy(), g["iou"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cls.numpy(), g["cls"], rtol=1e-4, atol=1e-5)


def test_blob_weight_decoder_matches_reference():
    """Round 6: the blob-mask weight set (synth.blob_heads -- what bench.py runs) through the oracle's prompt encoder + mask decoder
    against the REFERENCE's own run on the same weights (tests/golden/decoder_blob_test128.npz): logits, the pixel counts behind
    area / stability, IoU and class scores; and the property the weight set exists for -- every mask is a compact region around
    its prompt with a high stability score."""
    from oracle.make_goldens import decoder_blob_inputs
    g = _load("decoder_blob_test128.npz")
    bsd, emb, dino, pts = decoder_blob_inputs()
    with torch.no_grad():
        sparse = so.embed_points(bsd, torch.as_tensor(pts), torch.ones(len(pts), 1, dtype=torch.int))
        low, iou, cls = so.mask_decoder(bsd, emb, so.dense_pe(bsd), sparse, dino)
    scale = np.abs(g["low_sample"]).mean()
    np.testing.assert_allclose(low[:, :, 1::8, 3::8].numpy(), g["low_sample"], rtol=1e-4, atol=2e-4 * scale + 1e-4)
    np.testing.assert_allclose(low.double().abs().sum((2, 3)).numpy(), g["low_abs_sum"], rtol=1e-4)
    np.testing.assert_allclose(iou.numpy(), g["iou"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cls.numpy(), g["cls"], rtol=1e-4, atol=1e-5)
    for name, thr in (("area", 0.0), ("inter", 1.0), ("union", -1.0)):
        cnt = (low > thr).sum((2, 3)).numpy()
        assert np.abs(cnt - g[name]).max() <= 2, name                  # a pixel within fp32 round-off of a threshold may flip
    # the reference's own masks are blobs at the prompt: largest candidate 100 .. 12 000 of 65 536 low-res pixels, stability >= 0.8
    # for most, centred within a few pixels of the prompt
    area, stab = g["area"][:, 3], g["inter"][:, 3] / np.maximum(g["union"][:, 3], 1)
    assert (area > 100).all() and np.median(area) < 2000 and np.median(stab) > 0.85
    yy, xx = np.mgrid[0:256, 0:256]
    m3 = (low[:, 3] > 0).numpy()
    cx = (m3 * xx).sum((1, 2)) / np.maximum(m3.sum((1, 2)), 1) * 4
    cy = (m3 * yy).sum((1, 2)) / np.maximum(m3.sum((1, 2)), 1) * 4
    off = np.hypot(cx - pts[:, 0, 0], cy - pts[:, 0, 1])
    assert np.median(off) < 8.0, off


def test_encoder_matches_reference(sd):
    g = _load("encoder_test128.npz")
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, 3, 1024, 1024)).astype(np.float32))
    with torch.no_grad():
        y = so.image_encoder(sd, x, depth, heads, gidx)
    np.testing.assert_allclose(y[:, ::4, ::4, ::4].numpy(), g["sample"], rtol=1e-3, atol=2e-4)
    assert abs(float(y.double().abs().sum()) - float(g["abs_sum"])) < 1e-4 * float(g["abs_sum"])


def test_full_pipeline_matches_reference(sd):
    """Tier O2: reference CrowdSAM.generate (with our third-party stand-ins) vs OracleCrowdSAM."""
    from oracle.make_goldens import PIPE_CFG, StandInDino, pipeline_image
    g = _load("pipeline_test128.npz")
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    np.random.seed(42)
    o = po.OracleCrowdSAM(sd, (depth, heads, gidx), StandInDino(), dict(PIPE_CFG), rng=np.random)
    with torch.no_grad():
        out = o.generate(pipeline_image())
    assert out["boxes"].shape == g["boxes"].shape
    np.testing.assert_array_equal(out["boxes"], g["boxes"])
    np.testing.assert_allclose(out["scores"], g["scores"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(out["points"], g["points"])
    np.testing.assert_array_equal(out["categories"], g["categories"])
    # a pixel sitting within fp32 summation-order noise of the +-1 / 0 logit thresholds may flip
    # between two CPU implementations: allow a few pixels, not more.
    np.testing.assert_allclose(out["stability_score"], g["stability_score"], rtol=1e-3)
    for r, c in zip(out["rles"], g["rle_counts"]):
        a = _rle_decode(po_counts(r), 768 * 1024)
        b = _rle_decode(_coco_decode(str(c)), 768 * 1024)
        assert int((a != b).sum()) <= 4


def test_full_pipeline_fuse_simmap_matches_reference(sd):
    """Same with test.fuse_simmap = True: scores = sqrt(iou) * sqrt(clamp(mean prior over the mask + 0.5))."""
    from oracle.make_goldens import PIPE_CFG, StandInDino, pipeline_image
    g = _load("pipeline_test128_fuse.npz")
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    np.random.seed(42)
    cfg = dict(PIPE_CFG)
    cfg["fuse_simmap"] = True
    o = po.OracleCrowdSAM(sd, (depth, heads, gidx), StandInDino(), cfg, rng=np.random)
    with torch.no_grad():
        out = o.generate(pipeline_image())
    np.testing.assert_array_equal(out["boxes"], g["boxes"])
    np.testing.assert_array_equal(out["points"], g["points"])
    np.testing.assert_allclose(out["scores"], g["scores"], rtol=1e-5, atol=1e-6)


def test_shipped_scale_eps_chain_matches_reference():
    """The first 4 of the 16 pruned batches of the reference's shipped-scale run (tests/golden/pipeline_eps_shipped.npz;
    crowdsam/model.py:226-248, grid 192, 32 prompts per batch, filter_thresh 0.7): the oracle prompts the same points in
    the same order, i.e. its FG prior set, shuffle, selection, filters and occupancy pruning are the reference's."""
    from oracle.make_goldens import StandInDino, pipeline_image
    g = _load("pipeline_eps_shipped.npz")
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    sd2 = synth.shipped_scale_heads(synth.make_sam_state_dict(ARCH))
    rec = {}
    np.random.seed(42)
    o = po.OracleCrowdSAM(sd2, (depth, heads, gidx), StandInDino(), dict(max_prompts=128), rng=np.random, record=rec)
    with torch.no_grad():
        o.generate(pipeline_image())
    assert len(rec["batches"]) == 4
    for i, b in enumerate(rec["batches"]):
        np.testing.assert_array_equal(b["points"], g["batch_points"][i])
        np.testing.assert_array_equal(b["sel"].numpy(), g["sel"][i])
        sc = b["iou_fused"][torch.arange(32), b["sel"]].numpy()
        np.testing.assert_allclose(sc, g["score"][i], rtol=1e-5, atol=1e-6)


def po_counts(r):
    return _coco_decode(r["counts"])


def _rle_decode(counts, n):
    out = np.zeros(n, dtype=bool)
    idx, par = 0, False
    for c in counts:
        out[idx:idx + c] = par
        idx += c
        par = not par
    return out


def _coco_decode(s):
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def test_coco_rle_string_roundtrip():
    """pycocotools' rleFrString (published inverse) recovers the counts."""
    decode = _coco_decode
    rs = np.random.RandomState(0)
    for _ in range(20):
        counts = rs.randint(0, 70000, size=rs.randint(1, 40)).tolist()
        assert decode(po.coco_rle_string(counts)) == counts


def test_nms_semantics():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], dtype=torch.float32)
    scores = torch.tensor([0.9, 0.8, 0.7, 0.9])
    keep = po.nms(boxes, scores, 0.5)
    assert keep.tolist() == [0, 2]          # stable: index 0 before 3 at equal score
    assert po.nms(boxes, scores, 1.0).tolist() == [0, 3, 1, 2]
