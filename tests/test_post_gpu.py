"""GPU parity (bit-exact integer outputs) of selection, fused mask post-processing, NMS and RLE
against the oracle restatement of amg.py / torchvision semantics on the same seeded inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _smooth_logits(B, seed, scale=4.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, 256, 256, generator=g)
    x = torch.nn.functional.avg_pool2d(x, 15, 1, 7) * 15 * scale / 4
    return x.contiguous()


def test_select_masks(cuda):
    from crowdsam_amd import hip
    g = torch.Generator().manual_seed(0)
    B = 300
    iou = torch.randn(B, 4, generator=g)
    cls = torch.randn(B, 4, 1, generator=g)
    iou[5] = torch.tensor([0.3, 0.3, 0.3, 0.3])
    cls[5] = 0.5          # exact tie -> first index wins (torch.max semantics)
    sel = torch.empty(B, dtype=torch.int32, device=cuda)
    score = torch.empty(B, device=cuda)
    cat = torch.empty(B, dtype=torch.int32, device=cuda)
    fused = torch.empty(B, 4, device=cuda)
    hip.select_masks(iou.to(cuda), cls.to(cuda).contiguous(), 1, sel, score, cat, fused, B)
    ref = torch.clamp(iou, 0.) * cls.squeeze(2).sigmoid()
    np.testing.assert_allclose(fused.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
    f = fused.cpu()
    assert torch.equal(sel.cpu().long(), f.max(dim=-1)[1])
    assert sel[5].item() == 0


@pytest.mark.parametrize("in_hw,out_hw", [((1024, 768), (1024, 768)), ((683, 1024), (683, 1024)),
                                          ((683, 1024), (682, 1023))])
def test_mask_post_matches_oracle(cuda, in_hw, out_hw):
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po, sam_oracle as so
    B = 6
    low = _smooth_logits(B, 3)
    sel = torch.tensor([0, 1, 2, 3, 1, 0], dtype=torch.int32)
    low[4, 1] = -50.0           # empty mask
    H, W = out_hw
    mask = torch.zeros(B, H, W, dtype=torch.uint8, device=cuda)
    inter = torch.empty(B, dtype=torch.int32, device=cuda)
    uni = torch.empty(B, dtype=torch.int32, device=cuda)
    box = torch.empty(B, 4, dtype=torch.int32, device=cuda)
    tmp = torch.empty(B, in_hw[0], in_hw[1], device=cuda) if in_hw != out_hw else None
    hip.mask_post(low.to(cuda), sel.to(cuda), B, in_hw, out_hw, 0.0, 1.0, mask, inter, uni, box, tmp)
    stab = torch.empty(B, device=cuda)
    keep = torch.empty(B, dtype=torch.uint8, device=cuda)
    occ = torch.empty(B, dtype=torch.uint8, device=cuda)
    score = torch.linspace(0.05, 0.9, B).to(cuda)
    hip.post_finalize(score, inter, uni, box, 0.1, 0.5, 0.7, stab, keep, occ, B)
    # oracle: full (B,4,H,W) upsample, gather, thresholds
    full = so.postprocess_masks(low, in_hw, out_hw)
    selm = full[torch.arange(B), sel.long()]
    r_inter, r_uni = po.stability_counts(selm, 0.0, 1.0)
    r_mask = selm > 0.0
    r_box = po.batched_mask_to_box(r_mask)
    # pixels whose logit sits within fp32 interpolation round-off of a threshold may legitimately flip
    near = lambda t: ((selm - t).abs() < 2e-5).flatten(1).sum(-1)
    m = mask.cpu().bool()
    diff = (m != r_mask).flatten(1).sum(-1)
    npx = selm[0].numel()
    print("mask post %s -> %s: pixels within 2e-5 of a threshold per mask: 0: %s  +1: %s  -1: %s of %d; mask pixels that "
          "differ from the oracle: %s" % (in_hw, out_hw, near(0.0).tolist(), near(1.0).tolist(), near(-1.0).tolist(), npx,
                                          diff.tolist()))
    assert torch.all(diff <= near(0.0)), (diff, near(0.0))
    # the escape hatch is narrow: at most 0.02 % of a mask's pixels may sit that close to a threshold at all
    assert float(near(0.0).max()) <= 2e-4 * npx and float(near(1.0).max()) <= 2e-4 * npx and float(near(-1.0).max()) <= 2e-4 * npx
    assert torch.all((inter.cpu() - r_inter).abs() <= near(1.0))
    assert torch.all((uni.cpu() - r_uni).abs() <= near(-1.0))
    ok = near(0.0) == 0
    assert torch.equal(box.cpu().long()[ok], r_box[ok])
    assert box[4].tolist() == [0, 0, 0, 0]
    r_stab = r_inter / r_uni
    np.testing.assert_allclose(stab.cpu().numpy()[:4], r_stab.numpy()[:4], rtol=1e-3)
    exp_keep = (score.cpu() > 0.1) & (stab.cpu() >= 0.5)
    assert torch.equal(keep.cpu().bool(), exp_keep)
    assert torch.equal(occ.cpu().bool(), exp_keep & (score.cpu() > 0.7))


@pytest.mark.parametrize("in_hw,out_hw", [((1024, 1024), (1024, 1024)), ((683, 1024), (600, 900))])
def test_mask_post_scored_skips_only_what_the_iou_filter_drops(cuda, in_hw, out_hw):
    """csam_mask_post_scored == csam_mask_post for prompts with score > thr; the others keep the initial statistics and
    are rejected by the finalize step exactly as before (same keep / occ flags)."""
    from crowdsam_amd import hip
    B = 6
    low = _smooth_logits(B, 5).to(cuda)
    sel = torch.tensor([0, 1, 2, 3, 1, 0], dtype=torch.int32, device=cuda)
    score = torch.tensor([0.05, 0.3, 0.1, 0.8, 0.0999, 0.11], device=cuda)
    i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=cuda)
    tmp = torch.empty(B, in_hw[0], in_hw[1], device=cuda) if in_hw != out_hw else None
    ia, ua, ba = i32(B), i32(B), i32(B, 4)
    ib, ub, bb = i32(B), i32(B), i32(B, 4)
    hip.mask_post(low, sel, B, in_hw, out_hw, 0.0, 1.0, None, ia, ua, ba, tmp)
    hip.mask_post_scored(low, sel, score, 0.1, B, in_hw, out_hw, 0.0, 1.0, ib, ub, bb, tmp)
    act = (score > 0.1).cpu()
    assert torch.equal(ia.cpu()[act], ib.cpu()[act]) and torch.equal(ua.cpu()[act], ub.cpu()[act])
    assert torch.equal(ba.cpu()[act], bb.cpu()[act])
    assert (ib.cpu()[~act] == 0).all() and (ub.cpu()[~act] == 0).all()
    outs = []
    for (i, u, b) in ((ia, ua, ba), (ib, ub, bb)):
        stab = torch.empty(B, device=cuda)
        keep = torch.empty(B, dtype=torch.uint8, device=cuda)
        occ = torch.empty(B, dtype=torch.uint8, device=cuda)
        hip.post_finalize(score, i, u, b, 0.1, 0.5, 0.7, stab, keep, occ, B)
        outs.append((keep.cpu(), occ.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # threshold <= 0 disables the skip (the reference only filters when pred_iou_thresh > 0)
    hip.mask_post_scored(low, sel, score, 0.0, B, in_hw, out_hw, 0.0, 1.0, ib, ub, bb, tmp)
    assert torch.equal(ia.cpu(), ib.cpu()) and torch.equal(ua.cpu(), ub.cpu())


def test_occupancy_lookup(cuda):
    from crowdsam_amd import hip
    g = torch.Generator().manual_seed(2)
    B, H, W, P = 7, 300, 400, 1000
    masks = (torch.rand(B, H, W, generator=g) > 0.8).to(torch.uint8)
    occ = torch.tensor([1, 0, 1, 1, 0, 0, 1], dtype=torch.uint8)
    pts = torch.stack([torch.randint(0, W, (P,), generator=g), torch.randint(0, H, (P,), generator=g)], 1).int()
    out = torch.empty(P, dtype=torch.uint8, device=cuda)
    hip.occupancy_lookup(pts.to(cuda).contiguous(), masks.to(cuda), occ.to(cuda), B, H, W, out)
    ref = masks[occ.bool()].any(0)[pts[:, 1].long(), pts[:, 0].long()]
    assert torch.equal(out.cpu().bool(), ref)


@pytest.mark.parametrize("N", [1, 7, 64, 65, 500, 4096, 5000])
def test_box_nms_matches_oracle(cuda, N):
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    rs = np.random.RandomState(N)
    xy = rs.uniform(0, 900, size=(N, 2)).astype(np.float32)
    wh = rs.uniform(5, 200, size=(N, 2)).astype(np.float32)
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], 1))
    scores = torch.from_numpy(rs.uniform(0, 1, size=N).astype(np.float32))
    if N > 10:
        scores[3] = scores[9]                 # tie: stable order keeps index 3 first
        boxes[10] = boxes[2]                  # duplicate box
    for thr in (0.3, 0.65):
        keep = hip.box_nms(boxes.to(cuda), scores.to(cuda), thr).cpu()
        ref = po.nms(boxes, scores, thr)
        assert torch.equal(keep, ref), (N, thr, keep[:10], ref[:10])


def test_rle_matches_oracle(cuda):
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    low = _smooth_logits(2, 9)
    masks = (torch.nn.functional.interpolate(low, (333, 517), mode="bilinear")[:, :3] > 0).flatten(0, 1)
    masks[1] = False
    masks[2] = True
    m8 = masks.to(torch.uint8).to(cuda).contiguous()
    pos, offs = hip.rle_encode(m8)
    pos = pos.cpu().numpy().astype(np.int64)
    ref = po.mask_to_rle(masks)
    h, w = 333, 517
    for i in range(masks.shape[0]):
        p = pos[offs[i]:offs[i + 1]]
        idx = np.concatenate([[0], p, [h * w]])
        counts = [] if not bool(masks[i, 0, 0]) else [0]
        counts.extend(np.diff(idx).tolist())
        assert counts == ref[i]["counts"]


def test_rle_with_boxes_equals_full_scan(cuda):
    """csam_rle_count_box / csam_rle_write_box (round 4): with the masks' bounding boxes the passes read the boxes only; the
    change positions must be those of the full scan -- boxes touching every frame edge, a one-pixel mask at the origin, an
    empty mask (box 0,0,0,0), a full mask, blobs."""
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    H, W = 300, 512
    rs = np.random.RandomState(4)
    masks = np.zeros((12, H, W), bool)
    for i, (y0, y1, x0, x1) in enumerate([(0, 40, 0, 30), (260, 299, 480, 511), (0, 299, 200, 203), (100, 100, 0, 511),
                                           (0, 0, 0, 0), (299, 299, 511, 511), (50, 250, 3, 8), (1, 298, 1, 510)]):
        sub = rs.rand(y1 - y0 + 1, x1 - x0 + 1) > 0.4
        sub[0, :] |= True; sub[-1, :] |= True; sub[:, 0] |= True; sub[:, -1] |= True      # the box is tight
        masks[i, y0:y1 + 1, x0:x1 + 1] = sub
    masks[9] = True                                                                     # 8: empty, 9: full
    low = _smooth_logits(1, 21)
    masks[10:12] = (torch.nn.functional.interpolate(low, (H, W), mode="bilinear")[0, :2] > 0.5).numpy()
    m8 = torch.as_tensor(masks).to(torch.uint8).to(cuda).contiguous()
    boxes = po.batched_mask_to_box(torch.as_tensor(masks)).to(torch.int32).to(cuda).contiguous()
    pos_a, offs_a = hip.rle_encode(m8)
    pos_b, offs_b = hip.rle_encode(m8, boxes=boxes)
    assert torch.equal(offs_a, offs_b) and int(offs_a[-1]) > 1000
    assert torch.equal(pos_a[: int(offs_a[-1])], pos_b[: int(offs_b[-1])])
    idx = torch.as_tensor(np.array([11, 0, 5, 8, 2], np.int32)).to(cuda)               # through a slot list, too
    pos_c, offs_c = hip.rle_encode(m8, idx)
    pos_d, offs_d = hip.rle_encode(m8, idx, boxes=boxes[idx.long()].contiguous())
    assert torch.equal(offs_c, offs_d) and torch.equal(pos_c[: int(offs_c[-1])], pos_d[: int(offs_d[-1])])


@pytest.mark.parametrize("H,W", [(300, 512), (333, 517), (1024, 1024)])
def test_device_coco_strings_equal_the_host_packer(cuda, H, W):
    """csam_coco_rle_pack (round 5): COCO compressed-RLE strings packed on the device from the change positions must equal,
    character for character, the host chain they replace -- run lengths from the positions (mask_to_rle_arrays, pinned to the
    oracle's mask_to_rle by test_rle_matches_oracle) through the C string packer (pinned to hand-computed pycocotools strings
    by test_host_logic_cpu).  Noise (1e5+ runs per mask: the frame that stalled rounds 3-4's host path), blobs, an empty and a
    full mask, masks whose pixel (0, 0) is set (leading zero-length run), a single pixel at either end, slot lists, boxes."""
    from oracle import pipeline_oracle as po
    from segment_anything_cs.utils.amg import coco_encode_rles, mask_to_coco_rles, mask_to_rle_arrays
    rs = np.random.RandomState(H + W)
    masks = np.zeros((11, H, W), bool)
    masks[0] = rs.rand(H, W) > 0.5                                            # salt and pepper
    masks[1] = rs.rand(H, W) > 0.97
    low = _smooth_logits(1, 33)
    masks[2:5] = (torch.nn.functional.interpolate(low, (H, W), mode="bilinear")[0, :3] > 0.3).numpy()
    masks[3, 0, 0] = True
    masks[5] = True                                                           # 4: blob, 5: full, 6: empty
    masks[7, 0, 0] = True                                                     # one pixel at the very start ...
    masks[8, H - 1, W - 1] = True                                             # ... and at the very end
    masks[9, H // 3: 2 * H // 3, W // 4: W // 2] = True                       # a rectangle: long equal runs (the k - 2 differencing)
    masks[10] = ~masks[0]
    m8 = torch.as_tensor(masks).to(torch.uint8).to(cuda).contiguous()
    ref = [r["counts"] for r in coco_encode_rles(mask_to_rle_arrays(m8))]
    got = mask_to_coco_rles(m8)
    assert [r["size"] for r in got] == [[H, W]] * 11
    assert [r["counts"] for r in got] == ref
    assert sum(len(c) for c in ref) > 50000
    boxes = po.batched_mask_to_box(torch.as_tensor(masks)).to(cuda)
    assert [r["counts"] for r in mask_to_coco_rles(m8, boxes=boxes)] == ref
    idx = torch.as_tensor(np.array([10, 3, 6, 5, 0, 7], np.int32)).to(cuda)
    sub = mask_to_coco_rles(m8, idx=idx, boxes=boxes[idx.long()])
    assert [r["counts"] for r in sub] == [ref[i] for i in idx.tolist()]
    # already-packed entries pass through coco_encode_rles unchanged (crowdsam.model hands them over like that)
    assert coco_encode_rles(got) == got
    # decodes to the mask (oracle's own decoder of the COCO string)
    for i in (2, 3, 7, 8):
        assert np.array_equal(po.coco_rle_decode(got[i]["counts"], H, W), masks[i])


@pytest.mark.parametrize("w", [516, 517])
def test_rle_of_non_boolean_bytes_is_the_rle_of_their_truth_value(cuda, w):
    """the 4-column kernels (W % 4 == 0) compare bit 0 of packed bytes, the scalar ones raw bytes; both are only
    right for 0 / 1 bytes, so mask_to_rle_arrays normalises anything that is not bool (amg.py:107-135 encodes a bool
    tensor).  uint8 masks with values {0, 2, 3, 255} must encode like (mask != 0) on both kernel routes."""
    from segment_anything_cs.utils.amg import mask_to_rle_arrays
    rs = np.random.RandomState(3)
    raw = rs.choice(np.array([0, 0, 2, 3, 255], np.uint8), size=(3, 64, w))
    raw = np.repeat(np.repeat(raw[:, ::8, ::6], 8, 1), 6, 2)[:, :64, :w].copy()          # runs, not salt and pepper
    a = mask_to_rle_arrays(torch.from_numpy(raw).to(cuda))
    b = mask_to_rle_arrays(torch.from_numpy(raw != 0).to(cuda))
    assert len(a) == len(b) == 3
    for ra, rb in zip(a, b):
        assert ra["size"] == rb["size"] and np.array_equal(np.asarray(ra["counts"]), np.asarray(rb["counts"]))


@pytest.mark.parametrize("downscale", [1.0, 2.0, 3.0, 1.5])
def test_crop_edge_filter_at_the_20px_boundary(cuda, downscale):
    """The in-kernel crop-edge filter of csam_post_finalize_compact (crowdsam/utils.py:213-223 through
    uncrop_boxes_xyxy's tensor / Python-scalar division, which ATen evaluates as a multiply by the fp32 reciprocal) with box
    sides placed 19 / 20 / 21 px (after un-scaling) from every crop edge and from the frame border: keep flags must equal
    the oracle's `is_box_near_crop_edge` on the same boxes, also for downscales whose reciprocal is inexact."""
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    crop = [300, 200, 1300, 900]                       # x0, y0, x1, y1 in the frame
    orig = [0, 0, 1600, 900]                           # the crop's bottom edge IS the frame border (never "near a crop edge")
    rows = []
    for d in (19, 20, 21, 40):
        for side in range(4):
            # a box (crop-local, scaled by downscale) whose one side sits d frame pixels inside the corresponding crop edge
            fx0, fy0, fx1, fy1 = 500.0, 400.0, 900.0, 700.0               # frame coordinates of a harmless interior box
            f = [fx0, fy0, fx1, fy1]
            f[side] = crop[side] + d if side < 2 else crop[side] - d
            rows.append([round((f[0] - crop[0]) * downscale), round((f[1] - crop[1]) * downscale),
                         round((f[2] - crop[0]) * downscale), round((f[3] - crop[1]) * downscale)])
    box = torch.tensor(rows, dtype=torch.int32, device=cuda)
    B = box.shape[0]
    score = torch.full((B,), 0.9, device=cuda)
    inter = torch.full((B,), 1000, dtype=torch.int32, device=cuda)
    uni = torch.full((B,), 1000, dtype=torch.int32, device=cuda)
    category = torch.zeros(B, dtype=torch.int32, device=cuda)
    points = torch.zeros(B, 2, dtype=torch.int32, device=cuda)
    keep = torch.empty(B, dtype=torch.uint8, device=cuda)
    occ = torch.empty(B, dtype=torch.uint8, device=cuda)
    slot = torch.empty(B, dtype=torch.int32, device=cuda)
    cap = B + 4
    e = lambda *s, dt: torch.zeros(*s, dtype=dt, device=cuda)
    store = dict(score=e(cap, dt=torch.float32), stability=e(cap, dt=torch.float32), boxes=e(cap, 4, dt=torch.int32),
                 category=e(cap, dt=torch.int32), points=e(cap, 2, dt=torch.int32), counter=e(1, dt=torch.int32))
    hip.post_finalize_compact(score, inter, uni, box, category, points, 0.1, 0.5, 3.0e38, keep, occ, slot, store["counter"],
                              store, B, edge=(crop, orig, downscale, 20.0))
    near = po.is_box_near_crop_edge(box.cpu().float(), crop, orig, downscale, atol=20.0).numpy()
    got = keep.cpu().numpy().astype(bool)
    assert near.any() and (~near).any()
    assert np.array_equal(got, ~near), (downscale, np.nonzero(got != ~near)[0].tolist())
    assert int(store["counter"].item()) == int((~near).sum())
