"""Full-size properties (BASELINE.json configs 2 and 4 prompt counts) through the whole HIP path.  The oracle cannot
finish these sizes in seconds, so the checks are size-independent invariants: determinism, score ordering, NMS
idempotence, RLE / box consistency, and agreement of the two batch sizes of the dense sweep."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ARCH = "vit_test128"        # narrow encoder; the decoder, PWD-Net and post-processing are full size


def _model(cuda, grid, ppb):
    from crowdsam.model import CrowdSAM
    from crowdsam.utils import DEFAULT_TEST_CONFIG
    from crowdsam_amd import synth
    t = dict(DEFAULT_TEST_CONFIG)
    t.update(grid_size=grid, points_per_batch=ppb, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"),
             max_prompts=grid * grid, stability_score_thresh=0.25, pred_iou_thresh=0.05)
    cfg = {"environ": {"device": "cuda"}, "model": {"sam_model": ARCH, "sam_arch": "crowdsam", "n_class": 1,
                                                    "trainfree": False}, "test": t}
    return CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict(ARCH), dino_state_dict=synth.make_dino_state_dict(depth=1),
                    dino_depth=1)


def _rle_area_and_box(rle):
    from segment_anything_cs.utils.amg import rle_to_mask
    m = rle_to_mask(rle)
    ys, xs = np.nonzero(m)
    return int(m.sum()), (xs.min(), ys.min(), xs.max(), ys.max()) if len(xs) else None


def _decode(counts_str, h, w):
    """COCO compressed RLE string -> uncompressed counts (pycocotools rleFrString)."""
    cnts, p, s = [], 0, counts_str
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return {"size": [h, w], "counts": cnts}


@pytest.mark.parametrize("grid", [64, 128])
def test_dense_sweep_full_size_invariants(cuda, grid):
    from crowdsam_amd import synth
    from crowdsam_amd import hip
    img = synth.synthetic_crowd_frame(7, 1024, 150)
    m = _model(cuda, grid, 2048)
    np.random.seed(0)
    a = m.generate(img)
    n_cand = m.last_candidates
    assert 0 < n_cand <= grid * grid
    np.random.seed(0)
    b = m.generate(img)
    for k in ("boxes", "scores", "points", "stability_score"):            # determinism (integer atomics only)
        assert np.array_equal(a[k], b[k]), k
    assert [r["counts"] for r in a["rles"]] == [r["counts"] for r in b["rles"]]
    boxes, scores = a["boxes"], a["scores"]
    assert len(boxes) == len(scores) == len(a["rles"]) and len(boxes) > 0
    assert (boxes[:, 0] >= 0).all() and (boxes[:, 2] <= 1024).all() and (boxes[:, 1] >= 0).all() and (boxes[:, 3] <= 1024).all()
    assert (scores > 0.05).all() and (a["stability_score"] >= 0.25).all()
    # NMS idempotence: the kept boxes survive a second pass unchanged
    keep = hip.box_nms(torch.from_numpy(boxes).float().to(cuda), torch.from_numpy(scores).float().to(cuda), m.box_nms_thresh)
    assert len(keep) == len(boxes)
    # RLE <-> box consistency on a sample of masks (boxes are inclusive max indices of the final mask)
    for i in range(0, len(boxes), max(1, len(boxes) // 8)):
        rle = _decode(a["rles"][i]["counts"], *a["rles"][i]["size"])
        area, bb = _rle_area_and_box(rle)
        assert sum(rle["counts"]) == 1024 * 1024 and area > 0
        assert tuple(int(v) for v in boxes[i]) == tuple(int(v) for v in bb)


def test_dense_sweep_batch_size_independence(cuda):
    """The dense sweep has no pruning, so the prompt batch size must not change what is found."""
    from crowdsam_amd import synth
    img = synth.synthetic_crowd_frame(11, 1024, 150)
    outs = []
    for ppb in (512, 2048):
        m = _model(cuda, 64, ppb)
        np.random.seed(3)
        o = m.generate(img)
        order = np.lexsort((o["points"][:, 1], o["points"][:, 0]))
        outs.append({k: o[k][order] for k in ("boxes", "scores", "points")})
    assert np.array_equal(outs[0]["points"], outs[1]["points"])
    assert np.array_equal(outs[0]["boxes"], outs[1]["boxes"])
    np.testing.assert_allclose(outs[0]["scores"], outs[1]["scores"], rtol=0, atol=1e-6)


def test_config4_vit_h_1500_frame_grid128(cuda):
    """BASELINE configs[4] with all three stress axes TOGETHER: ViT-H geometry (1280 wide, 16 heads x 80 -> the generic
    attention route; reduced depth 4, one windowed/global pair twice) + a 1500x1500 frame (device cv2-style down-scale to
    1024, boxes / points un-cropped by /downscale) + a 128x128 prompt grid (16 384 prompts, dense sweep).  The oracle
    cannot finish this size in seconds: size-independent invariants."""
    from crowdsam.model import CrowdSAM
    from crowdsam.utils import DEFAULT_TEST_CONFIG
    from crowdsam_amd import hip, synth
    from segment_anything_cs.build_sam import register_sam_arch
    register_sam_arch("vit_h_depth4", 1280, 4, 16, (1, 3))
    synth.SAM_CONFIGS["vit_h_depth4"] = (1280, 4, 16, (1, 3))
    t = dict(DEFAULT_TEST_CONFIG)
    t.update(grid_size=128, points_per_batch=2048, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"),
             max_prompts=128 * 128, stability_score_thresh=0.25, pred_iou_thresh=0.05)
    cfg = {"environ": {"device": "cuda"}, "model": {"sam_model": "vit_h_depth4", "sam_arch": "crowdsam", "n_class": 1,
                                                    "trainfree": False}, "test": t}
    m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_h_depth4"),
                 dino_state_dict=synth.make_dino_state_dict(depth=1), dino_depth=1)
    img = synth.synthetic_crowd_frame(4, 1500, 400)
    np.random.seed(0)
    a = m.generate(img)
    assert abs(m.downscale - 1024 / 1500) < 1e-12 and m.image_hw == (1024, 1024)
    n_cand = m.last_candidates
    assert 0 < n_cand <= 128 * 128
    np.random.seed(0)
    b = m.generate(img)
    for k in ("boxes", "scores", "points", "stability_score"):
        assert np.array_equal(a[k], b[k]), k
    boxes, scores = a["boxes"], a["scores"]
    assert len(boxes) == len(scores) == len(a["rles"]) > 0
    assert (boxes >= 0).all() and (boxes <= 1500).all() and (a["points"] >= 0).all() and (a["points"] < 1500).all()
    assert (scores > 0.05).all() and (a["stability_score"] >= 0.25).all() and np.all(np.diff(scores) <= 0)
    # the points are the 128x128 grid cells (1024-frame pixels / downscale): all distinct
    assert len({(float(x), float(y)) for x, y in a["points"]}) == len(a["points"])
    keep = hip.box_nms(torch.from_numpy(boxes).float().to(cuda), torch.from_numpy(scores).float().to(cuda), m.box_nms_thresh)
    assert len(keep) == len(boxes)
    for i in range(0, len(boxes), max(1, len(boxes) // 4)):      # RLE (1024 frame) <-> box (1500 frame) consistency
        rle = _decode(a["rles"][i]["counts"], *a["rles"][i]["size"])
        area, bb = _rle_area_and_box(rle)
        assert a["rles"][i]["size"] == [1024, 1024] and sum(rle["counts"]) == 1024 * 1024 and area > 0
        np.testing.assert_allclose(boxes[i], np.array(bb, np.float32) / np.float32(m.downscale), rtol=1e-6)
