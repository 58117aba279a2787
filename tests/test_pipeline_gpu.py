"""End-to-end GPU parity of CrowdSAM.generate (HIP path) against the reference's own end-to-end run
(tests/golden/pipeline_test128.npz, tier O2) and against the CPU oracle on the same inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
# Mask-level bounds: measured on MI355X (the tests print the measurement) x 2.5.  Random-weight masks
# are noise-like, so a large share of their pixels sits near the threshold; XOR is quoted relative to the mask area.
#   case                                  measured min IoU / max XOR share      bound (2.5 x the measured defect)
MASK_BOUNDS = {"golden": (0.9867, 0.0133),      # 0.994707 / 5.30e-3  (reference's own 3 x 8 EPS run, 11 masks)
               "dense": (0.9941, 0.0059),       # 0.997647 / 2.35e-3  (24 masks vs oracle)
               "multicrop": (0.9967, 0.0033),   # 0.998705 / 1.30e-3
               "max_size": (0.9953, 0.0047),    # 0.998123 / 1.88e-3
               "vit_b_sample": (0.958, None)}   # 0.98324 on every 8th pixel of every 8th row (noise masks, 64 of them)
# reference golden run (3 x 8 EPS): measured score error 3.6e-4, box difference 0 px, stability 2.4e-4 abs / 8.8e-3 rel -> x 2.5
# (a box side moves by whole pixels: one pixel is the smallest non-zero bound)
GOLD_SCORE_ATOL, BOX_PX, STAB_RTOL, STAB_ATOL = 1e-3, 1, 0.022, 6e-4
G = os.path.join(os.path.dirname(__file__), "golden")
ARCH = "vit_test128"


class GpuStandInDino:
    """Same arithmetic as oracle.make_goldens.StandInDino, on the GPU (test-side stand-in for the
    un-vendored DINOv2 when replaying the reference's golden run)."""

    def __init__(self, device, seed=1):
        rs = np.random.RandomState(seed)
        self.w = torch.from_numpy(rs.standard_normal((3, 1024)).astype(np.float32)).to(device)
        self.b = torch.from_numpy(rs.standard_normal((5329, 1024)).astype(np.float32)).to(device)

    def forward_features(self, x):
        p = torch.nn.functional.avg_pool2d(x, 14, 14)
        return {"x_norm_patchtokens": p.flatten(2).transpose(1, 2) @ self.w + self.b}


def mask_agreement(rles_hip, masks_ref, label=""):
    """Per-mask IoU and XOR pixel count between the HIP path's COCO RLE strings and reference / oracle masks (RLE dicts or
    arrays), aligned index by index.  Prints the worst of each and returns (min IoU, max XOR fraction of the mask area)."""
    import crowdsam.utils as cu
    ious, xors, fr = [], [], []
    for r, mref in zip(rles_hip, masks_ref):
        a = cu.coco_decode_rle(r).astype(bool)
        b = (cu.coco_decode_rle(mref) if isinstance(mref, dict) else np.asarray(mref)).astype(bool)
        assert a.shape == b.shape, (a.shape, b.shape)
        x = int((a ^ b).sum())
        u = int((a | b).sum())
        ious.append(1.0 if u == 0 else 1.0 - x / u)
        xors.append(x)
        fr.append(x / max(int(b.sum()), 1))
    if ious:
        print("%s mask agreement over %d masks: min IoU %.6f, max XOR %d px (%.2e of the mask area), mean XOR %.1f px"
              % (label, len(ious), min(ious), max(xors), max(fr), float(np.mean(xors))))
    return (min(ious), max(fr)) if ious else (1.0, 0.0)


def _config(test_cfg):
    from oracle.pipeline_oracle import DEFAULT_TEST_CFG
    t = dict(DEFAULT_TEST_CFG)
    t.update(test_cfg)
    return {"environ": {"device": "cuda"},
            "model": {"sam_model": ARCH, "sam_arch": "crowdsam", "n_class": 1, "trainfree": False},
            "test": t}


@pytest.fixture(scope="module")
def model(cuda):
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG
    sd = synth.make_sam_state_dict(ARCH)
    return CrowdSAM(_config(PIPE_CFG), sam_state_dict=sd, dino_model=GpuStandInDino(cuda))


def test_generate_matches_reference_golden(model):
    from oracle.make_goldens import pipeline_image
    g = np.load(os.path.join(G, "pipeline_test128.npz"), allow_pickle=True)
    np.random.seed(42)
    out = model.generate(pipeline_image())
    boxes, scores, points = out["boxes"], out["scores"], out["points"]
    print("hip:", boxes.shape, scores[:5], "ref:", g["boxes"].shape, g["scores"][:5])
    # indices: the surviving prompts (identified by their point) and their order must be identical
    assert boxes.shape == g["boxes"].shape
    np.testing.assert_array_equal(points, g["points"])
    np.testing.assert_array_equal(out["categories"], g["categories"])
    # values: fp16 operand tolerance on scores; boxes come from thresholded masks -> a few pixels
    np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=GOLD_SCORE_ATOL)
    sd_ = np.abs(out["stability_score"] - g["stability_score"])
    print("reference golden: max |score| error %.2e, max box difference %.1f px, stability error max abs %.2e / max rel %.2e"
          % (np.abs(scores - g["scores"]).max(), np.abs(boxes - g["boxes"]).max(), sd_.max(),
             (sd_ / np.maximum(g["stability_score"], 1e-9)).max()))
    assert np.abs(boxes - g["boxes"]).max() <= BOX_PX
    np.testing.assert_allclose(out["stability_score"], g["stability_score"], rtol=STAB_RTOL, atol=STAB_ATOL)
    assert len(out["rles"]) == len(g["rle_counts"])
    assert all(isinstance(r["counts"], str) and r["size"] == [768, 1024] for r in out["rles"])
    # the masks themselves: decode both RLE sets (the reference's strings are in the fixture), per-mask IoU and XOR count
    ref_rles = [{"size": [768, 1024], "counts": str(c)} for c in g["rle_counts"]]
    min_iou, max_xor = mask_agreement(out["rles"], ref_rles, "reference golden (vit_test128, 3 x 8 EPS)")
    assert min_iou >= MASK_BOUNDS["golden"][0] and max_xor <= MASK_BOUNDS["golden"][1]


def test_generate_dense_sweep_matches_oracle(model, cuda):
    """Dense-sweep mode (no pruning, no host sync inside the sweep) vs the CPU oracle."""
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle.make_goldens import StandInDino, pipeline_image
    cfg = dict(grid_size=6, pos_sim_thresh=-1.0, points_per_batch=16, max_prompts=64, pred_iou_thresh=0.0,
               stability_score_thresh=0.0, filter_thresh=float("inf"), min_mask_region_area=0,
               box_nms_thresh=1.0, crop_nms_thresh=1.0)
    old = {k: getattr(model, k) for k in cfg}
    for k, v in cfg.items():
        setattr(model, k, v)
    try:
        np.random.seed(7)
        out = model.generate(pipeline_image())
    finally:
        for k, v in old.items():
            setattr(model, k, v)
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    np.random.seed(7)
    o = po.OracleCrowdSAM(synth.make_sam_state_dict(ARCH), (depth, heads, gidx), StandInDino(), cfg, rng=np.random)
    with torch.no_grad():
        ref = o.generate(pipeline_image())
    assert out["boxes"].shape == ref["boxes"].shape == (24, 4)     # 6x6 grid cropped to 4x6 valid cells
    # with NMS disabled the output order is the score order; near-tied scores may swap under fp16
    # operands, so align the two results by prompt point before comparing values
    ka = np.lexsort((out["points"][:, 1], out["points"][:, 0]))
    kb = np.lexsort((ref["points"][:, 1], ref["points"][:, 0]))
    np.testing.assert_array_equal(out["points"][ka], ref["points"][kb])
    np.testing.assert_allclose(out["scores"][ka], ref["scores"][kb], rtol=0, atol=5e-3)
    assert np.abs(out["boxes"][ka] - ref["boxes"][kb]).max() <= 3
    assert np.all(np.diff(out["scores"]) <= 0)
    min_iou, max_xor = mask_agreement([out["rles"][i] for i in ka], [ref["rles"][i] for i in kb], "dense sweep vs oracle")
    assert min_iou >= MASK_BOUNDS["dense"][0] and max_xor <= MASK_BOUNDS["dense"][1]


def test_generate_blob_weights_matches_oracle(cuda):
    """The blob-mask weight set (synth.blob_heads, VERDICT r5 item 7b: masks are compact blobs around the prompt, stability 0.9+, the
    shipped thresholds are live) through the dense sweep vs the CPU oracle on the same weights: what the fp16 path does on masks
    with real edges instead of noise fields.  All 24 prompts are kept (filters off) so that every mask is compared."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle.make_goldens import PIPE_CFG, StandInDino, pipeline_image
    cfg = dict(grid_size=6, pos_sim_thresh=-1.0, points_per_batch=16, max_prompts=64, pred_iou_thresh=-10.0,
               stability_score_thresh=0.0, filter_thresh=float("inf"), min_mask_region_area=0,
               box_nms_thresh=1.0, crop_nms_thresh=1.0)
    pc = dict(PIPE_CFG)
    pc.update(cfg)
    m = CrowdSAM(_config(pc), sam_state_dict=synth.blob_heads(synth.make_sam_state_dict(ARCH)), dino_model=GpuStandInDino(cuda))
    np.random.seed(7)
    out = m.generate(pipeline_image())
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    np.random.seed(7)
    o = po.OracleCrowdSAM(synth.blob_heads(synth.make_sam_state_dict(ARCH)), (depth, heads, gidx), StandInDino(), cfg, rng=np.random)
    with torch.no_grad():
        ref = o.generate(pipeline_image())
    assert out["boxes"].shape == ref["boxes"].shape == (24, 4)
    ka = np.lexsort((out["points"][:, 1], out["points"][:, 0]))
    kb = np.lexsort((ref["points"][:, 1], ref["points"][:, 0]))
    np.testing.assert_array_equal(out["points"][ka], ref["points"][kb])
    ds = np.abs(out["scores"][ka] - ref["scores"][kb]).max()
    db = np.abs(out["boxes"][ka] - ref["boxes"][kb]).max()
    bw = ref["boxes"][kb][:, 2] - ref["boxes"][kb][:, 0]
    print("blob weights: %d masks, box widths %d..%d px (median %d), max score difference %.2e, max box difference %d px"
          % (len(ka), bw.min(), bw.max(), np.median(bw), ds, db))
    min_iou, max_xor = mask_agreement([out["rles"][i] for i in ka], [ref["rles"][i] for i in kb], "blob weights, dense sweep vs oracle")
    assert ds <= BLOB_BOUNDS[0] and db <= BLOB_BOUNDS[1] and min_iou >= BLOB_BOUNDS[2]


# measured on MI355X: score 1.3e-3, boxes 2 px, min mask IoU 0.9663 (blobs of 40-100 px: one ring of edge pixels is 3 % of such a mask)
#   -> x 2.5 on the defects
BLOB_BOUNDS = (3.3e-3, 5, 0.916)


def test_generate_fuse_simmap_matches_reference_golden(cuda):
    """test.fuse_simmap = True: prior-fused scores against the reference's own run."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG, pipeline_image
    cfg = dict(PIPE_CFG)
    cfg["fuse_simmap"] = True
    m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    g = np.load(os.path.join(G, "pipeline_test128_fuse.npz"), allow_pickle=True)
    np.random.seed(42)
    out = m.generate(pipeline_image())
    assert out["boxes"].shape == g["boxes"].shape
    np.testing.assert_array_equal(out["points"], g["points"])
    np.testing.assert_allclose(out["scores"], g["scores"], rtol=0, atol=5e-3)


def test_mask_mean_bilinear_matches_oracle(cuda):
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    rs = np.random.RandomState(0)
    H, W, fh, fw = 683, 1024, 43, 64
    sim_full = torch.from_numpy(rs.rand(64, 64).astype(np.float32))
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.stack([((yy - rs.uniform(0, H)) / rs.uniform(5, 200)) ** 2 + ((xx - rs.uniform(0, W)) / rs.uniform(5, 300)) ** 2 <= 1
                      for _ in range(9)] + [np.zeros((H, W), bool), np.ones((H, W), bool)])
    iou = torch.from_numpy(rs.rand(len(masks)).astype(np.float32))
    ref = po.fuse_simmap_scores(torch.from_numpy(masks), iou, sim_full[:fh, :fw], (H, W))
    mean = hip.mask_mean_bilinear(torch.from_numpy(masks).to(cuda), sim_full.to(cuda)[:fh, :fw])
    got = iou.to(cuda) ** 0.5 * torch.clamp(mean + 0.5, 0, 1) ** 0.5
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=1e-7)


def test_generate_vit_b_512_matches_reference_golden(cuda):
    """BASELINE configs[0]: ViT-B (_build_sam(768,12,12,1,[2,5,8,11])) + 8x8 grid on one 512x512 frame against the
    reference's own CrowdSAM.generate (tests/golden/pipeline_vit_b_512.npz).  Exercises the up-scaling frame resize
    (512 -> 1024, device cv2 restatement; the fixture pins the reference from the post-resize frame on), the ViT-B
    encoder geometry (12 heads x 64, global blocks 2/5/8/11) and the /downscale un-cropping of boxes and points."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import VITB_CFG, vitb_image
    cfg = _config(VITB_CFG)
    cfg["model"]["sam_model"] = "vit_b"
    m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_b"), dino_model=GpuStandInDino(cuda))
    g = np.load(os.path.join(G, "pipeline_vit_b_512.npz"), allow_pickle=True)
    np.random.seed(42)
    out = m.generate(vitb_image())
    assert m.downscale == 2.0 and out["boxes"].shape == g["boxes"].shape == (64, 4)
    # box NMS at threshold 1.0 suppresses nothing but returns the prompts in descending-score order; near-tied
    # scores may swap under fp16 operands, so the two results are aligned by prompt point (all 64 distinct)
    ka = np.lexsort((out["points"][:, 1], out["points"][:, 0]))
    kb = np.lexsort((g["points"][:, 1], g["points"][:, 0]))
    np.testing.assert_array_equal(out["points"][ka], g["points"][kb])
    np.testing.assert_array_equal(out["categories"][ka], g["categories"][kb])
    np.testing.assert_allclose(out["scores"][ka], g["scores"][kb], rtol=0, atol=5e-3)
    assert np.all(np.diff(out["scores"]) <= 0)
    assert np.abs(out["boxes"][ka] - g["boxes"][kb]).max() <= 1.5   # 3 px at the 1024 frame / downscale 2
    np.testing.assert_allclose(out["stability_score"][ka], g["stability_score"][kb], rtol=0.03, atol=2e-3)
    assert all(r["size"] == [1024, 1024] for r in out["rles"]) and len(out["rles"]) == 64
    import crowdsam.utils as cu
    area = np.array([int(cu.coco_decode_rle(r).sum()) for r in out["rles"]])
    np.testing.assert_allclose(area[ka], g["mask_area"][kb], rtol=5e-3)      # RLE strings decode to the reference's masks
    # pixel level: the fixture holds every 8th pixel of every 8th row of the reference's 64 masks
    mine = np.stack([cu.coco_decode_rle(out["rles"][i])[::8, ::8].astype(bool) for i in ka])
    theirs = np.unpackbits(g["mask_sample"])[: 64 * 128 * 128].reshape(64, 128, 128).astype(bool)[kb]
    xor = (mine ^ theirs).reshape(64, -1).sum(1)
    iou = 1.0 - xor / np.maximum((mine | theirs).reshape(64, -1).sum(1), 1)
    print("ViT-B 512: sampled mask agreement, min IoU %.5f, max XOR %d of %d sampled pixels" % (iou.min(), xor.max(), 128 * 128))
    assert iou.min() >= MASK_BOUNDS["vit_b_sample"][0]


@pytest.mark.parametrize("frame", ["768x1024", "700x1366"])
def test_generate_multi_crop_matches_oracle(cuda, frame):
    """crop_n_layers = 1: 1 + 4 crops; per-crop resize (cv2 restatement, incl. a 1023-side crop on the
    second frame -> the PIL 1023 -> 1024 step and the non-identity second interpolate of postprocess_masks), crop-edge
    filter applied per batch before the occupancy update, per-crop NMS, cross-crop NMS preferring small crops -- the
    kept prompts, their order and categories must equal the CPU oracle's, values within tolerance."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle.make_goldens import PIPE_CFG, StandInDino, pipeline_image
    cfg = dict(PIPE_CFG)
    cfg.update(crop_n_layers=1, crop_nms_thresh=0.7, box_nms_thresh=0.8, max_prompts=16, min_mask_region_area=0)
    img = pipeline_image() if frame == "768x1024" else synth.synthetic_crowd_frame(9, 1366, 120)[:700]
    m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    np.random.seed(1)
    out = m.generate(img)
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    np.random.seed(1)
    o = po.OracleCrowdSAM(synth.make_sam_state_dict(ARCH), (depth, heads, gidx), StandInDino(), cfg, rng=np.random)
    with torch.no_grad():
        ref = o.generate(img)
    print(frame, "kept", out["boxes"].shape, "oracle", ref["boxes"].shape)
    assert out["boxes"].shape == ref["boxes"].shape and len(ref["boxes"]) > 0
    np.testing.assert_array_equal(out["points"], ref["points"])              # same prompts survive, same order
    np.testing.assert_array_equal(out["categories"], ref["categories"])
    np.testing.assert_allclose(out["scores"], ref["scores"], rtol=0, atol=5e-3)
    assert np.abs(out["boxes"] - ref["boxes"]).max() <= 3.0 / min(1.0, 1024.0 / max(img.shape[:2]))
    np.testing.assert_array_equal(out["rles_crop"], ref["rles_crop"])        # per-mask crop box (build's sane rles_info)
    min_iou, max_xor = mask_agreement(out["rles"], ref["rles"], "multi-crop " + frame)
    assert min_iou >= MASK_BOUNDS["multicrop"][0] and max_xor <= MASK_BOUNDS["multicrop"][1]
    assert "crop_boxes" not in out and len(out["rles"]) == len(ref["rles"])


def test_generate_two_crop_layers_runs(cuda):
    """crop_n_layers = 1: 1 + 4 crops, cross-crop NMS preferring small crops, per-crop rles_info
    kept as a per-crop record (the reference index-filters that list and raises once > 2*n_crops masks survive)."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG, pipeline_image
    cfg = dict(PIPE_CFG)
    cfg.update(crop_n_layers=1, crop_nms_thresh=0.7, max_prompts=16)
    m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    np.random.seed(1)
    img = pipeline_image()
    out = m.generate(img)
    b = out["boxes"]
    assert b.ndim == 2 and b.shape[1] == 4 and len(out["scores"]) == len(b) == len(out["rles"])
    assert (b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 2] <= img.shape[1]).all() and (b[:, 3] <= img.shape[0]).all()
    assert "crop_boxes" not in out


def test_generate_without_detections_and_area_selection(cuda):
    """an image on which no prompt survives must still give a well-formed result (the reference's MaskData
    then holds boxes / scores / rles only; tools/test.py copies the keys present) -- here every per-mask field exists and
    is empty.  Also: mask_selection max_area / min_area (crowdsam/model.py:320-323) against the oracle's choice."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle import sam_oracle as so
    from oracle.make_goldens import PIPE_CFG, pipeline_image
    cfg = dict(PIPE_CFG)
    cfg.update(stability_score_thresh=0.9999, pred_iou_thresh=0.99)
    m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    np.random.seed(3)
    out = m.generate(pipeline_image())
    assert out["boxes"].shape == (0, 4) and len(out["scores"]) == 0 and out["rles"] == []
    assert len(out["categories"]) == 0 and out["points"].shape == (0, 2)
    rec = {k: v.tolist() for k, v in out.items() if k in ["boxes", "scores", "categories"]}      # tools/test.py:71
    assert rec == {"boxes": [], "scores": [], "categories": []}
    # area-based selection: same surviving prompts and the same candidate index as picking by thresholded area
    for mode in ("max_area", "min_area"):
        cfg2 = dict(PIPE_CFG)
        cfg2.update(mask_selection=mode, filter_thresh=float("inf"), max_prompts=16, box_nms_thresh=1.0, min_mask_region_area=0,
                    pred_iou_thresh=0.0, stability_score_thresh=0.0)
        m2 = CrowdSAM(_config(cfg2), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
        np.random.seed(5)
        o2 = m2.generate(pipeline_image())
        assert len(o2["boxes"]) == 16
        # oracle: decode the same 16 prompts, choose by area of (logit > 0) at full resolution
        D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
        np.random.seed(5)
        orc = po.OracleCrowdSAM(synth.make_sam_state_dict(ARCH), (depth, heads, gidx), __import__("oracle.make_goldens", fromlist=["x"]).StandInDino(),
                                dict(cfg2, mask_selection="max_iou"), rng=np.random)
        img = pipeline_image()
        orc.orig_image, orc.crop_box = img, [0, 0, img.shape[1], img.shape[0]]
        orc.image, orc.downscale = img, 1.0
        with torch.no_grad():
            orc.set_image(img)
            pts = o2["points"].astype(np.int64)
            masks, iou, cls, low = orc.predict_torch(torch.as_tensor(po.apply_coords(pts, orc.original_size))[:, None, :],
                                                     torch.ones(len(pts), 1, dtype=torch.int))
        area = (masks > 0).sum(dim=[-1, -2])
        ind = area.max(dim=-1)[1] if mode == "max_area" else area.min(dim=-1)[1]
        fused = torch.clamp(iou, 0.) * cls.squeeze(2).sigmoid()
        ref_score = fused[torch.arange(len(pts)), ind].numpy()
        srt = np.sort(area.numpy(), 1)
        clear = (srt[:, -1] - srt[:, -2] > 200) if mode == "max_area" else (srt[:, 1] - srt[:, 0] > 200)
        assert clear.sum() >= 4
        np.testing.assert_allclose(o2["scores"][clear], ref_score[clear], rtol=0, atol=5e-3)


def test_generate_max_size_1536_matches_oracle(cuda):
    """test.max_size is a free knob in the reference (crowdsam/utils.py:141-156, configs/crowdhuman.yaml:50).  At 1536 the
    768x1024 frame is cv2-enlarged to 1152x1536, Pillow-shrunk to 768x1024 for the encoder (device restatement, filter
    support 1.5), and postprocess_masks' second interpolate goes 768x1024 -> 1152x1536 (mask_post column strips beyond
    1024 pixels); masks, boxes and RLEs live in the 1152x1536 frame, boxes / points are divided by downscale 1.5."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle.make_goldens import PIPE_CFG, StandInDino, pipeline_image
    cfg = dict(PIPE_CFG)
    # stability threshold 0.008: PIPE_CFG's 0.004 sits exactly ON one mask's stability at this frame size (0.0040 here,
    # just below in the fp32 CPU oracle) -- a threshold-edge flip, not a difference in the path
    cfg.update(max_size=1536, max_prompts=16, min_mask_region_area=0, stability_score_thresh=0.008)
    img = pipeline_image()
    m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    np.random.seed(1)
    out = m.generate(img)
    assert m.image_hw == (1152, 1536) and abs(m.downscale - 1.5) < 1e-12
    D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
    np.random.seed(1)
    o = po.OracleCrowdSAM(synth.make_sam_state_dict(ARCH), (depth, heads, gidx), StandInDino(), cfg, rng=np.random)
    with torch.no_grad():
        ref = o.generate(img)
    print("max_size 1536: kept", out["boxes"].shape, "oracle", ref["boxes"].shape)
    assert out["boxes"].shape == ref["boxes"].shape and len(ref["boxes"]) > 0
    np.testing.assert_array_equal(out["points"], ref["points"])
    np.testing.assert_allclose(out["scores"], ref["scores"], rtol=0, atol=5e-3)
    assert np.abs(out["boxes"] - ref["boxes"]).max() <= 3.0 / 1.5
    assert all(r["size"] == [1152, 1536] for r in out["rles"])
    min_iou, max_xor = mask_agreement(out["rles"], ref["rles"], "max_size 1536")
    assert min_iou >= MASK_BOUNDS["max_size"][0] and max_xor <= MASK_BOUNDS["max_size"][1]


def test_predictor_box_prompt_api(cuda):
    """SamPredictor.predict(box=...) / predict_torch(None, None, boxes) (predictor.py:133-292): the numpy front-end scales the box
    into the input frame (ResizeLongestSide.apply_boxes) and returns the reference's four values; a box together with points is
    refused (eight tokens per prompt), mask prompts stay refused."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG
    m = CrowdSAM(_config(dict(PIPE_CFG)), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    img = synth.synthetic_crowd_frame(2, 1024, 60)[:768]
    p = m.predictor
    p.set_image(img)
    masks, iou, low, masks_t = p.predict(box=np.array([100, 80, 400, 500]), multimask_output=True)
    assert masks.shape == (4, 768, 1024) and masks.dtype == bool and iou.shape == (4,) and low.shape == (4, 256, 256)
    bt = torch.tensor([[100.0, 80.0, 400.0, 500.0], [10.0, 20.0, 300.0, 200.0]], device=cuda)
    mk, io, cl, lw = p.predict_torch(None, None, p.transform.apply_boxes_torch(bt, p.original_size), return_logits=True)
    assert mk.shape == (2, 4, 768, 1024) and io.shape == (2, 4) and cl.shape[:2] == (2, 4) and lw.shape == (2, 4, 256, 256)
    np.testing.assert_allclose(lw[0].cpu().numpy(), low, rtol=0, atol=1e-4)      # same box, either entry point
    with pytest.raises(NotImplementedError):
        p.predict_torch(torch.zeros(1, 1, 2, device=cuda), torch.ones(1, 1, device=cuda), bt[:1])
    with pytest.raises(NotImplementedError):
        p.predict(box=np.array([1, 2, 30, 40]), mask_input=np.zeros((1, 256, 256), np.float32))
