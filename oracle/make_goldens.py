"""AUTHORING-CONTAINER ONLY: run the *reference itself* (imported in place from /root/reference)
on seeded inputs and write the golden vectors committed under tests/golden/.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_goldens

Inputs are all derivable from seeds (crowdsam_amd/synth.py weights, RandomState images), so the
fixtures hold expected OUTPUTS only (strided samples + checksums where the tensor is large).
Tier O1 (zero-shim): modeling.*, build_sam._build_sam, utils.amg.*.
Tier O2 (shimmed): predictor.py + crowdsam/model.py end-to-end, with stand-ins (ours) for
torchvision / cv2 / loguru / pycocotools and a stand-in DINO -- pins the reference-owned control
flow (EPS loop, selection, filters), not the third-party arithmetic.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crowdsam_amd import synth  # noqa: E402
from oracle import ref_import, pipeline_oracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TEST_ARCH = "vit_test128"


def load_ref_sam(arch, n_class=1, seed=0):
    modeling, build, amg = ref_import.load_modeling()
    D, depth, heads, gidx = synth.SAM_CONFIGS[arch]
    sam = build._build_sam(D, depth, heads, n_class, list(gidx))
    sd = synth.make_sam_state_dict(arch, n_class, seed)
    sam.load_state_dict(sd, strict=True)
    sam.eval()
    return sam, sd, amg


def golden_encoder():
    sam, sd, _ = load_ref_sam(TEST_ARCH)
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, 3, 1024, 1024)).astype(np.float32))
    with torch.no_grad():
        y = sam.image_encoder(x)
    np.savez_compressed(os.path.join(OUT, "encoder_test128.npz"),
                        sample=y[:, ::4, ::4, ::4].numpy(), sum=np.float64(y.double().sum()),
                        abs_sum=np.float64(y.double().abs().sum()))
    print("encoder", y.shape, float(y.abs().mean()))


def decoder_inputs():
    rs = np.random.RandomState(11)
    emb = torch.from_numpy(rs.standard_normal((1, 256, 64, 64)).astype(np.float32))
    dino = torch.from_numpy(rs.standard_normal((1, 73, 73, 1024)).astype(np.float32))
    pts = rs.randint(0, 1024, size=(5, 1, 2)).astype(np.float64)
    return emb, dino, pts


def golden_decoder():
    sam, sd, _ = load_ref_sam(TEST_ARCH)
    emb, dino, pts = decoder_inputs()
    coords = torch.as_tensor(pts)
    labels = torch.ones(5, 1, dtype=torch.int)
    with torch.no_grad():
        sparse, dense = sam.prompt_encoder(points=(coords, labels), boxes=None, masks=None)
        pe = sam.prompt_encoder.get_dense_pe()
        low, iou, cls = sam.mask_decoder(image_embeddings=emb, image_pe=pe,
                                         sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                                         multimask_output=True, dino_feats=dino)
        m1 = sam.postprocess_masks(low, (1024, 768), (1024, 768))
        m2 = sam.postprocess_masks(low, (683, 1024), (682, 1023))   # trap 9: 1023-side original
        fg = torch.nn.functional.interpolate(
            sam.mask_decoder.point_classifier(sam.mask_decoder.dino_proj(dino)).permute(0, 3, 1, 2),
            (256, 256), mode="bilinear")
    np.savez_compressed(os.path.join(OUT, "decoder_test128.npz"),
                        sparse=sparse.numpy(), dense_pe_sample=pe[:, ::8, ::4, ::4].numpy(),
                        low_sample=low[:, :, ::8, ::8].numpy(), low_sum=low.double().sum((2, 3)).numpy(),
                        iou=iou.numpy(), cls=cls.numpy(),
                        post1_sample=m1[:, :, ::32, ::32].numpy(), post2_sample=m2[:, :, ::31, ::31].numpy(),
                        fg_sample=fg[:, :, ::8, ::8].numpy())
    print("decoder", low.shape, iou.flatten()[:4], cls.flatten()[:4])


def decoder_box_inputs(n=6):
    """Box prompts for the decoder case above: n seeded XYXY boxes in the 1024 input frame (x0 < x1, y0 < y1)."""
    rs = np.random.RandomState(21)
    a = rs.randint(0, 700, size=(n, 2)).astype(np.float64)
    wh = rs.randint(40, 320, size=(n, 2)).astype(np.float64)
    return np.concatenate([a, np.minimum(a + wh, 1023.0)], 1)


def golden_decoder_box():
    """Box prompts through the REFERENCE prompt encoder (prompt_encoder.py:95-102: two corner tokens, no padding point) +
    mask decoder, on the embedding / DINO tokens of decoder_inputs()."""
    sam, sd, _ = load_ref_sam(TEST_ARCH)
    emb, dino, _ = decoder_inputs()
    boxes = torch.as_tensor(decoder_box_inputs())
    with torch.no_grad():
        sparse, dense = sam.prompt_encoder(points=None, boxes=boxes, masks=None)
        pe = sam.prompt_encoder.get_dense_pe()
        low, iou, cls = sam.mask_decoder(image_embeddings=emb, image_pe=pe, sparse_prompt_embeddings=sparse,
                                         dense_prompt_embeddings=dense, multimask_output=True, dino_feats=dino)
    np.savez_compressed(os.path.join(OUT, "decoder_box_test128.npz"), sparse=sparse.numpy(),
                        low_sample=low[:, :, ::8, ::8].numpy(), low_sum=low.double().sum((2, 3)).numpy(),
                        iou=iou.numpy(), cls=cls.numpy())
    print("decoder_box", sparse.shape, low.shape, iou.flatten()[:4], cls.flatten()[:4])


def decoder_big_inputs(n=320):
    """Production-batch decoder case: the embedding / DINO tokens of decoder_inputs() and n seeded prompts (>= 256 so the
    persistent stream kernels -- t2i_stream / upscale_stream / i2t_rank -- are the ones that run on the GPU)."""
    emb, dino, _ = decoder_inputs()
    pts = np.random.RandomState(12).randint(0, 1024, size=(n, 1, 2)).astype(np.float64)
    return emb, dino, pts


def golden_decoder_big():
    """>= 320 prompts through the REFERENCE prompt encoder + mask decoder (mask_decoder.py:138-199) in the reference's
    own batches of 32 (prompts are independent in the decoder: batching only changes fp32 summation order)."""
    sam, sd, _ = load_ref_sam(TEST_ARCH)
    emb, dino, pts = decoder_big_inputs()
    lows, ious, clss = [], [], []
    with torch.no_grad():
        pe = sam.prompt_encoder.get_dense_pe()
        for i in range(0, len(pts), 32):
            coords = torch.as_tensor(pts[i:i + 32])
            labels = torch.ones(len(coords), 1, dtype=torch.int)
            sparse, dense = sam.prompt_encoder(points=(coords, labels), boxes=None, masks=None)
            low, iou, cls = sam.mask_decoder(image_embeddings=emb, image_pe=pe, sparse_prompt_embeddings=sparse,
                                             dense_prompt_embeddings=dense, multimask_output=True, dino_feats=dino)
            lows.append(low), ious.append(iou), clss.append(cls)
    low, iou, cls = torch.cat(lows), torch.cat(ious), torch.cat(clss)
    np.savez_compressed(os.path.join(OUT, "decoder_big_test128.npz"),
                        low_sample=low[:, :, 5::32, 9::32].numpy(), low_sum=low.double().sum((2, 3)).numpy(),
                        low_abs_sum=low.double().abs().sum((2, 3)).numpy(), low_max=low.amax((2, 3)).numpy(),
                        iou=iou.numpy(), cls=cls.numpy())
    print("decoder_big", low.shape, float(low.abs().mean()), iou[:2], cls[:2, :, 0])


def decoder_blob_inputs(n=24):
    """Decoder case for the blob-mask weight set (crowdsam_amd.synth.blob_heads): an image embedding shaped like the neck's output
    under those weights (per-pixel LayerNorm of seeded noise x the modified neck.3 gamma / beta: the kernel channels nearly empty,
    the blob channel empty), seeded DINO tokens, n seeded prompts (the frame centre among them)."""
    sd = synth.blob_heads(synth.make_sam_state_dict(TEST_ARCH))
    rs = np.random.RandomState(31)
    x = torch.from_numpy(rs.standard_normal((1, 256, 64, 64)).astype(np.float32))
    x = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-6)
    emb = x * sd["image_encoder.neck.3.weight"].view(1, -1, 1, 1) + sd["image_encoder.neck.3.bias"].view(1, -1, 1, 1)
    dino = torch.from_numpy(rs.standard_normal((1, 73, 73, 1024)).astype(np.float32))
    pts = rs.randint(0, 1024, size=(n, 1, 2)).astype(np.float64)
    pts[0, 0] = (512.0, 512.0)
    return sd, emb, dino, pts


def golden_decoder_blob():
    """The blob-mask weight set through the REFERENCE prompt encoder + mask decoder (mask_decoder.py:138-199): pins the oracle
    (and through it the HIP path, tests/test_pipeline_gpu.py::test_generate_blob_weights_matches_oracle) on the weights the
    bench runs since round 6."""
    modeling, build, amg = ref_import.load_modeling()
    D, depth, heads, gidx = synth.SAM_CONFIGS[TEST_ARCH]
    sd, emb, dino, pts = decoder_blob_inputs()
    sam = build._build_sam(D, depth, heads, 1, list(gidx))
    sam.load_state_dict(sd, strict=True)
    sam.eval()
    with torch.no_grad():
        coords = torch.as_tensor(pts)
        labels = torch.ones(len(pts), 1, dtype=torch.int)
        sparse, dense = sam.prompt_encoder(points=(coords, labels), boxes=None, masks=None)
        pe = sam.prompt_encoder.get_dense_pe()
        low, iou, cls = sam.mask_decoder(image_embeddings=emb, image_pe=pe, sparse_prompt_embeddings=sparse,
                                         dense_prompt_embeddings=dense, multimask_output=True, dino_feats=dino)
    area = (low > 0).sum((2, 3))
    inter, union = (low > 1).sum((2, 3)), (low > -1).sum((2, 3))
    np.savez_compressed(os.path.join(OUT, "decoder_blob_test128.npz"), low_sample=low[:, :, 1::8, 3::8].numpy(),
                        low_sum=low.double().sum((2, 3)).numpy(), low_abs_sum=low.double().abs().sum((2, 3)).numpy(),
                        area=area.numpy(), inter=inter.numpy(), union=union.numpy(), iou=iou.numpy(), cls=cls.numpy())
    yy, xx = torch.meshgrid(torch.arange(256.), torch.arange(256.), indexing="ij")
    m3 = (low[:, 3] > 0).float()
    cx = (m3 * xx).sum((1, 2)) / m3.sum((1, 2)).clamp(min=1) * 4
    cy = (m3 * yy).sum((1, 2)) / m3.sum((1, 2)).clamp(min=1) * 4
    off = torch.sqrt((cx - coords[:, 0, 0]) ** 2 + (cy - coords[:, 0, 1]) ** 2)
    print("decoder_blob", low.shape, "areas (256^2 px) median", area.median().item(), "stability median",
          float((inter.double() / union.clamp(min=1)).median()), "centroid offset of the largest blob from its prompt: median %.1f px" % off.median().item())


def golden_encoder_vitl():
    """Full-depth (24 blocks) ViT-L reference encoder on the BASELINE configs[1] input."""
    sam, sd, _ = load_ref_sam("vit_l")
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, 3, 1024, 1024)).astype(np.float32))
    with torch.no_grad():
        y = sam.image_encoder(x)
    np.savez_compressed(os.path.join(OUT, "encoder_vit_l.npz"), sample=y[:, ::4, 1::4, 2::4].numpy(),
                        sum=np.float64(y.double().sum()), abs_sum=np.float64(y.double().abs().sum()))
    print("encoder_vitl", y.shape, float(y.abs().mean()), float(y.std()))


def golden_amg():
    _, _, amg = ref_import.load_modeling()
    rs = np.random.RandomState(5)
    logits = torch.from_numpy((rs.standard_normal((6, 40, 56)) * 2).astype(np.float32))
    # smooth a little so there are real runs
    logits = torch.nn.functional.avg_pool2d(logits[None], 5, 1, 2)[0] * 3
    logits[4] = -5.0   # empty mask
    logits[5] = 5.0    # full mask
    stab = amg.calculate_stability_score(logits, 0.0, 1.0)
    masks = logits > 0
    boxes = amg.batched_mask_to_box(masks)
    rles = amg.mask_to_rle_pytorch(masks)
    crop = amg.generate_crop_boxes((445, 640), 2, 0.341)
    md = amg.MaskData(a=torch.arange(6), b=np.arange(6) * 2, c=list("abcdef"))
    md.filter(torch.tensor([True, False, True, True, False, True]))
    np.savez_compressed(os.path.join(OUT, "amg.npz"), stab=stab.numpy(), boxes=boxes.numpy(),
                        rle_counts=np.array([np.array(r["counts"]) for r in rles], dtype=object),
                        crop_boxes=np.array(crop[0]), crop_layers=np.array(crop[1]),
                        md_a=md["a"].numpy(), md_b=md["b"], md_c=np.array(md["c"]))
    print("amg", stab, boxes[:2])


# ------------------------------------------------------------------------------------------------
# Tier O2: full CrowdSAM.generate with stand-ins
# ------------------------------------------------------------------------------------------------
def _install_shims():
    from PIL import Image

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvo = types.ModuleType("torchvision.ops")
    tvb = types.ModuleType("torchvision.ops.boxes")

    def to_pil_image(a):
        return Image.fromarray(a)

    def resize(img, size):
        return img.resize((size[1], size[0]), Image.BILINEAR)

    tvf.resize, tvf.to_pil_image = resize, to_pil_image

    def batched_nms(boxes, scores, idxs, iou_threshold):
        assert int(idxs.abs().sum()) == 0
        return po.nms(boxes, scores, iou_threshold)

    def box_area(b):
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    tvb.batched_nms, tvb.box_area = batched_nms, box_area
    tvo.boxes = tvb
    def box_iou(a, b):      # torchvision.ops.box_iou semantics (public): areas (x2-x1)(y2-y1), intersection clamped at 0
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        return inter / (area_a[:, None] + area_b[None, :] - inter)

    tvo.box_iou = box_iou
    tvo.batched_nms = batched_nms
    tvt.functional = tvf
    for n in ("Compose", "Resize", "ToTensor", "Normalize"):
        setattr(tvt, n, object)
    tv.transforms, tv.ops = tvt, tvo
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "torchvision.ops": tvo,
                        "torchvision.ops.boxes": tvb})

    cv2 = types.ModuleType("cv2")

    def cv_resize(img, wh):
        # same size: identity copy (cv2's behaviour); otherwise OUR restatement of cv2's INTER_LINEAR
        # (oracle/resize_oracle.py) -- a stand-in, so fixtures pin the reference from the post-resize frame onward
        from oracle import resize_oracle
        return resize_oracle.cv2_resize_linear_u8(img, wh)

    def cc(working, conn):
        from scipy import ndimage
        assert conn == 8
        regions, n = ndimage.label(working, structure=np.ones((3, 3), dtype=np.uint8))
        sizes = np.bincount(regions.reshape(-1), minlength=n + 1)
        stats = np.zeros((n + 1, 5), dtype=np.int64)
        stats[:, -1] = sizes
        return n + 1, regions, stats, None

    cv2.resize, cv2.connectedComponentsWithStats = cv_resize, cc
    sys.modules["cv2"] = cv2

    lg = types.ModuleType("loguru")

    class _L:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    lg.logger = _L()
    sys.modules["loguru"] = lg

    pct = types.ModuleType("pycocotools")
    pcm = types.ModuleType("pycocotools.mask")
    pcm.frPyObjects = lambda rle, h, w: {"size": [h, w], "counts": po.coco_rle_string(rle["counts"]).encode()}
    pct.mask = pcm
    sys.modules.update({"pycocotools": pct, "pycocotools.mask": pcm})
    for n in ("matplotlib", "matplotlib.pyplot"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                sys.modules[n] = types.ModuleType(n)
    torch.Tensor.cuda = lambda self, *a, **k: self   # trap 2


class StandInDino:
    """Seeded stand-in for DINOv2 (source not under /root/reference): a fixed random projection of
    14x14 mean-pooled patches, so the features still depend on the image deterministically."""

    def __init__(self, seed=1):
        rs = np.random.RandomState(seed)
        self.w = torch.from_numpy(rs.standard_normal((3, 1024)).astype(np.float32))
        self.b = torch.from_numpy(rs.standard_normal((5329, 1024)).astype(np.float32))

    def forward_features(self, x):
        p = torch.nn.functional.avg_pool2d(x, 14, 14)            # [1,3,73,73]
        t = p.flatten(2).transpose(1, 2) @ self.w + self.b      # [1,5329,1024]
        return {"x_norm_patchtokens": t}

    def __call__(self, x):
        return self.forward_features(x)["x_norm_patchtokens"]

    def to(self, *a, **k):
        return self


PIPE_CFG = dict(grid_size=8, pos_sim_thresh=-1.0, points_per_batch=8, max_prompts=24,
                pred_iou_thresh=0.4, stability_score_thresh=0.004, filter_thresh=0.45,
                min_mask_region_area=30, box_nms_thresh=1.0, crop_nms_thresh=1.0)


def pipeline_image():
    return synth.synthetic_crowd_frame(3, size=1024, n_ellipses=60)[:768]   # 768 x 1024 (h x w)


def golden_pipeline(fuse=False):
    import importlib
    _install_shims()
    ref_import.load_modeling()
    predictor_mod = importlib.import_module("segment_anything_cs.predictor")
    sys.path.insert(0, ref_import.REF)
    model_mod = importlib.import_module("crowdsam.model")
    sys.path.remove(ref_import.REF)
    sam, sd, _ = load_ref_sam(TEST_ARCH)
    predictor = predictor_mod.SamPredictor(sam, StandInDino())
    cs = object.__new__(model_mod.CrowdSAM)
    cfg = dict(po.DEFAULT_TEST_CFG)
    cfg.update(PIPE_CFG)
    cfg["fuse_simmap"] = fuse
    cs.device = torch.device("cpu")
    cs.train_free = False
    cs.predictor = predictor
    for k, v in cfg.items():
        setattr(cs, k, v)
    np.random.seed(42)
    img = pipeline_image()
    with torch.no_grad():
        out = cs.generate(img)
    res = {k: out[k] for k in ("boxes", "scores", "categories", "points", "stability_score")}
    rle_counts = np.array([r["counts"] for r in out["rles"]], dtype=object)
    name = "pipeline_test128_fuse.npz" if fuse else "pipeline_test128.npz"
    np.savez_compressed(os.path.join(OUT, name), rle_counts=rle_counts, **res)
    print("pipeline", name, {k: v.shape for k, v in res.items()}, out["scores"][:5])


EPS_SHIPPED_FRAGILE = 1.0      # |max logit over the passing masks at a list point| below this goes into the fragile list


def golden_pipeline_eps_shipped():
    """The reference's own CrowdSAM.generate with the SHIPPED test block (configs/crowdhuman.yaml:33-58 unchanged: grid
    192, max_prompts 500, 32 prompts per batch, filter_thresh 0.7, pos_sim_thresh 0.5, stability 0.8 / offset 1,
    pred_iou 0.1, box NMS 0.65, min region 100) on the vit_test128 encoder: 16 sequential pruned batches
    (crowdsam/model.py:226-248).  Random weights make those thresholds degenerate, so the decoder heads are the
    calibrated seeded variant synth.shipped_scale_heads (documented there); nothing else differs from a shipped run.
    The reference is instrumented from outside (np.random.shuffle, predictor.predict_torch and the instance's
    _process_batch are wrapped; no reference source is edited) to record, per batch: the prompt list, every prompt's
    PWD-Net choice / fused score / top-2 margin / stability, who survives the filters, who feeds the occupancy mask, the
    occupancy decision of EVERY point of the shuffled list and -- sparse -- the decision margins (max logit over the
    feeding masks at the point) below EPS_SHIPPED_FRAGILE.  The chain is re-simulated from those records and asserted to
    reproduce the reference's own batches before anything is written."""
    import importlib
    _install_shims()
    _, build, amg = ref_import.load_modeling()
    predictor_mod = importlib.import_module("segment_anything_cs.predictor")
    sys.path.insert(0, ref_import.REF)
    model_mod = importlib.import_module("crowdsam.model")
    sys.path.remove(ref_import.REF)
    D, depth, heads, gidx = synth.SAM_CONFIGS[TEST_ARCH]
    sam = build._build_sam(D, depth, heads, 1, list(gidx))
    sam.load_state_dict(synth.shipped_scale_heads(synth.make_sam_state_dict(TEST_ARCH)), strict=True)
    sam.eval()
    predictor = predictor_mod.SamPredictor(sam, StandInDino())
    cs = object.__new__(model_mod.CrowdSAM)
    cfg = dict(po.DEFAULT_TEST_CFG)            # == the shipped test: block
    cs.device, cs.train_free, cs.predictor = torch.device("cpu"), False, predictor
    for k, v in cfg.items():
        setattr(cs, k, v)
    rec = {"batches": []}
    orig_shuffle, orig_predict = np.random.shuffle, predictor.predict_torch

    def shuffle(a):
        orig_shuffle(a)
        rec["list"] = a.copy()

    def predict_torch(*a, **k):
        out = orig_predict(*a, **k)
        rec["cur"] = out[:3]
        return out

    def process_batch(points, im_size, crop_box):
        data = model_mod.CrowdSAM._process_batch(cs, points, im_size, crop_box)
        masks, iou, cls = rec.pop("cur")
        L = rec["list"]
        B = len(points)
        fused = torch.clamp(iou, 0.) * cls.squeeze(2).sigmoid()
        srt = fused.sort(dim=-1, descending=True)[0]
        sel = fused.max(dim=-1)[1]
        sl = masks[torch.arange(B), sel]                                   # [B, H, W] logits of the chosen candidates
        inter = (sl > 1.0).flatten(1).sum(-1).int()
        union = (sl > -1.0).flatten(1).sum(-1).int()
        surv_pts = {tuple(int(v) for v in p) for p in data["points"].numpy()}
        assert len({tuple(p) for p in points.tolist()}) == B              # prompts of a batch are distinct pixels
        survive = np.array([tuple(int(v) for v in p) in surv_pts for p in points], dtype=bool)
        score = fused[torch.arange(B), sel]
        feeds = torch.as_tensor(survive) & (score > cs.filter_thresh)
        G = sl[:, L[:, 1], L[:, 0]]                                        # [B, P] chosen-candidate logits at every list point
        m = G[feeds].max(0)[0] if bool(feeds.any()) else torch.full((len(L),), -1e30)
        occ = (m > 0).numpy()
        ref_occ = (data["masks"][data["iou_preds"] > cs.filter_thresh]).any(0)[L[:, 1], L[:, 0]].numpy()   # :246
        assert np.array_equal(occ, ref_occ)
        assert int(survive.sum()) == len(data["points"]) and int(feeds.sum()) == int((data["iou_preds"] > cs.filter_thresh).sum())
        rec["batches"].append(dict(points=np.array(points), sel=sel.numpy(), score=score.numpy(),
                                   top2=(srt[:, 0] - srt[:, 1]).numpy(), inter=inter.numpy(), union=union.numpy(),
                                   survive=survive, feeds=feeds.numpy(), occ=occ, margin=m.numpy()))
        print("  batch %2d: %2d survive, %2d feed the occupancy (%.3f of the list points)"
              % (len(rec["batches"]), survive.sum(), int(feeds.sum()), occ.mean()), flush=True)
        return data

    np.random.shuffle, predictor.predict_torch, cs._process_batch = shuffle, predict_torch, process_batch
    np.random.seed(42)
    try:
        with torch.no_grad():
            out = cs.generate(pipeline_image())
    finally:
        np.random.shuffle = orig_shuffle
    L, bt = rec["list"], rec["batches"]
    # re-simulate the sampler from the records: must reproduce the reference's own batches (crowdsam/model.py:229-240)
    alive, alive_after, last_pos = np.arange(len(L)), [], 0
    for b in bt:
        n = len(b["points"])
        assert np.array_equal(L[alive[:n]], b["points"])
        last_pos = int(alive[n - 1])
        alive = alive[n:]
        alive = alive[~b["occ"][alive]]
        alive_after.append(len(alive))
    nb = len(bt)
    assert nb == 16 and all(len(b["points"]) == 32 for b in bt), [len(b["points"]) for b in bt]
    fb, fp, fm = [], [], []
    for i, b in enumerate(bt):
        idx = np.nonzero(np.abs(b["margin"]) < EPS_SHIPPED_FRAGILE)[0]
        fb.append(np.full(len(idx), i, np.int8)), fp.append(idx.astype(np.int32)), fm.append(b["margin"][idx].astype(np.float32))
    st = lambda k, dt: np.stack([b[k] for b in bt]).astype(dt)
    res = {k: out[k] for k in ("boxes", "scores", "categories", "points", "stability_score")}
    np.savez_compressed(
        os.path.join(OUT, "pipeline_eps_shipped.npz"), list=L.astype(np.int16), batch_points=st("points", np.int16),
        sel=st("sel", np.int8), score=st("score", np.float32), top2=st("top2", np.float32), inter=st("inter", np.int32),
        union=st("union", np.int32), survive=st("survive", np.uint8), feeds=st("feeds", np.uint8),
        occ_bits=np.stack([np.packbits(b["occ"]) for b in bt]), alive_after=np.array(alive_after, np.int32),
        last_pos=np.int32(last_pos), fragile_batch=np.concatenate(fb), fragile_point=np.concatenate(fp),
        fragile_margin=np.concatenate(fm), fragile_below=np.float32(EPS_SHIPPED_FRAGILE),
        rle_counts=np.array([r["counts"] for r in out["rles"]], dtype=object), **res)
    print("pipeline_eps_shipped: %d list points, alive after each batch %s, last list position consumed %d, %d fragile "
          "records, kept %s" % (len(L), alive_after, last_pos, sum(len(x) for x in fp), res["boxes"].shape))



VITB_CFG = dict(grid_size=8, pos_sim_thresh=-1.0, points_per_batch=32, max_prompts=500, pred_iou_thresh=-1e9,
                stability_score_thresh=0.0, min_mask_region_area=0, box_nms_thresh=1.0, crop_nms_thresh=1.0)


def vitb_image():
    return np.random.RandomState(0).randint(0, 256, (512, 512, 3)).astype(np.uint8)


def golden_pipeline_vitb():
    """BASELINE configs[0]: ViT-B via _build_sam(768,12,12,1,[2,5,8,11]) + 8x8 grid on one 512x512 seeded image through
    the reference's CrowdSAM.generate (tier O2).  The 512 -> 1024 up-scaling goes through OUR cv2 stand-in, so the
    fixture pins the reference from the post-resize frame onward."""
    import importlib
    _install_shims()
    ref_import.load_modeling()
    predictor_mod = importlib.import_module("segment_anything_cs.predictor")
    sys.path.insert(0, ref_import.REF)
    model_mod = importlib.import_module("crowdsam.model")
    sys.path.remove(ref_import.REF)
    sam, sd, _ = load_ref_sam("vit_b")
    predictor = predictor_mod.SamPredictor(sam, StandInDino())
    cs = object.__new__(model_mod.CrowdSAM)
    cfg = dict(po.DEFAULT_TEST_CFG)
    cfg.update(VITB_CFG)
    cs.device = torch.device("cpu")
    cs.train_free = False
    cs.predictor = predictor
    for k, v in cfg.items():
        setattr(cs, k, v)
    np.random.seed(42)
    with torch.no_grad():
        out = cs.generate(vitb_image())
    res = {k: out[k] for k in ("boxes", "scores", "categories", "points", "stability_score")}
    # random-weight ViT-B masks are noise-like (1e4+ runs each): keep the run COUNT and the mask area per mask only
    dec = [po.coco_rle_decode(r["counts"], *r["size"]) for r in out["rles"]]
    res["mask_area"] = np.array([int(d.sum()) for d in dec], dtype=np.int64)
    # ... and every 8th pixel of every 8th row of each mask (128 x 128 bits per mask): a pixel-level sample of the masks
    res["mask_sample"] = np.packbits(np.stack([np.asarray(d)[::8, ::8].astype(bool) for d in dec]))
    np.savez_compressed(os.path.join(OUT, "pipeline_vit_b_512.npz"), **res)
    print("pipeline_vitb", {k: v.shape for k, v in res.items()}, out["scores"][:8], out["boxes"][:3])


def golden_dino_hf():
    """Second opinion for the DINOv2 restatement (the reference's dinov2/ submodule is empty and unpinned): the seeded
    DINOv2 ViT-L/14 state dict mapped into transformers.models.dinov2 (v5.15, present in this image), full depth, on a
    1022x1022 input.  Fixture = strided x_norm_patchtokens.  Still *unpinned* w.r.t. the un-vendored submodule, but
    no longer self-referential: an independent implementation of the same published architecture."""
    from transformers import Dinov2Config, Dinov2Model
    from oracle import sam_oracle as so
    depth = 24
    sd = synth.make_dino_state_dict()
    cfg = Dinov2Config(hidden_size=1024, num_hidden_layers=depth, num_attention_heads=16, mlp_ratio=4, patch_size=14,
                       image_size=518, qkv_bias=True, use_swiglu_ffn=False, layer_norm_eps=1e-6, hidden_act="gelu")
    m = Dinov2Model(cfg).eval()
    hf = {"embeddings.cls_token": sd["cls_token"], "embeddings.mask_token": sd["mask_token"],
          "embeddings.position_embeddings": sd["pos_embed"],
          "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
          "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
          "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    for i in range(depth):
        a, b = f"blocks.{i}.", f"encoder.layer.{i}."
        qw, kw, vw = sd[a + "attn.qkv.weight"].chunk(3, 0)
        qb, kb, vb = sd[a + "attn.qkv.bias"].chunk(3, 0)
        for n, w, bb in (("query", qw, qb), ("key", kw, kb), ("value", vw, vb)):
            hf[b + f"attention.attention.{n}.weight"], hf[b + f"attention.attention.{n}.bias"] = w, bb
        for src, dst in (("norm1", "norm1"), ("norm2", "norm2"), ("attn.proj", "attention.output.dense"),
                         ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
            hf[b + dst + ".weight"], hf[b + dst + ".bias"] = sd[a + src + ".weight"], sd[a + src + ".bias"]
        hf[b + "layer_scale1.lambda1"], hf[b + "layer_scale2.lambda1"] = sd[a + "ls1.gamma"], sd[a + "ls2.gamma"]
    missing, unexpected = m.load_state_dict(hf, strict=True), None
    img = synth.synthetic_crowd_frame(5, 1024, 40)[:768]
    x = so.preprocess(torch.from_numpy(img).permute(2, 0, 1).float().contiguous())[None]
    xd = torch.nn.functional.interpolate(x, (1022, 1022), mode="bilinear")
    with torch.no_grad():
        y = m(pixel_values=xd, interpolate_pos_encoding=True).last_hidden_state[:, 1:]
        o_size = so.dinov2_forward(sd, xd, depth=depth, pos_offset=None)
        o_off = so.dinov2_forward(sd, xd, depth=depth, pos_offset=0.1)
    e_size, e_off = (y - o_size).abs(), (y - o_off).abs()
    print("dino_hf: HF vs oracle(size= form) max %.3g mean %.3g | vs oracle(+0.1 form) max %.3g mean %.3g | |y| mean %.3g"
          % (e_size.max(), e_size.mean(), e_off.max(), e_off.mean(), y.abs().mean()))
    np.savez_compressed(os.path.join(OUT, "dino_hf_vitl14.npz"), sample=y[0, ::7, ::8].numpy(),
                        sum=np.float64(y.double().sum()), abs_sum=np.float64(y.double().abs().sum()),
                        err_vs_oracle_size=np.array([e_size.max(), e_size.mean()]),
                        err_vs_oracle_offset=np.array([e_off.max(), e_off.mean()]))


def _hf_dino(depth=24):
    from transformers import Dinov2Config, Dinov2Model
    sd = synth.make_dino_state_dict()
    cfg = Dinov2Config(hidden_size=1024, num_hidden_layers=depth, num_attention_heads=16, mlp_ratio=4, patch_size=14,
                       image_size=518, qkv_bias=True, use_swiglu_ffn=False, layer_norm_eps=1e-6, hidden_act="gelu")
    m = Dinov2Model(cfg).eval()
    hf = {"embeddings.cls_token": sd["cls_token"], "embeddings.mask_token": sd["mask_token"],
          "embeddings.position_embeddings": sd["pos_embed"],
          "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
          "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
          "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    for i in range(depth):
        a, b = f"blocks.{i}.", f"encoder.layer.{i}."
        qw, kw, vw = sd[a + "attn.qkv.weight"].chunk(3, 0)
        qb, kb, vb = sd[a + "attn.qkv.bias"].chunk(3, 0)
        for n, w, bb in (("query", qw, qb), ("key", kw, kb), ("value", vw, vb)):
            hf[b + f"attention.attention.{n}.weight"], hf[b + f"attention.attention.{n}.bias"] = w, bb
        for src, dst in (("norm1", "norm1"), ("norm2", "norm2"), ("attn.proj", "attention.output.dense"),
                         ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
            hf[b + dst + ".weight"], hf[b + dst + ".bias"] = sd[a + src + ".weight"], sd[a + src + ".bias"]
        hf[b + "layer_scale1.lambda1"], hf[b + "layer_scale2.lambda1"] = sd[a + "ls1.gamma"], sd[a + "ls2.gamma"]
    m.load_state_dict(hf, strict=True)
    return m, sd


def _hf_dino_offset_form(m, sd, xd, offset=0.1):
    """transformers' DINOv2 blocks with the position embedding interpolated the way the dinov2 hub models of the
    reference's era do it (interpolate_offset = 0.1: F.interpolate(scale_factor=((g + 0.1) / 37), bicubic) -- the form
    `configs/crowdhuman_mi355x.yaml: model.dino_pos_offset: 0.1` ships): the block arithmetic is the independent
    implementation, only the interpolated table is injected."""
    from oracle import sam_oracle as so
    gh, gw = xd.shape[2] // 14, xd.shape[3] // 14
    table = so.dino_interp_pos_embed(sd["pos_embed"], gh, gw, offset)
    m.embeddings.interpolate_pos_encoding = lambda emb, h, w: table.to(emb.dtype)
    with torch.no_grad():
        return m(pixel_values=xd, interpolate_pos_encoding=True).last_hidden_state[:, 1:]


def full_frame(cfg_index):
    """The frames of the full-composition GPU tests (tests/test_full_composition_gpu.py)."""
    return synth.synthetic_crowd_frame(7, 1024, 150) if cfg_index == 2 else synth.synthetic_crowd_frame(4, 1500, 400)


def golden_full_vitl():
    """BASELINE configs[2] at full composition, stage outputs INSIDE the pipeline: the reference's 24-block ViT-L encoder
    and (unpinned by nature, independent implementation) transformers' DINOv2-L x24 with the shipped (+0.1 offset)
    position-embedding form, both on the preprocessed 1024^2 synthetic crowd frame the GPU test feeds CrowdSAM.generate."""
    from oracle import sam_oracle as so
    img = full_frame(2)
    x = so.preprocess(torch.from_numpy(img).permute(2, 0, 1).float().contiguous())[None]
    sam, sd, _ = load_ref_sam("vit_l")
    with torch.no_grad():
        y = sam.image_encoder(x)
    m, dsd = _hf_dino()
    xd = torch.nn.functional.interpolate(x, (1022, 1022), mode="bilinear")
    d = _hf_dino_offset_form(m, dsd, xd)
    np.savez_compressed(os.path.join(OUT, "full_vit_l.npz"), feat_sample=y[:, ::4, 1::4, 2::4].numpy(),
                        feat_abs_sum=np.float64(y.double().abs().sum()), dino_sample=d[0, ::7, ::8].numpy(),
                        dino_abs_sum=np.float64(d.double().abs().sum()))
    print("full_vitl: |feat| %.4f |dino| %.4f" % (float(y.abs().mean()), float(d.abs().mean())))


def golden_full_vith():
    """BASELINE configs[4]: the reference's 32-block ViT-H encoder on the 1500^2 stress frame after the cv2-style
    down-scale to 1024 (oracle/resize_oracle.py; the device resize is bit-exact against it: tests/test_resize_gpu.py)."""
    from oracle import sam_oracle as so, resize_oracle as ro
    img, _ = ro.resize_image(full_frame(4), 1024)[:2]
    assert img.shape == (1024, 1024, 3)
    x = so.preprocess(torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().contiguous())[None]
    sam, sd, _ = load_ref_sam("vit_h")
    with torch.no_grad():
        y = sam.image_encoder(x)
    np.savez_compressed(os.path.join(OUT, "full_vit_h.npz"), feat_sample=y[:, ::4, 1::4, 2::4].numpy(),
                        feat_abs_sum=np.float64(y.double().abs().sum()))
    print("full_vith: |feat| %.4f" % float(y.abs().mean()))


def golden_pipeline_fuse():
    """Same run with test.fuse_simmap = True (crowdsam/model.py:273-286; its hard-coded .cuda() is the identity
    under the trap-2 shim)."""
    golden_pipeline(fuse=True)


def evaluator_dataset(seed=0, n_img=14):
    """Synthetic CrowdHuman-style GT (.odgt records) and COCO-format detections: crowded boxes, ignore regions,
    score ties, detections hanging over the image border, one image without detections."""
    rs = np.random.RandomState(seed)
    records, images, annots = [], [], []
    aid = 0
    for i in range(n_img):
        w, h = int(rs.randint(400, 1200)), int(rs.randint(300, 900))
        name = "img%03d,%05x" % (i, rs.randint(0, 1 << 20))
        n_gt = int(rs.randint(3, 40))
        gts = []
        for _ in range(n_gt):
            bw, bh = int(rs.randint(15, 160)), int(rs.randint(30, 320))
            x, y = int(rs.randint(-20, w - 10)), int(rs.randint(-20, h - 10))
            vis = [x + int(rs.randint(0, 6)), y + int(rs.randint(0, 6)), max(4, bw - int(rs.randint(0, 12))),
                   max(4, bh - int(rs.randint(0, 12)))]
            tag = "person" if rs.rand() < 0.85 else "mask"
            rb = {"tag": tag, "fbox": [x, y, bw, bh], "vbox": vis, "hbox": [x, y, bw // 3, bh // 6]}
            if rs.rand() < 0.12:
                rb["extra"] = {"ignore": 1}
            elif rs.rand() < 0.5:
                rb["extra"] = {"ignore": 0, "box_id": 1}
            gts.append(rb)
        if not any(g["tag"] == "person" and g.get("extra", {}).get("ignore", 0) == 0 for g in gts):
            gts[0]["tag"] = "person"
            gts[0].pop("extra", None)
        records.append({"ID": name, "gtboxes": gts})
        images.append({"id": name, "file_name": name + ".jpg", "width": w, "height": h})
        if i == 5:
            continue                                    # an image the detector returned nothing for
        dets = []
        for g in gts:                                   # jittered true positives, duplicates, misses
            if rs.rand() < 0.8:
                for _ in range(1 + (rs.rand() < 0.3)):
                    x, y, bw, bh = g["vbox"]
                    j = rs.normal(0, 0.12, 4)
                    dets.append([x + j[0] * bw, y + j[1] * bh, bw * (1 + j[2]), bh * (1 + j[3])])
        for _ in range(int(rs.randint(0, 15))):         # clutter
            dets.append([rs.uniform(-30, w), rs.uniform(-30, h), rs.uniform(10, 200), rs.uniform(10, 300)])
        scores = np.round(rs.rand(len(dets)), 2)        # two decimals: plenty of exact ties
        for d, sc in zip(dets, scores):
            annots.append({"category_id": 1, "bbox": [float(v) for v in d], "image_id": name, "iscrowd": False,
                           "area": float(d[2] * d[3]), "id": aid, "score": float(sc)})
            aid += 1
    coco = {"images": [im for k, im in enumerate(images) if k != 5], "annotations": annots,
            "categories": [{"id": 1, "name": "person"}]}
    return records, coco


def mask_nms_inputs(seed=11, n=48, hw=(97, 211)):
    """Overlapping blob masks (several near-duplicates, one empty mask) with distinct scores."""
    rs = np.random.RandomState(seed)
    H, W = hw
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.zeros((n, H, W), bool)
    for i in range(n):
        if i % 3 == 1:                                   # jittered copy of the previous blob
            cy, cx, ry, rx = prev
            cy, cx = cy + rs.uniform(-4, 4), cx + rs.uniform(-6, 6)
            ry, rx = ry * rs.uniform(0.7, 1.2), rx * rs.uniform(0.7, 1.2)
        else:
            cy, cx, ry, rx = rs.uniform(0, H), rs.uniform(0, W), rs.uniform(4, 30), rs.uniform(4, 50)
        prev = (cy, cx, ry, rx)
        masks[i] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
    masks[17] = False
    scores = rs.permutation(n).astype(np.float32) / n
    return masks, scores


def golden_mask_nms():
    """Tier O2 import (crowdsam/utils.py pulls cv2 / loguru / torchvision at module level; none of them is touched
    by the three functions captured here, which are pure torch/numpy)."""
    _install_shims()
    sys.path.insert(0, "/root/reference")
    import importlib
    ru = importlib.import_module("crowdsam.utils")
    assert ru.__file__.startswith("/root/reference")
    masks, scores = mask_nms_inputs()
    mt = torch.from_numpy(masks)
    res = {"masks_packed": np.packbits(masks), "shape": np.array(masks.shape), "scores": scores}
    for thr in (0.3, 0.5, 0.8):
        res["keep_%02d" % int(thr * 100)] = np.asarray(ru.mask_iou_nms(torch.zeros(len(masks), 4), scores, mt, thr))
    a, b = mt[:8].unsqueeze(1), mt[None, 8:20]
    res["coverage"] = ru.coverage(a, b).numpy()
    res["mask_iou"] = ru.mask_iou(a, b).numpy()
    np.savez_compressed(os.path.join(OUT, "mask_nms.npz"), **res)
    print("mask_nms", {k: (v.shape if hasattr(v, "shape") else v) for k, v in res.items() if k.startswith("keep")})


def tools_inputs(seed=3):
    """Detections / gt boxes for evaluate_boxes (duplicates, score ties, unmatched gt, crowd overlaps) and a per-image
    result list + GT json for convert_to_coco."""
    rs = np.random.RandomState(seed)
    cases = []
    for n_gt, n_pred in ((6, 9), (0, 4), (5, 0), (12, 30), (3, 3)):
        gt = rs.uniform(0, 300, (n_gt, 2))
        gt = np.concatenate([gt, gt + rs.uniform(20, 120, (n_gt, 2))], 1)
        if n_gt and n_pred:
            src = gt[rs.randint(0, n_gt, n_pred)] + rs.normal(0, 8, (n_pred, 4))
            src[::4] = rs.uniform(0, 400, (len(src[::4]), 4))
            src[::4, 2:] += src[::4, :2]
        else:
            src = rs.uniform(0, 300, (n_pred, 4))
            src[:, 2:] += src[:, :2] + 10
        scores = np.round(rs.rand(n_pred), 1)             # one decimal: exact ties
        cases.append((src.astype(np.float32), scores.astype(np.float32), gt))
    det = [{"image_id": "ignored", "boxes": [[1.5, 2.0, 11.5, 22.0], [5, 5, 6, 9]], "scores": [0.9, 0.25]},
           {"image_id": "ignored", "boxes": [], "scores": []},
           {"image_id": "ignored", "boxes": [[0, 0, 100.25, 50]], "scores": [0.5]}]
    gt_js = {"images": [{"id": 7, "file_name": "273271,c9db000d5146c15.jpg"}, {"id": 8, "file_name": "b.jpg"},
                        {"id": 9, "file_name": "dir_c.png"}], "categories": [{"id": 1, "name": "person"}],
             "annotations": []}
    return cases, det, gt_js


def golden_tools():
    """Tier O2: crowdsam/utils.py::evaluate_boxes (tools/test.py:80) with our box_iou stand-in, and
    tools/batch_eval.py::convert_to_coco / merge-free parts (pure python).  Fixture: tests/golden/tools.json."""
    import copy
    import importlib
    import json
    _install_shims()
    sys.path.insert(0, "/root/reference")
    ru = importlib.import_module("crowdsam.utils")
    assert ru.__file__.startswith("/root/reference")
    spec = importlib.util.spec_from_file_location("ref_batch_eval", "/root/reference/tools/batch_eval.py")
    be = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(be)
    sys.path.remove("/root/reference")
    cases, det, gt_js = tools_inputs()
    ev = []
    for pb, ps, gt in cases:
        for thr in (0.5, 0.3):
            p, r, fp, fn = ru.evaluate_boxes(pb, ps, gt, thr)
            ev.append({"precision": float(p), "recall": float(r), "FP": [int(v) for v in fp], "FN": [int(v) for v in fn]})
    coco = be.convert_to_coco(copy.deepcopy(det), copy.deepcopy(gt_js))
    cfg = ru.modify_config({"test": {"grid_size": 192}}, ["test.grid_size", "64", "new.section.flag", "TRUE",
                                                          "test.pos_sim_thresh", "-1.5", "model.sam_model", "vit_b"])
    conv = [ru.convert_value(v) for v in ("True", "false", "12", "-3.5", "1e-3", "vit_l", "FALSE")]
    meta = {k: [v[0], v[1], len(v[2])] for k, v in ru.data_meta.items()}
    with open(os.path.join(OUT, "tools.json"), "w") as f:
        json.dump({"evaluate_boxes": ev, "convert_to_coco": coco, "modify_config": cfg, "convert_value": conv,
                   "data_meta": meta}, f)
    print("tools", ev[:2], len(coco["annotations"]), cfg, conv)


def golden_evaluator():
    """Tier O1: tools/crowdhuman_eval.py is pure numpy and imports with zero shims.  Fixture = the synthetic
    GT/detection files (data) + AP / MR / recall / tp / fp and the curves the reference computes from them."""
    import importlib.util
    import json
    d = os.path.join(OUT, "crowdhuman_eval")
    os.makedirs(d, exist_ok=True)
    records, coco = evaluator_dataset()
    gt_file, dt_file = os.path.join(d, "gt.odgt"), os.path.join(d, "det.json")
    with open(gt_file, "w") as f:
        f.write("\n".join(json.dumps(r) for r in records) + "\n")
    with open(dt_file, "w") as f:
        json.dump(coco, f)
    spec = importlib.util.spec_from_file_location("ref_crowdhuman_eval", "/root/reference/tools/crowdhuman_eval.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = {}
    for rm in (False, True):
        for vis in (False, True):
            mod.gt_path = gt_file          # Database.__init__ reads the module-level name (crowdhuman_eval.py:368,372)
            db = mod.Database(gt_file, dt_file, "boxes", None, 0, rm, visible_flag=vis)
            db.compare()
            ap, recall, data = db.eval_AP()
            mr, _, (tp, fp) = db.eval_MR(fppiX=data[-2], fppiY=data[-1])
            key = "rm%d_vis%d" % (rm, vis)
            res[key + "_summary"] = np.array([ap, mr, recall, tp, fp], dtype=np.float64)
            res[key + "_labels"] = np.array([it[1] for it in db.scorelist], dtype=np.int8)
            res[key + "_scores"] = np.array([it[0][-1] for it in db.scorelist], dtype=np.float64)
            res[key + "_recall"] = np.array(data[0], dtype=np.float64)
            res[key + "_precision"] = np.array(data[1], dtype=np.float64)
            print("evaluator", key, res[key + "_summary"])
    np.savez_compressed(os.path.join(OUT, "crowdhuman_eval.npz"), **res)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["amg", "decoder", "encoder", "pipeline", "evaluator", "mask_nms"]
    for w in which:
        globals()["golden_" + w]()
