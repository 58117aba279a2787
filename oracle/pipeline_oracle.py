"""ORACLE (test infrastructure, NOT product code): CPU restatement of the Crowd-SAM driver --
EPS loop, PWD-Net selection, filters, NMS, small-region post-processing, RLE packing.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
Each function cites the reference lines it follows (paths relative to /root/reference).

Third-party arithmetic the reference reaches through un-vendored wheels is restated from the
published algorithms and is "parity unpinned" (SURVEY.md §8c): torchvision.ops.nms (greedy,
suppress IoU > thr, stable descending score order), cv2.connectedComponentsWithStats
(8-connectivity; scipy.ndimage.label here), pycocotools rleToString, cv2.resize (oracle/resize_oracle.py;
the reference-pinned fixtures start at the post-resize frame).  Pillow's bilinear resize IS pinned (against Pillow).
"""
import math
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F

from . import sam_oracle as so


# ------------------------------------------------------------------------------------------------
# amg.py tensor utilities
# ------------------------------------------------------------------------------------------------
def calculate_stability_score(masks, mask_threshold, offset):
    """segment_anything_cs/utils/amg.py:156-176."""
    inter = (masks > (mask_threshold + offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    union = (masks > (mask_threshold - offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    return inter / union


def stability_counts(masks, mask_threshold, offset):
    inter = (masks > (mask_threshold + offset)).flatten(1).sum(-1).to(torch.int32)
    union = (masks > (mask_threshold - offset)).flatten(1).sum(-1).to(torch.int32)
    return inter, union


def batched_mask_to_box(masks):
    """amg.py:303-346 for [N,H,W] bool: XYXY inclusive max index, empty -> zeros (int64)."""
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4)
    h, w = masks.shape[-2:]
    in_h, _ = torch.max(masks, dim=-1)
    hc = in_h * torch.arange(h)[None, :]
    bottom, _ = torch.max(hc, dim=-1)
    top, _ = torch.min(hc + h * (~in_h), dim=-1)
    in_w, _ = torch.max(masks, dim=-2)
    wc = in_w * torch.arange(w)[None, :]
    right, _ = torch.max(wc, dim=-1)
    left, _ = torch.min(wc + w * (~in_w), dim=-1)
    empty = (right < left) | (bottom < top)
    out = torch.stack([left, top, right, bottom], dim=-1)
    return out * (~empty).unsqueeze(-1)


def mask_to_rle(masks):
    """amg.py:107-135 mask_to_rle_pytorch: column-major uncompressed RLE per mask."""
    out = []
    m = masks.numpy() if isinstance(masks, torch.Tensor) else np.asarray(masks)
    for i in range(m.shape[0]):
        h, w = m[i].shape
        flat = m[i].T.reshape(-1)
        change = np.nonzero(flat[1:] != flat[:-1])[0] + 1
        idx = np.concatenate([[0], change, [h * w]])
        counts = [] if flat[0] == 0 else [0]
        counts.extend(np.diff(idx).tolist())
        out.append({"size": [h, w], "counts": counts})
    return out


def coco_rle_string(counts):
    """pycocotools rleToString (published algorithm): 6 bits/char, delta vs counts[i-2] for i>2."""
    s = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            s.append(chr(ch + 48))
    return "".join(s)


def coco_rle_decode(counts_str, h, w):
    """pycocotools rleFrString + rleDecode (published algorithm): compressed string -> bool [h, w]."""
    if isinstance(counts_str, bytes):
        counts_str = counts_str.decode("ascii")
    counts, p = [], 0
    while p < len(counts_str):
        x, k, more = 0, 0, True
        while more:
            c = ord(counts_str[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    flat = np.zeros(h * w, dtype=bool)
    idx, val = 0, False
    for c in counts:
        if val:
            flat[idx:idx + c] = True
        idx += c
        val = not val
    return flat.reshape(w, h).T


def coco_encode_rle(rle):
    """amg.py:294-300 (frPyObjects on an uncompressed RLE -> compressed string)."""
    return {"size": list(rle["size"]), "counts": coco_rle_string(rle["counts"])}


def remove_small_regions(mask, area_thresh, mode):
    """amg.py:267-291 with scipy 8-connected labelling standing in for cv2."""
    from scipy import ndimage
    correct_holes = mode == "holes"
    working = (correct_holes ^ mask).astype(np.uint8)
    regions, n = ndimage.label(working, structure=np.ones((3, 3), dtype=np.uint8))
    n_labels = n + 1
    sizes = np.bincount(regions.reshape(-1), minlength=n_labels)[1:]
    small = [i + 1 for i, s in enumerate(sizes) if s < area_thresh]
    if len(small) == 0:
        return mask, False
    fill = [0] + small
    if not correct_holes:
        fill = [i for i in range(n_labels) if i not in fill]
        if len(fill) == 0:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


def coverage(mask1, mask2):
    """crowdsam/utils.py:460-469: max(inter/|A|, inter/|B|) over the last two dims (fp32 true division of counts)."""
    inter = (mask1 * mask2).sum([-1, -2])
    return torch.maximum(inter / mask1.sum([-1, -2]), inter / mask2.sum([-1, -2]))


def mask_iou(mask1, mask2):
    """crowdsam/utils.py:471-478."""
    return torch.logical_and(mask1, mask2).sum([-1, -2]) / torch.logical_or(mask1, mask2).sum([-1, -2])


def mask_iou_nms(boxes, scores, mask_preds, threshold):
    """crowdsam/utils.py:422-458: masks nearest-resampled to 150x150, greedy in descending score (stable order here:
    the reference's np.argsort default leaves tie order unspecified), a mask is dropped when its coverage with any
    kept mask exceeds the threshold.  -> kept original indices in descending-score order."""
    if mask_preds.numel() == 0:
        return []
    m = torch.nn.functional.interpolate(mask_preds.float().unsqueeze(0), (150, 150))[0].bool()
    order = np.argsort(-np.asarray(scores), kind="stable").tolist()
    keep = []
    for i in order:
        if keep and torch.any(coverage(m[i].unsqueeze(0), m[keep]) > threshold):
            continue
        keep.append(i)
    return np.array(keep)


def fuse_simmap_scores(masks, iou_preds, sim_map, image_hw):
    """crowdsam/model.py:273-286: per mask the mean of the bilinearly resized prior map over its pixels (0 when
    empty), clamp(+0.5, 0, 1), score = sqrt(iou) * sqrt(cls)."""
    hi = F.interpolate(sim_map.unsqueeze(0).unsqueeze(0), tuple(image_hw), mode="bilinear")[0, 0]
    cls = []
    for m in masks:
        c = hi[m].mean() if m.sum() > 0 else torch.tensor(0.)
        cls.append(torch.clamp(c + 0.5, 0, 1))
    cls = torch.stack(cls) if cls else torch.zeros(0)
    return iou_preds ** 0.5 * cls ** 0.5


def generate_crop_boxes(im_size, n_layers, overlap_ratio):
    """amg.py:200-234."""
    crop_boxes, layer_idxs = [], []
    im_h, im_w = im_size
    short = min(im_h, im_w)
    crop_boxes.append([0, 0, im_w, im_h])
    layer_idxs.append(0)

    def crop_len(orig, n, ov):
        return int(math.ceil((ov * (n - 1) + orig) / n))

    for i in range(n_layers):
        n = 2 ** (i + 1)
        ov = int(overlap_ratio * short * (2 / n))
        cw, ch = crop_len(im_w, n, ov), crop_len(im_h, n, ov)
        xs = [int((cw - ov) * k) for k in range(n)]
        ys = [int((ch - ov) * k) for k in range(n)]
        for x0 in xs:
            for y0 in ys:
                crop_boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
                layer_idxs.append(i + 1)
    return crop_boxes, layer_idxs


# ------------------------------------------------------------------------------------------------
# torchvision.ops.nms (published semantics)
# ------------------------------------------------------------------------------------------------
def nms(boxes, scores, thr):
    """Greedy NMS as torchvision's CPU kernel: stable descending score order, areas
    (x2-x1)*(y2-y1), suppress when inter/(a_i+a_j-inter) > thr.  Returns kept indices (int64)
    in descending-score order.  fp32 arithmetic throughout."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    n = b.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = np.argsort(-s, kind="stable")
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > np.float32(thr)]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


# ------------------------------------------------------------------------------------------------
# crowdsam/utils.py helpers
# ------------------------------------------------------------------------------------------------
def resize_shape(h, w, max_size):
    """crowdsam/utils.py:141-149: r = min(max/w, max/h); (int(r*h), int(r*w)) (trap 9)."""
    r = min(max_size / w, max_size / h)
    return int(r * h), int(r * w), r


def get_preprocess_shape(oldh, oldw, long_side):
    """segment_anything_cs/utils/transforms.py:94-102."""
    scale = long_side * 1.0 / max(oldh, oldw)
    newh, neww = oldh * scale, oldw * scale
    return int(newh + 0.5), int(neww + 0.5)


def apply_coords(coords, original_size, target=1024):
    """transforms.py:33-45 (float64 arithmetic)."""
    old_h, old_w = original_size
    new_h, new_w = get_preprocess_shape(old_h, old_w, target)
    c = deepcopy(coords).astype(float)
    c[..., 0] = c[..., 0] * (new_w / old_w)
    c[..., 1] = c[..., 1] * (new_h / old_h)
    return c


def is_box_near_crop_edge(boxes, crop_box, orig_box, downscale, atol=20.0):
    """crowdsam/utils.py:213-223."""
    cb = torch.as_tensor(crop_box, dtype=torch.float)
    ob = torch.as_tensor(orig_box, dtype=torch.float)
    x0, y0 = crop_box[0], crop_box[1]
    b = (boxes / downscale + torch.tensor([[x0, y0, x0, y0]])).float()
    near_crop = torch.isclose(b, cb[None, :], atol=atol, rtol=0)
    near_img = torch.isclose(b, ob[None, :], atol=atol, rtol=0)
    return torch.any(torch.logical_and(near_crop, ~near_img), dim=1)


# ------------------------------------------------------------------------------------------------
# the driver (crowdsam/model.py)
# ------------------------------------------------------------------------------------------------
DEFAULT_TEST_CFG = dict(  # configs/crowdhuman.yaml:33-58
    mask_selection="max_iou", apply_box_offsets=False, max_prompts=500, filter_thresh=0.7,
    max_size=1024, grid_size=192, pred_iou_thresh=0.1, fuse_simmap=False,
    stability_score_thresh=0.8, stability_score_offset=1, box_nms_thresh=0.65,
    points_per_batch=32, crop_n_layers=0, crop_nms_thresh=0.7, crop_overlap_ratio=0.341,
    min_mask_region_area=100, pos_sim_thresh=0.5, output_rles=True)


class OracleCrowdSAM:
    """CPU restatement of crowdsam/model.py::CrowdSAM for sam_arch == 'crowdsam', trainfree False,
    crop_n_layers == 0.  ``dino_fn(x[1,3,1022,1022]) -> [1,5329,1024]`` supplies the DINOv2 patch
    tokens (real restatement or a seeded stand-in).  ``rng`` stands for the global NumPy RNG the
    reference shuffles with (crowdsam/model.py:231; seeded at tools/test.py:29)."""

    mask_threshold = 0.0  # sam.py:18

    def __init__(self, sam_sd, arch_cfg, dino_fn, test_cfg=None, n_class=1, rng=None, record=None):
        self.sd = sam_sd
        self.depth, self.heads, self.global_idx = arch_cfg
        self.dino_fn = dino_fn
        self.cfg = dict(DEFAULT_TEST_CFG)
        if test_cfg:
            self.cfg.update(test_cfg)
        self.n_class = n_class
        self.rng = rng if rng is not None else np.random
        self.record = record  # optional dict collecting intermediates for tests
        self._dense_pe = so.dense_pe(sam_sd)

    # predictor.set_image / set_torch_image (predictor.py:32-112)
    def set_image(self, image):
        from . import resize_oracle
        h, w = image.shape[:2]
        nh, nw = get_preprocess_shape(h, w, 1024)            # ResizeLongestSide.apply_image (transforms.py:26-31):
        if (nh, nw) != (h, w):                               # PIL bilinear; identity unless the long side is 1023
            image = resize_oracle.pil_resize_bilinear_u8(image, (nh, nw))
        self.original_size = (h, w)
        self.input_size = (nh, nw)
        x = torch.as_tensor(np.ascontiguousarray(image)).permute(2, 0, 1).contiguous()[None].float()
        x = so.preprocess(x[0])[None]
        self.features = so.image_encoder(self.sd, x, self.depth, self.heads, self.global_idx)
        xd = F.interpolate(x, (1022, 1022), mode="bilinear")
        self.dino_feats = self.dino_fn(xd).view(1, 73, 73, -1)

    def predict_torch(self, coords, labels):
        """predictor.py:214-292 with return_logits=True."""
        sparse = so.embed_points(self.sd, coords, labels)
        low, iou, cls = so.mask_decoder(self.sd, self.features, self._dense_pe, sparse, self.dino_feats)
        masks = so.postprocess_masks(low, self.input_size, self.original_size)
        return masks, iou, cls, low

    def sample_points(self):
        """crowdsam/model.py:196-223: FG prior -> grid -> threshold -> pixel coords."""
        cfg = self.cfg
        img_size = torch.tensor(self.image.shape[:2])
        g = cfg["grid_size"]
        feat_size = (img_size * min(g / img_size)).int()
        sim = so.predict_fg_map(self.sd, self.dino_feats)
        sim = F.interpolate(sim, (g, g), mode="bilinear")
        sim = sim.sigmoid().max(dim=1)[0]
        sim = sim[0, :feat_size[0], :feat_size[1]]
        self._sim = sim
        fg = sim > cfg["pos_sim_thresh"]
        coords = fg.nonzero()[:, [1, 0]]
        inv = torch.tensor([feat_size[1] / self.image.shape[1], feat_size[0] / self.image.shape[0]])
        coords = coords / inv
        if self.record is not None:
            self.record["sim_map"] = sim.clone()
        return coords.numpy()

    def process_batch(self, points):
        """crowdsam/model.py:334-390 (single crop == whole image, so the crop-edge filter is a no-op)."""
        cfg = self.cfg
        tp = apply_coords(points, self.original_size)
        in_pts = torch.as_tensor(tp)
        in_lbl = torch.ones(in_pts.shape[0], dtype=torch.int)
        masks, iou, cls, low = self.predict_torch(in_pts[:, None, :], in_lbl[:, None])
        iou = torch.clamp(iou, 0.) * cls.squeeze(2).sigmoid()      # :351 (n_class == 1)
        assert cfg["mask_selection"] == "max_iou"
        ind = iou.max(dim=-1)[1]
        ar = torch.arange(len(masks))
        categories = cls.max(dim=-1)[1][ar, ind]
        sel_masks, sel_iou = masks[ar, ind], iou[ar, ind]
        d = dict(masks=sel_masks, iou_preds=sel_iou, points=torch.as_tensor(points), categories=categories)
        if self.record is not None:
            self.record.setdefault("batches", []).append(dict(
                points=np.array(points), low_res=low.clone(), iou_fused=iou.clone(), sel=ind.clone(),
                iou_raw=None))

        def filt(keep):
            for k in d:
                d[k] = d[k][keep]

        if cfg["pred_iou_thresh"] > 0.0:
            filt(d["iou_preds"] > cfg["pred_iou_thresh"])
        d["stability_score"] = calculate_stability_score(d["masks"], self.mask_threshold,
                                                         cfg["stability_score_offset"])
        if cfg["stability_score_thresh"] > 0.0:
            filt(d["stability_score"] >= cfg["stability_score_thresh"])
        d["masks"] = d["masks"] > self.mask_threshold
        d["boxes"] = batched_mask_to_box(d["masks"])
        orig_h, orig_w = self.orig_image.shape[:2]
        keep = ~is_box_near_crop_edge(d["boxes"], self.crop_box, [0, 0, orig_w, orig_h], self.downscale)
        if not torch.all(keep):
            filt(keep)
        return d

    def generate(self, image):
        """crowdsam/model.py:133-190: crops -> _process_crop -> cross-crop NMS (smaller crops first) -> numpy dict.
        With more than one crop the reference index-filters its 2-entries-per-crop ``rles_info`` list and raises
        (SURVEY.md section 8 f4); the sane behaviour restated here keeps per-mask crop boxes (``rles_crop``) and
        filters those."""
        cfg = self.cfg
        image = np.asarray(image, dtype=np.uint8)
        crop_boxes, _ = generate_crop_boxes(image.shape[:2], cfg["crop_n_layers"], cfg["crop_overlap_ratio"])
        data = None
        self.n_batches = 0
        for crop_box in crop_boxes:
            cd = self._process_crop(image, crop_box)
            if cd is None:
                continue
            if data is None:
                data = cd
            else:
                for k in data:
                    data[k] = data[k] + cd[k] if isinstance(data[k], list) else torch.cat([data[k], cd[k]], dim=0)
        empty = dict(boxes=np.zeros((0, 4), np.float32), scores=np.zeros((0,), np.float32),
                     categories=np.zeros((0,), np.int64), rles=[], points=np.zeros((0, 2)))
        if data is None:
            return empty
        if len(crop_boxes) > 1 and len(data["crop_boxes"]) > 0:
            cb = data["crop_boxes"]
            scores = 1 / ((cb[:, 2] - cb[:, 0]) * (cb[:, 3] - cb[:, 1]))          # :169 prefer smaller crops
            keep = nms(data["boxes"].float(), scores, cfg["crop_nms_thresh"])
            for k in data:
                data[k] = [data[k][int(i)] for i in keep] if isinstance(data[k], list) else data[k][keep]
        out = dict(boxes=data["boxes"].numpy(), scores=data["scores"].numpy(), categories=data["categories"].numpy(),
                   points=data["points"].numpy(), stability_score=data["stability_score"].numpy(),
                   rles=[coco_encode_rle(r) for r in data["rles"]], rles_uncompressed=data["rles"],
                   rles_crop=data["crop_boxes"].numpy())
        if "masks" in data:
            out["masks"] = data["masks"]
        return out

    def _process_crop(self, image, crop_box):
        """crowdsam/model.py:192-306 for one crop."""
        from . import resize_oracle
        cfg = self.cfg
        self.orig_image = image
        self.crop_box = list(crop_box)
        x0, y0, x1, y1 = crop_box
        self.image, self.downscale = resize_oracle.resize_image(image[y0:y1, x0:x1, :], cfg["max_size"])   # :119-131
        self.set_image(self.image)
        pts = self.sample_points()
        occupy = torch.zeros(self.image.shape[0], self.image.shape[1], dtype=torch.bool)
        data = None
        points = pts.astype("int")           # :230 truncation
        self.rng.shuffle(points)             # :231 global RNG
        count = 0
        bs = cfg["points_per_batch"]
        while len(points) > 0 and count < cfg["max_prompts"]:
            bs = min(len(points), bs)
            sel = points[:bs]
            points = points[bs:]
            bd = self.process_batch(sel)
            occupy = (bd["masks"][bd["iou_preds"] > cfg["filter_thresh"]]).any(0)   # :246 (replaced)
            if data is None:
                data = {k: v.clone() for k, v in bd.items()}
            else:
                for k in data:
                    data[k] = torch.cat([data[k], bd[k]], dim=0)
            keep = (~occupy[points[:, 1], points[:, 0]]).numpy()
            points = points[keep]
            count += bs
            self.n_batches += 1
        if data is None or len(data["masks"]) == 0:
            return None
        keep = nms(data["boxes"].float(), data["iou_preds"], cfg["box_nms_thresh"])
        for k in data:
            data[k] = data[k][keep]
        if cfg["min_mask_region_area"] > 0:
            data = self.postprocess_small_regions(data, cfg["min_mask_region_area"],
                                                  max(cfg["box_nms_thresh"], cfg["crop_nms_thresh"]))
        if cfg["fuse_simmap"]:
            data["scores"] = fuse_simmap_scores(data["masks"], data["iou_preds"], self._sim, self.image.shape[:2])
        else:
            data["scores"] = data["iou_preds"]
        n = len(data["boxes"])
        out = dict(boxes=data["boxes"] / self.downscale + torch.tensor([[x0, y0, x0, y0]]),         # :298-300
                   points=data["points"] / self.downscale + torch.tensor([[x0, y0]]),
                   scores=data["scores"], categories=data["categories"], stability_score=data["stability_score"],
                   crop_boxes=torch.tensor([list(crop_box) for _ in range(n)]).reshape(n, 4),
                   rles=mask_to_rle(data["masks"]))
        out["masks"] = [m for m in data["masks"].numpy()]      # per-crop frames differ in size: a list
        return out

    @staticmethod
    def postprocess_small_regions(data, min_area, nms_thresh):
        """crowdsam/model.py:394-443."""
        if len(data["masks"]) == 0:
            return data
        new_masks, scores = [], []
        for mask in data["masks"].numpy():
            mask, changed = remove_small_regions(mask, min_area, "holes")
            unchanged = not changed
            mask, changed = remove_small_regions(mask, min_area, "islands")
            unchanged = unchanged and not changed
            new_masks.append(torch.as_tensor(mask).unsqueeze(0))
            scores.append(float(unchanged))
        masks = torch.cat(new_masks, dim=0)
        boxes = batched_mask_to_box(masks)
        keep = nms(boxes.float(), torch.as_tensor(scores), nms_thresh)
        for i in keep:
            if scores[i] == 0.0:
                data["boxes"][i] = boxes[i]
                data["masks"][i] = masks[i]
        for k in data:
            data[k] = data[k][keep]
        return data
