"""Hot-path helpers + config / IO utilities of the reference's crowdsam/utils.py that the drivers
import (tools/test.py:9-11, tools/batch_eval.py:7, crowdsam/model.py:12).

Visualisation, losses and dead code of the reference file are out of scope.
"""
import functools
import json
import logging
import os
import sys

import numpy as np
import torch
import yaml
from PIL import Image

# dataset table (crowdsam/utils.py:26-30): name -> (image dir, annotation json)
data_meta = {
    "crowdhuman": ("Images", "midval_visible.json"),
    "coco_occ": ("val2017", "occ_coco.json"),
    "occ_human": ("images", "occhuman_coco_format.json"),
}


# hot-path knobs with the shipped values (reference configs/crowdhuman.yaml:33-58)
DEFAULT_TEST_CONFIG = dict(
    mask_selection="max_iou", apply_box_offsets=False, max_prompts=500, filter_thresh=0.7, max_size=1024,
    grid_size=192, pred_iou_thresh=0.1, fuse_simmap=False, stability_score_thresh=0.8, stability_score_offset=1,
    box_nms_thresh=0.65, points_per_batch=32, crop_n_layers=0, crop_nms_thresh=0.7, crop_overlap_ratio=0.341,
    min_mask_region_area=100, pos_sim_thresh=0.5, output_rles=True)


# ---- config (crowdsam/utils.py:31-58): YAML -> nested dict, trailing "a.b.c value" overrides
def load_config(path):
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def convert_value(value):
    if value in ("True", "true"):
        return True
    if value in ("False", "false"):
        return False
    for cast in (int, float):
        try:
            return cast(value)
        except ValueError:
            pass
    return value


def modify_config(config, options):
    """options = [key1, value1, key2, value2, ...] with dotted keys."""
    for key, value in zip(options[0::2], options[1::2]):
        node = config
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = convert_value(value)
    return config


# ---- geometry (crowdsam/utils.py:141-156, 175-190, 213-223)
def resize_shape(h, w, max_size):
    r = min(max_size / w, max_size / h)
    return int(r * h), int(r * w), r


def resize_image(image, max_size):
    """Scale so the long side becomes max_size (up- or down-scaling); returns (image, r).
    The reference calls cv2.resize (third-party fixed-point bilinear, parity unpinned); PIL bilinear
    here.  Identity (copy) when the size does not change."""
    h, w = image.shape[:2]
    nh, nw, r = resize_shape(h, w, max_size)
    if isinstance(image, np.ndarray):
        if (nh, nw) == (h, w):
            return image.copy(), r
        return np.array(Image.fromarray(image).resize((nw, nh), Image.BILINEAR)), r
    raise TypeError("resize_image expects a numpy image on the inference path")


def uncrop_boxes_xyxy(boxes, crop_box, downscale):
    x0, y0 = crop_box[0], crop_box[1]
    offset = torch.tensor([[x0, y0, x0, y0]], device=boxes.device)
    if boxes.dim() == 3:
        offset = offset.unsqueeze(1)
    return boxes / downscale + offset


def uncrop_points(points, crop_box, downscale):
    x0, y0 = crop_box[0], crop_box[1]
    offset = torch.tensor([[x0, y0]], device=points.device)
    if points.dim() == 3:
        offset = offset.unsqueeze(1)
    return points / downscale + offset


def is_box_near_crop_edge(boxes, crop_box, orig_box, downscale, atol=20.0):
    """True for boxes touching a crop edge that is not an image edge."""
    crop = torch.as_tensor(crop_box, dtype=torch.float, device=boxes.device)[None, :]
    orig = torch.as_tensor(orig_box, dtype=torch.float, device=boxes.device)[None, :]
    b = uncrop_boxes_xyxy(boxes, crop_box, downscale).float()
    near_crop = torch.isclose(b, crop, atol=atol, rtol=0)
    near_img = torch.isclose(b, orig, atol=atol, rtol=0)
    return torch.any(near_crop & ~near_img, dim=1)


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def apply_box_offsets(boxes, box_delta):
    xy = boxes[:, :2] + box_delta[:, :2] * boxes[:, 2:]
    wh = boxes[:, 2:] * torch.exp(box_delta[:, 2:])
    return box_cxcywh_to_xyxy(torch.cat([xy, wh], dim=-1))


# ---- IO (crowdsam/utils.py:370-390, 164-172)
def load_img_and_annotation(dataset_root, dataset, id_, coco):
    """Returns (RGB uint8 image, file name, gt boxes xyxy float array) for a COCO-style index."""
    info = coco.loadImgs(id_)[0] if hasattr(coco, "loadImgs") else coco["images_by_id"][id_]
    file_name = info["file_name"]
    path = os.path.join(dataset_root, data_meta[dataset][0], file_name)
    image = np.array(Image.open(path).convert("RGB"))
    if hasattr(coco, "loadAnns"):
        anns = coco.loadAnns(coco.getAnnIds(imgIds=id_))
    else:
        anns = coco["anns_by_image"].get(id_, [])
    boxes = np.array([a["bbox"] for a in anns], dtype=np.float64).reshape(-1, 4)
    if len(boxes):
        boxes[:, 2:] += boxes[:, :2]
    return image, file_name, boxes


class _JsonCoco(dict):
    """Minimal COCO index (pycocotools is not a dependency of this build)."""

    def __init__(self, json_file):
        with open(json_file) as f:
            data = json.load(f)
        by_img = {}
        for a in data.get("annotations", []):
            by_img.setdefault(a["image_id"], []).append(a)
        super().__init__(images_by_id={i["id"]: i for i in data["images"]}, anns_by_image=by_img)
        self.image_ids = [i["id"] for i in data["images"]]

    def getImgIds(self):
        return list(self.image_ids)


def load_coco_index(json_file):
    return _JsonCoco(json_file)


@functools.lru_cache()
def setup_logger(save_path, quiet=False):
    """Plain ``logging`` (the reference's loguru sinks filter everything out, SURVEY.md trap 10)."""
    logger = logging.getLogger("crowdsam")
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        os.makedirs(save_path, exist_ok=True)
        fh = logging.FileHandler(os.path.join(save_path, "run.log"))
        fh.setLevel(logging.DEBUG)
        logger.addHandler(fh)
        if not quiet:
            sh = logging.StreamHandler(sys.stdout)
            sh.setLevel(logging.INFO)
            logger.addHandler(sh)
    return logger


# ---- evaluation helper used for per-image FP/FN bookkeeping (crowdsam/utils.py:482-524)
def box_iou_np(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1, 4), np.asarray(b, np.float64).reshape(-1, 4)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter + 1e-12)


def evaluate_boxes(pred_boxes, pred_scores, gt_boxes, score_thresh=0.5, iou_thresh=0.5):
    """Greedy score-ordered matching; returns (tp_flags per kept prediction, matched gt flags)."""
    pred_boxes, pred_scores = np.asarray(pred_boxes).reshape(-1, 4), np.asarray(pred_scores).reshape(-1)
    sel = pred_scores > score_thresh
    pb, ps = pred_boxes[sel], pred_scores[sel]
    order = np.argsort(-ps, kind="stable")
    gt_used = np.zeros(len(gt_boxes), dtype=bool)
    tp = np.zeros(len(pb), dtype=bool)
    if len(gt_boxes) and len(pb):
        iou = box_iou_np(pb, gt_boxes)
        for i in order:
            cand = np.where(~gt_used, iou[i], -1.0)
            j = int(np.argmax(cand))
            if cand[j] >= iou_thresh:
                gt_used[j] = True
                tp[i] = True
    return tp, gt_used


def coverage(mask1, mask2):
    """max(inter/|A|, inter/|B|) over the last two dims (reference: crowdsam/utils.py:460-469)."""
    inter = (mask1 * mask2).sum([-1, -2])
    return torch.maximum(inter / mask1.sum([-1, -2]), inter / mask2.sum([-1, -2]))


def mask_iou(mask1, mask2):
    """Mask IoU over the last two dims (reference: crowdsam/utils.py:471-478)."""
    return torch.logical_and(mask1, mask2).sum([-1, -2]) / torch.logical_or(mask1, mask2).sum([-1, -2])


def mask_iou_nms(boxes, scores, mask_preds, threshold):
    """Coverage NMS on 150x150 nearest-resampled masks (reference: crowdsam/utils.py:422-458; `boxes` is unused
    there as well).  Runs in csam_mask_nms; returns the kept original indices in descending-score order."""
    if mask_preds.numel() == 0:
        return []
    from crowdsam_amd import hip
    dev = mask_preds.device if mask_preds.is_cuda else torch.device("cuda")
    keep = hip.mask_nms(mask_preds.to(dev).bool(), torch.as_tensor(np.asarray(scores), dtype=torch.float32, device=dev),
                        threshold)
    return keep.cpu().numpy()


def visualize_result(*a, **k):
    raise NotImplementedError("visualisation is out of scope of the MI355X hot-path build (SURVEY.md §2 #12)")
