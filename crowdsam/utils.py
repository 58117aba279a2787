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

from .coco_names import coco_classes

# data_meta = [dataset_path, n_class, categories]  (crowdsam/utils.py:25-30 of the reference; tools/test.py:40 reads [1:])
data_meta = {"crowdhuman": ["./datasets/crowdhuman", 1, {1: "person"}],
             "occhuman": ["./datasets/OCHuman", 1, {1: "person"}],
             "coco_occ": ["./datasets/coco", 80, coco_classes],
             "coco": ["./datasets/occ_coco", 80, coco_classes]}
# image sub-directory per dataset (the if/elif chain of load_img_and_annotation, crowdsam/utils.py:372-383)
_IMAGE_DIRS = {"crowdhuman": "Images", "coco": "val2017", "coco_occ": "occ2017", "occhuman": "images",
               "mineapple": "images"}


# hot-path knobs with the shipped values (reference configs/crowdhuman.yaml:33-58)
DEFAULT_TEST_CONFIG = dict(
    mask_selection="max_iou", apply_box_offsets=False, max_prompts=500, filter_thresh=0.7, max_size=1024,
    grid_size=192, pred_iou_thresh=0.1, fuse_simmap=False, stability_score_thresh=0.8, stability_score_offset=1,
    box_nms_thresh=0.65, points_per_batch=32, crop_n_layers=0, crop_nms_thresh=0.7, crop_overlap_ratio=0.341,
    min_mask_region_area=100, pos_sim_thresh=0.5, output_rles=True)


# ---- config (crowdsam/utils.py:31-58): YAML -> nested dict, trailing "a.b.c value" overrides
def load_config(config_file):
    with open(config_file, "r") as file:
        return yaml.safe_load(file)


def convert_value(value):
    """'true'/'false' (any case) -> bool, else int, else float, else the string (crowdsam/utils.py:37-47)."""
    if value.lower() in {"true", "false"}:
        return value.lower() == "true"
    for cast in (int, float):
        try:
            return cast(value)
        except ValueError:
            pass
    return value


def modify_config(config_file, options):
    """options = [key1, value1, key2, value2, ...] with dotted keys; missing intermediate sections are created
    (``setdefault``), as in the reference (crowdsam/utils.py:48-58)."""
    assert len(options) % 2 == 0
    for key, value in zip(options[0::2], options[1::2]):
        parts = key.split(".")
        node = config_file
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = convert_value(value)
    return config_file


def coco_decode_rle(encoded_rle):
    """Compressed COCO RLE {'size': [h, w], 'counts': str} -> uint8 [h, w] mask (crowdsam/utils.py:59-70, which calls
    pycocotools.mask.decode; the string format is public: 5 data bits + continuation bit per char offset by 48,
    sign-extended, runs after the third delta-coded against counts[i-2]; column-major fill)."""
    h, w = encoded_rle["size"]
    s = encoded_rle["counts"]
    s = s.decode("ascii") if isinstance(s, bytes) else s
    counts, p = [], 0
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
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    flat = np.zeros(h * w, dtype=np.uint8)
    idx, val = 0, 0
    for c in counts:
        if val:
            flat[idx:idx + c] = 1
        idx += c
        val ^= 1
    return flat.reshape(w, h).T


# ---- geometry (crowdsam/utils.py:141-156, 175-190, 213-223)
def resize_shape(h, w, max_size):
    r = min(max_size / w, max_size / h)
    return int(r * h), int(r * w), r


def resize_frame_device(image, max_size, device):
    """The reference's ``resize_image`` for an ndarray frame, on the GPU: ONE H2D of the uint8 crop, then
    csam_resize_linear_u8 (cv2.resize INTER_LINEAR restated: crowdsam_amd/resize.py).  Returns
    (uint8 [nh,nw,3] device tensor, fp32 [3,nh,nw] device tensor | None, r).  Same size -> the uploaded frame itself
    (cv2.resize returns a copy there)."""
    from crowdsam_amd import hip
    from crowdsam_amd.resize import cv2_linear_tables_device
    h, w = image.shape[:2]
    nh, nw, r = resize_shape(h, w, max_size)
    # (a pinned staging buffer for this upload was measured and rejected: with a torch pin_memory allocation alive
    # every kernel of the frame ran 1.6-1.75x slower on the box -- 43.0 vs 64.7-75.3 ms/image, A/B in one process pair,
    # profiles/r02_pinned_upload_ab.txt; the pageable copy stays)
    dev = torch.from_numpy(np.ascontiguousarray(image)).to(device, non_blocking=True)
    if (nh, nw) == (h, w):
        return dev, None, r
    u8, f32 = hip.resize_linear_u8(dev, cv2_linear_tables_device(h, w, nh, nw, str(device)), (nh, nw))
    return u8, f32, r


def resize_image(image, max_size):
    """Scale so the long side becomes max_size (up- or down-scaling); returns (image, r)  (crowdsam/utils.py:141-156).
    ndarray uint8 HWC -> the device restatement of cv2.resize (ndarray back); torch tensors -> F.interpolate
    (nearest), as the reference."""
    h, w = image.shape[:2]
    nh, nw, r = resize_shape(h, w, max_size)
    if isinstance(image, np.ndarray):
        if image.ndim == 2 or image.shape[2] != 3 or image.dtype != np.uint8:
            raise TypeError("resize_image: the HIP path resizes uint8 HWC frames with 3 channels")
        if (nh, nw) == (h, w):
            return image.copy(), r           # cv2.resize to the same size is a copy
        dev = torch.device("cuda", torch.cuda.current_device())
        return resize_frame_device(image, max_size, dev)[0].cpu().numpy(), r
    if isinstance(image, torch.Tensor):
        assert image.ndim in (2, 3)
        F = torch.nn.functional
        if image.ndim == 2:
            return F.interpolate(image[None, None], (nh, nw))[0, 0], r
        return F.interpolate(image.permute(2, 0, 1)[None], (nh, nw))[0].permute(1, 2, 0), r
    return image, r


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
def load_img_and_annotation(dataset_path, annots, dataset, id=0):
    """(RGB uint8 image, gt boxes xyxy ndarray, image id) of ``annots['images'][id]`` (crowdsam/utils.py:370-390).
    ``annots`` is the loaded COCO-style json dict.  The reference decodes with cv2.imread + BGR->RGB; Pillow here
    (both sit on libjpeg; decoder rounding is third-party and unpinned)."""
    img_meta = annots["images"][id]
    if dataset not in _IMAGE_DIRS:
        raise NotImplementedError
    file_name = img_meta["file_name"].split("/")[-1] if dataset == "coco_occ" else img_meta["file_name"]
    img_path = os.path.join(dataset_path, _IMAGE_DIRS[dataset], file_name)
    image = np.array(Image.open(img_path).convert("RGB"))
    bboxes = np.array([a["bbox"] for a in annots["annotations"] if a["image_id"] == img_meta["id"]])
    bboxes[..., 2:] += bboxes[..., :2]
    return image, bboxes, img_meta["id"]


class _JsonCoco(dict):
    """Minimal COCO index for the build's own harness (pycocotools is not a dependency of this build)."""

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
    """Plain ``logging`` with loguru's call surface used by the tools (.info / .debug / .warning / .error); the
    reference's loguru sinks filter everything out (SURVEY.md trap 10), this one logs to ``save_path/run.log``."""
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


# ---- per-image FP / FN bookkeeping for the visualiser (crowdsam/utils.py:482-524)
def box_iou_np(a, b):
    """torchvision.ops.box_iou semantics (areas (x2-x1)(y2-y1), inter clamped at 0) in the input dtype."""
    a, b = np.asarray(a).reshape(-1, 4), np.asarray(b).reshape(-1, 4)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area_a[:, None] + area_b[None, :] - inter)


def evaluate_boxes(pred_boxes, pred_scores, gt_boxes, iou_thresh):
    """-> (precision, recall, FP_list, FN_list)  (crowdsam/utils.py:482-524).  Predictions in descending score order;
    each takes the FIRST (lowest index) still-unmatched gt whose IoU exceeds the threshold; ``precision`` is the sum
    of running precisions at the true positives over the number of gt boxes; FP_list holds original prediction
    indices, FN_list the unmatched gt indices.  Empty prediction set -> (0, 0, [], [])."""
    assert len(pred_scores) >= 0
    assert len(pred_boxes) == len(pred_scores)
    assert len(gt_boxes) >= 0
    if len(pred_boxes) == 0:
        return 0, 0, [], []
    pred_boxes, pred_scores = np.asarray(pred_boxes), np.asarray(pred_scores)
    gt_boxes = np.asarray(gt_boxes).reshape(-1, 4)
    ind = torch.as_tensor(pred_scores).sort(descending=True)[1].numpy()      # torch.sort's own tie order
    iou = box_iou_np(pred_boxes[ind], gt_boxes)
    match = np.zeros(len(gt_boxes), dtype=bool)
    prec, TP, FP, FP_list = [], 0, 0, []
    for i in range(iou.shape[0]):
        cand = np.nonzero((iou[i] > iou_thresh) & ~match)[0]
        if len(cand):
            match[cand[0]] = True
            TP += 1
            prec.append(TP / (TP + FP))
        else:
            FP += 1
            FP_list.append(int(ind[i]))
    if len(gt_boxes) > 0:
        precision = sum(prec) / len(gt_boxes) if prec else 0
        recall = TP / len(gt_boxes)
    else:
        precision = recall = 0
    return precision, recall, FP_list, np.nonzero(~match)[0].tolist()


def visualize_result(image, result, class_names, save_path, vis_masks=True, conf_thresh=0.001, FP_ind=None,
                     FN_ind=None):
    """Draw boxes ("class:score"; false positives red, detections yellow, missed gt boxes blue) and, with
    ``vis_masks``, the decoded RLE masks onto the frame and save it (crowdsam/utils.py:71-102; Pillow instead of
    cv2 drawing -- visualisation is not on the measured path)."""
    from PIL import ImageDraw
    canvas = np.array(image).astype(np.float32)
    n = len(result["boxes"])
    keep = [i for i in range(n) if round(float(result["scores"][i]), 3) >= conf_thresh]
    rles = dict(result.items()).get("rles", [])
    if vis_masks and len(rles) and "rles_info" in dict(result.items()):
        rs = np.random.RandomState(0)
        (x0, y0, x1, y1), (oh, ow) = result["rles_info"]
        for i in keep:
            m = coco_decode_rle(rles[i]).astype(bool)
            mh, mw = min(m.shape[0], oh - y0), min(m.shape[1], ow - x0)
            if m.shape != (y1 - y0, x1 - x0):          # masks live in the resized crop frame
                m = np.array(Image.fromarray(m.astype(np.uint8)).resize((x1 - x0, y1 - y0), Image.NEAREST)).astype(bool)
                mh, mw = m.shape
            sel = np.zeros(canvas.shape[:2], bool)
            sel[y0:y0 + mh, x0:x0 + mw] = m[:mh, :mw]
            canvas[sel] = np.minimum(canvas[sel] + 0.5 * rs.random(3) * 255, 255)
    pil = Image.fromarray(canvas.astype(np.uint8))
    draw = ImageDraw.Draw(pil)
    for i in keep:
        box = [float(v) for v in result["boxes"][i]]
        color = (255, 0, 0) if FP_ind is not None and i in FP_ind else (255, 255, 0)
        name = class_names[int(result["categories"][i]) + 1]
        draw.rectangle(box, outline=color)
        draw.text((box[0], max(box[1] - 10, 0)), f"{name}:{round(float(result['scores'][i]), 3)}", fill=color)
    if FN_ind is not None:
        for i in FN_ind:
            draw.rectangle([float(v) for v in result["gt_boxes"][i]], outline=(0, 0, 255))
    pil.save(save_path)


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
