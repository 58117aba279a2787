"""CrowdHuman evaluator (AP / log-average miss rate / recall) with the Caltech matching on device.

Reference: tools/crowdhuman_eval.py -- Database.loadData_odgt / loadData (:395-434), Image.load_gt_boxes
(:262-297), Image.load_cocojson (:27-67), Image.clip_all_boader (:238-260), Image.compare_caltech (:113-143),
Database.compare / eval_MR / eval_AP (:436-548), _evaluate_predictions_on_crowdhuman (:550-558), for the
combination tools/batch_eval.py:100 uses (body boxes, GT from .odgt or COCO json, detections from COCO json).

Split of work: parsing and the two cumulative curves are host numpy (they are O(#detections)); the per-image
IoU / IoA matrices and the greedy matching -- the O(N*K) part, a Python double loop over a materialised matrix in
the reference -- run in csam_caltech_match, one wave per image, float64, bit-identical labels.  The same entry
takes the RCCL-gathered detection rows of tools/batch_eval.py directly (evaluate_rows), so a multi-GPU run never
writes the temp_result_*.json / test.json files of the reference.
"""
import json

import numpy as np
import torch

from crowdsam_amd import hip

PERSON_CLASSES = ["background", "person"]
MR_REF = {"CALTECH_-2": [0.0100, 0.0178, 0.03160, 0.0562, 0.1000, 0.1778, 0.3162, 0.5623, 1.000],
          "CALTECH_-4": [0.0001, 0.0003, 0.00100, 0.0032, 0.0100, 0.0316, 0.1000, 0.3162, 1.000]}


class ImageRecord:
    """Per-image boxes: gt [K,5] = x0,y0,x1,y1,tag (1 person / -1 ignore), dt [N,5] = x0,y0,x1,y1,score or None."""

    def __init__(self, image_id, width=None, height=None):
        self.ID, self.width, self.height = image_id, width, height
        self.gt = np.zeros((0, 5))
        self.dt = None
        self.gt_num = 0
        self.ign_num = 0

    def clip(self):
        """clip_all_boader: x0,y0 into [0,w-1]/[0,h-1], x1,y1 into [0,w]/[0,h] -- detections and GT alike."""
        for b in (self.dt, self.gt):
            b[:, 0] = np.minimum(np.maximum(b[:, 0], 0), self.width - 1)
            b[:, 1] = np.minimum(np.maximum(b[:, 1], 0), self.height - 1)
            b[:, 2] = np.maximum(np.minimum(b[:, 2], self.width), 0)
            b[:, 3] = np.maximum(np.minimum(b[:, 3], self.height), 0)


def _gt_from_odgt(record, visible_flag):
    rows = []
    for rb in record["gtboxes"]:
        tag = PERSON_CLASSES.index(rb["tag"]) if rb["tag"] in PERSON_CLASSES else -1
        if rb.get("extra", {}).get("ignore", 0) != 0:
            tag = -1
        if visible_flag:
            box = rb["vbox"][0] if isinstance(rb["vbox"][0], list) else rb["vbox"]
        else:
            box = rb["fbox"]
        rows.append([box[0], box[1], box[0] + box[2], box[1] + box[3], tag])
    return np.array(rows, dtype=np.float64).reshape(-1, 5)


def _xywh_rows(annots, last):
    if not annots:
        return np.zeros((0, 5))
    b = np.array([a["bbox"] for a in annots], dtype=np.float64)
    b[:, 2:4] += b[:, :2]
    return np.concatenate([b, np.asarray(last, dtype=np.float64).reshape(-1, 1)], axis=1)


def _split_by_image(coco):
    """Annotations grouped by image the way the reference walks them: contiguous runs in file order."""
    annots = coco["annotations"]
    a = 0
    for item in coco["images"]:
        k = 0
        while a + k < len(annots) and annots[a + k]["image_id"] == item["id"]:
            k += 1
        yield item, annots[a:a + k]
        a += k


def load_gt(gt_path, visible_flag=False):
    """-> {image id: ImageRecord} in file order (.odgt: one json record per line or one line holding the list;
    .json: COCO annotations with an optional `ignore` field)."""
    images = {}
    if ".json" in gt_path:
        coco = json.load(open(gt_path))
        ids = [im["id"] for im in coco["images"]]
        assert len(ids) == len(set(ids)), "duplicate image ids"
        for item, annots in _split_by_image(coco):
            rec = ImageRecord(item["id"], item["width"], item["height"])
            tags = [(-1 if a.get("ignore", 0) == 1 else 1) for a in annots] if annots and "ignore" in annots[0] \
                else np.ones(len(annots))
            rec.gt = _xywh_rows(annots, tags)
            rec.gt_num = len(annots)
            rec.ign_num = int((rec.gt[:, -1] == -1).sum())
            images[item["id"]] = rec
    elif ".odgt" in gt_path:
        with open(gt_path) as f:
            lines = f.readlines()
        records = json.loads(lines[0]) if len(lines) == 1 else [json.loads(l) for l in lines]
        for r in records:
            rec = ImageRecord(r["ID"], r.get("width"), r.get("height"))
            rec.gt = _gt_from_odgt(r, visible_flag)
            rec.gt_num = len(r["gtboxes"])
            rec.ign_num = int((rec.gt[:, -1] == -1).sum())
            images[r["ID"]] = rec
    else:
        raise NotImplementedError(gt_path)
    return images


def attach_coco_detections(images, dt_path_or_dict):
    coco = dt_path_or_dict if isinstance(dt_path_or_dict, dict) else json.load(open(dt_path_or_dict))
    for item, annots in _split_by_image(coco):
        rec = images[item["id"]]
        if rec.width is None:
            rec.width = item["width"]
        if rec.height is None:
            rec.height = item["height"]
        scores = [a["score"] for a in annots] if annots and "score" in annots[0] else np.ones(len(annots))
        rec.dt = _xywh_rows(annots, scores)
        rec.clip()


def match(records, thres=0.5, device=None):
    """Caltech matching of every image on device -> (scores, labels, pos) of the kept detections, globally sorted
    by descending score (stable over the image order, as Database.compare's list sort)."""
    device = torch.device(device or "cuda")
    dts, gts, doff, goff, npos = [], [], [0], [0], []
    for r in records:
        if r.dt is None or len(r.dt) == 0 or len(r.gt) == 0:
            d, g = np.zeros((0, 5)), np.zeros((0, 5))
        else:
            d = r.dt[np.argsort(-r.dt[:, -1], kind="stable")]
            g = r.gt[np.argsort(-r.gt[:, -1], kind="stable")]
        dts.append(d)
        gts.append(g)
        doff.append(doff[-1] + len(d))
        goff.append(goff[-1] + len(g))
        npos.append(int((g[:, -1] > 0).sum()))
    dt = np.concatenate(dts) if dts else np.zeros((0, 5))
    gt = np.concatenate(gts) if gts else np.zeros((0, 5))
    if len(dt) == 0:
        return np.zeros(0), np.zeros(0, np.int8), np.zeros(0, bool)
    label, pos = hip.caltech_match(torch.from_numpy(dt).to(device), torch.tensor(doff, dtype=torch.int64, device=device),
                                   torch.from_numpy(gt).to(device), torch.tensor(goff, dtype=torch.int64, device=device),
                                   torch.tensor(npos, dtype=torch.int32, device=device), thres)
    label = label.cpu().numpy()
    pos = pos.cpu().numpy().astype(bool)
    keep = label >= 0
    scores, label, pos = dt[keep, -1], label[keep], pos[keep]
    order = np.argsort(-scores, kind="stable")
    return scores[order], label[order], pos[order]


def curves(labels, total_gt, n_images):
    """Cumulative recall / precision / fppi / miss-rate exactly as eval_AP's loop computes them."""
    tp = np.cumsum(labels == 1).astype(np.float64)
    fp = np.cumsum(labels == 0).astype(np.float64)
    recall = tp / (tp + (total_gt - tp))
    precision = tp / (tp + fp)
    return recall, precision, fp / n_images, 1 - recall


def summarize(scores, labels, gt_num, ign_num, n_images, ref="CALTECH_-2"):
    """-> dict(AP, MR, recall, tp, fp, + curves): eval_AP (:484-548) then eval_MR on its fppi / miss-rate curves."""
    if len(labels) == 0:
        return {"AP": 0.0, "MR": 1.0, "recall": 0.0, "tp": 0, "fp": 0, "recall_curve": np.zeros(0),
                "precision_curve": np.zeros(0), "fppi": np.zeros(0), "labels": labels, "scores": scores}
    recall, precision, fppi, mr = curves(labels, gt_num - ign_num, n_images)
    # trapezoid area accumulated left to right like the reference's `area += ...` loop (np.sum would pair-sum)
    terms = (recall[1:] - recall[:-1]) * ((precision[:-1] + precision[1:]) / 2)
    ap = float(np.add.accumulate(terms)[-1]) if len(terms) else 0.0
    idx = np.searchsorted(fppi, MR_REF[ref], side="left")
    idx[idx >= len(fppi)] = len(fppi) - 1
    mmr = float(np.exp(np.log(mr[idx]).mean()))
    return {"AP": ap, "MR": mmr, "recall": float(recall[-1]), "tp": int((labels == 1).sum()),
            "fp": int((labels == 0).sum()), "recall_curve": recall, "precision_curve": precision, "fppi": fppi,
            "labels": labels, "scores": scores}


def evaluate(gt_path, dt_path, remove_empty_gt=False, visible_flag=False, thres=0.5, device=None):
    """_evaluate_predictions_on_crowdhuman: GT file + COCO detection file (or dict) -> summary dict."""
    images = load_gt(gt_path, visible_flag)
    attach_coco_detections(images, dt_path)
    recs = list(images.values())
    if remove_empty_gt:
        recs = [r for r in recs if r.dt is not None]
    scores, labels, pos = match(recs, thres, device)
    return summarize(scores, labels, sum(r.gt_num for r in recs), sum(r.ign_num for r in recs), len(recs))


def rows_to_coco(rows, gt_images):
    """Gathered detection rows [n,6] = (image_index, x0,y0,x1,y1, score) -> the COCO dict tools/batch_eval.py:31-58
    builds (images keyed by file_name[:-4], xyxy -> xywh, running annotation ids)."""
    images = [dict(im, id=im["file_name"][:-4]) for im in gt_images]
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 6)
    order = np.argsort(rows[:, 0], kind="stable")
    annots = []
    for k, r in enumerate(rows[order]):
        x0, y0, x1, y1 = (float(v) for v in r[1:5])
        annots.append({"category_id": 1, "bbox": [x0, y0, x1 - x0, y1 - y0], "image_id": images[int(r[0])]["id"],
                       "iscrowd": False, "area": (y1 - y0) * (x1 - x0), "id": k, "score": float(r[5])})
    return {"images": images, "annotations": annots}


def evaluate_rows(rows, gt_images, gt_path, remove_empty_gt=True, visible_flag=True, thres=0.5, device=None):
    """Evaluate the all-gathered rows of a sharded run without touching the file system."""
    coco = rows_to_coco(rows, gt_images)
    seen = set(int(i) for i in np.asarray(rows).reshape(-1, 6)[:, 0])
    coco["images"] = [im for k, im in enumerate(coco["images"]) if k in seen] if remove_empty_gt else coco["images"]
    return evaluate(gt_path, coco, remove_empty_gt, visible_flag, thres, device)
