"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float64) of the CrowdHuman evaluator the reference ships
(tools/crowdhuman_eval.py), used only by tests/ as the checker for crowdsam_amd.evaluate (device matcher).

Pinned by tests/golden/crowdhuman_eval.npz, produced by importing the reference evaluator in place
(oracle/make_goldens.py::golden_evaluator; pure numpy, zero shims).

Reference lines restated:
  Image.load_gt_boxes            :262-297   odgt records -> [x0,y0,x1,y1,tag] (vbox when visible_flag, else fbox;
                                            tag 1 = person, -1 = other class / extra.ignore != 0)
  Image.load_cocojson (dets)     :27-67     COCO detections: xywh -> xyxy, score column
  Image.clip_all_boader          :238-260   clip x0,y0 to [0,w-1]/[0,h-1] and x1,y1 to [0,w]/[0,h] (dets AND gts)
  Image.box_overlap_opr          :215-236   IoU = inter/(da+ga-inter+1e-6), IoA = inter/(da+1e-6)
  Image.compare_caltech          :113-143   greedy match in descending score, matched GT column zeroed,
                                            unmatched dets overlapping an ignore region are dropped
  Database.compare / eval_AP / eval_MR :436-548
"""
import json

import numpy as np

PERSON_CLASSES = ["background", "person"]


def load_gt_boxes(record, visible_flag):
    """crowdhuman_eval.py:262-297 (body boxes only: the head branch is commented out in the reference)."""
    rows = []
    for rb in record["gtboxes"]:
        tag = PERSON_CLASSES.index(rb["tag"]) if rb["tag"] in PERSON_CLASSES else -1
        if "extra" in rb and "ignore" in rb["extra"] and rb["extra"]["ignore"] != 0:
            tag = -1
        if visible_flag:
            box = rb["vbox"][0] if isinstance(rb["vbox"][0], list) else rb["vbox"]
        else:
            box = rb["fbox"]
        rows.append((*box, tag))
    if not rows:
        return np.empty([0, 5])
    a = np.array(rows)          # integer inputs stay integer, exactly like np.array(list of tuples) in the reference
    a[:, 2:4] += a[:, :2]
    return a


def box_overlap(d, g, if_iou):
    """crowdhuman_eval.py:215-236."""
    eps = 1e-6
    dt = d[:, None, :]
    gt = g[None, :, :]
    iw = np.minimum(dt[:, :, 2], gt[:, :, 2]) - np.maximum(dt[:, :, 0], gt[:, :, 0])
    ih = np.minimum(dt[:, :, 3], gt[:, :, 3]) - np.maximum(dt[:, :, 1], gt[:, :, 1])
    inter = np.maximum(0, iw) * np.maximum(0, ih)
    da = (dt[:, :, 2] - dt[:, :, 0]) * (dt[:, :, 3] - dt[:, :, 1])
    if if_iou:
        ga = (gt[:, :, 2] - gt[:, :, 0]) * (gt[:, :, 3] - gt[:, :, 1])
        return inter / (da + ga - inter + eps)
    return inter / (da + eps)


def clip_boundary(boxes, height, width):
    """crowdhuman_eval.py:240-246 (in place on the first four columns)."""
    boxes[:, 0] = np.minimum(np.maximum(boxes[:, 0], 0), width - 1)
    boxes[:, 1] = np.minimum(np.maximum(boxes[:, 1], 0), height - 1)
    boxes[:, 2] = np.maximum(np.minimum(boxes[:, 2], width), 0)
    boxes[:, 3] = np.maximum(np.minimum(boxes[:, 3], height), 0)
    return boxes


def compare_caltech(dtboxes, gtboxes, thres):
    """crowdhuman_eval.py:113-143 -> list of (score, label, pos) in descending-score order.
    An image whose GT holds no positive box makes the reference raise (argmax of an empty row); here such
    detections are false positives unless an ignore region covers them."""
    if dtboxes is None or gtboxes is None or len(dtboxes) == 0 or len(gtboxes) == 0:
        return []
    dt = dtboxes[np.argsort(-dtboxes[:, -1], kind="stable")]
    gt = gtboxes[np.argsort(-gtboxes[:, -1], kind="stable")]
    iou = box_overlap(dt, gt[gt[:, -1] > 0], True)
    ioa = box_overlap(dt, gt[gt[:, -1] <= 0], False)
    ign = np.any(ioa > thres, 1)
    pos = np.any(iou > thres, 1)
    out = []
    for i in range(len(dt)):
        if iou.shape[1]:
            j = int(np.argmax(iou[i]))
            if iou[i, j] > thres:
                iou[:, j] = 0
                out.append((dt[i, -1], 1, bool(pos[i])))
                continue
        if not ign[i]:
            out.append((dt[i, -1], 0, bool(pos[i])))
    return out


def load_database(gt_path, dt_path, remove_empty_gt, visible_flag):
    """Database.__init__/loadData_odgt/loadData (crowdhuman_eval.py:361-434) for the combination tools/batch_eval.py
    uses: GT from an .odgt file, detections from a COCO-format json.  -> list of per-image dicts in GT order."""
    with open(gt_path) as f:
        lines = f.readlines()
    records = json.loads(lines[0]) if len(lines) == 1 else [json.loads(l) for l in lines]
    images = {}
    for rec in records:
        gt = load_gt_boxes(rec, visible_flag)
        images[rec["ID"]] = {"ID": rec["ID"], "gt": gt, "gt_num": len(rec["gtboxes"]),
                             "ign_num": int((gt[:, -1] == -1).sum()), "dt": None, "w": rec.get("width"),
                             "h": rec.get("height")}
    det = json.load(open(dt_path))
    annots = det["annotations"]
    a = 0
    for item in det["images"]:
        k = 0
        while a + k < len(annots) and annots[a + k]["image_id"] == item["id"]:
            k += 1
        im = images[item["id"]]
        if im["w"] is None:
            im["w"] = item["width"]
        if im["h"] is None:
            im["h"] = item["height"]
        mine = annots[a:a + k]
        boxes = np.array([x["bbox"] for x in mine])
        if len(boxes) > 0:
            boxes[:, 2:4] = boxes[:, :2] + boxes[:, 2:4]
        else:
            boxes = np.zeros((0, 4))
        if len(mine) > 0 and "score" in mine[0]:
            scores = np.array([x["score"] for x in mine])[:, None]
        else:
            scores = np.ones((len(boxes), 1))
        im["dt"] = np.concatenate([boxes, scores], axis=-1)
        im["dt"] = clip_boundary(im["dt"], im["h"], im["w"])
        im["gt"] = clip_boundary(im["gt"], im["h"], im["w"])
        a += k
    ims = list(images.values())
    if remove_empty_gt:
        ims = [im for im in ims if im["dt"] is not None]
    return ims


def evaluate(gt_path, dt_path, remove_empty_gt=False, visible_flag=False, thres=0.5):
    """_evaluate_predictions_on_crowdhuman (crowdhuman_eval.py:550-558) -> dict of AP, MR, recall, tp, fp and curves."""
    ims = load_database(gt_path, dt_path, remove_empty_gt, visible_flag)
    gt_num = sum(im["gt_num"] for im in ims)
    ign_num = sum(im["ign_num"] for im in ims)
    n_img = len(ims)
    scorelist = []
    for im in ims:
        scorelist.extend(compare_caltech(im["dt"], im["gt"], thres))
    scorelist.sort(key=lambda x: x[0], reverse=True)
    total_gt = gt_num - ign_num
    tp = fp = 0.0
    rpx, rpy, fppi, mr = [], [], [], []
    for score, label, pos in scorelist:
        if label == 1:
            tp += 1.0
        else:
            fp += 1.0
        recall = tp / (tp + (total_gt - tp))
        rpx.append(recall)
        rpy.append(tp / (tp + fp))
        fppi.append(fp / n_img)
        mr.append(1 - recall)
    ap = 0
    for i in range(1, len(rpx)):
        ap += (rpx[i] - rpx[i - 1]) * ((rpy[i - 1] + rpy[i]) / 2)
    ref = [0.0100, 0.0178, 0.03160, 0.0562, 0.1000, 0.1778, 0.3162, 0.5623, 1.000]
    pts = []
    for p in ref:
        idx = len(fppi) - 1
        for i, v in enumerate(fppi):
            if v >= p:
                idx = i
                break
        if idx >= 0:
            pts.append(mr[idx])
    mmr = float(np.exp(np.log(np.array(pts)).mean()))
    return {"AP": ap, "MR": mmr, "recall": rpx[-1] if rpx else 0.0, "tp": int(tp), "fp": int(fp),
            "recall_curve": np.array(rpx), "precision_curve": np.array(rpy), "fppi": np.array(fppi),
            "labels": np.array([s[1] for s in scorelist]), "scores": np.array([s[0] for s in scorelist])}
