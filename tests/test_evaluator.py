"""CrowdHuman evaluator: oracle vs the goldens captured from the reference (CPU), device matcher vs oracle (GPU)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GT = os.path.join(HERE, "golden", "crowdhuman_eval", "gt.odgt")
DT = os.path.join(HERE, "golden", "crowdhuman_eval", "det.json")
CASES = [(rm, vis) for rm in (False, True) for vis in (False, True)]


def _golden(rm, vis):
    g = np.load(os.path.join(HERE, "golden", "crowdhuman_eval.npz"))
    k = "rm%d_vis%d" % (rm, vis)
    return {n: g[k + "_" + n] for n in ("summary", "labels", "scores", "recall", "precision")}


def _check(r, g):
    s = g["summary"]
    assert (r["AP"], r["MR"], r["recall"], r["tp"], r["fp"]) == (s[0], s[1], s[2], s[3], s[4])   # bit-exact float64
    assert np.array_equal(r["labels"], g["labels"])
    assert np.array_equal(r["scores"], g["scores"])
    assert np.array_equal(r["recall_curve"], g["recall"])
    assert np.array_equal(r["precision_curve"], g["precision"])


@pytest.mark.parametrize("rm,vis", CASES)
def test_oracle_matches_reference_golden(rm, vis):
    from oracle import eval_oracle as eo
    _check(eo.evaluate(GT, DT, rm, vis), _golden(rm, vis))


@pytest.mark.gpu
@pytest.mark.parametrize("rm,vis", CASES)
def test_device_evaluator_matches_golden(cuda, rm, vis):
    from crowdsam_amd import evaluate as ev
    _check(ev.evaluate(GT, DT, rm, vis), _golden(rm, vis))


@pytest.mark.gpu
def test_device_matcher_random_vs_oracle(cuda):
    """Random crowded images incl. all-ignore GT, empty detections, exact IoU ties and > 64 boxes per image."""
    from crowdsam_amd import evaluate as ev
    from oracle import eval_oracle as eo
    rs = np.random.RandomState(3)
    recs = []
    for i in range(40):
        n_gt, n_dt = int(rs.randint(0, 200)), int(rs.randint(0, 300))
        xy = rs.randint(0, 600, (n_gt, 2)).astype(np.float64)
        wh = rs.randint(5, 120, (n_gt, 2)).astype(np.float64)
        tag = np.where(rs.rand(n_gt) < (1.0 if i == 7 else 0.15), -1.0, 1.0)
        r = ev.ImageRecord(i, 640, 640)
        r.gt = np.concatenate([xy, xy + wh, tag[:, None]], 1)
        dxy = rs.randint(0, 600, (n_dt, 2)).astype(np.float64)
        dwh = rs.randint(5, 120, (n_dt, 2)).astype(np.float64)
        r.dt = np.concatenate([dxy, dxy + dwh, np.round(rs.rand(n_dt, 1), 1)], 1)
        if n_gt and n_dt:
            r.dt[: min(n_gt, n_dt) // 2, :4] = r.gt[: min(n_gt, n_dt) // 2, :4]      # exact duplicates -> IoU ties
        recs.append(r)
    scores, labels, pos = ev.match(recs, 0.5)
    ref = []
    for r in recs:
        ref.extend(eo.compare_caltech(r.dt, r.gt, 0.5))
    ref.sort(key=lambda x: x[0], reverse=True)
    assert np.array_equal(scores, np.array([x[0] for x in ref]))
    assert np.array_equal(labels, np.array([x[1] for x in ref], dtype=np.int8))
    assert np.array_equal(pos, np.array([x[2] for x in ref], dtype=bool))


@pytest.mark.gpu
def test_cli_and_rows(cuda, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import crowdhuman_eval as cli
    from crowdsam_amd import evaluate as ev
    rec = tmp_path / "record.txt"
    res = cli.main(["-d", DT, "-g", GT, "-o", str(rec), "--remove_empty_gt", "--visible_flag"])
    s = _golden(True, True)["summary"]
    assert tuple(res) == (s[0], s[1], s[2], s[3], s[4])
    assert rec.read_text().startswith("AP: %.4f, MR: %.4f" % (s[0], s[1]))
    # gathered rows -> same numbers without any file
    coco = json.load(open(DT))
    records = [json.loads(l) for l in open(GT)]
    gt_images = [{"file_name": r["ID"] + ".jpg", "width": 0, "height": 0} for r in records]
    wh = {im["id"]: (im["width"], im["height"]) for im in coco["images"]}
    for im in gt_images:
        if im["file_name"][:-4] in wh:
            im["width"], im["height"] = wh[im["file_name"][:-4]]
    index = {r["ID"]: i for i, r in enumerate(records)}
    rows = [[index[a["image_id"]], a["bbox"][0], a["bbox"][1], a["bbox"][0] + a["bbox"][2], a["bbox"][1] + a["bbox"][3],
             a["score"]] for a in coco["annotations"]]
    r = ev.evaluate_rows(np.array(rows), gt_images, GT, remove_empty_gt=True, visible_flag=True)
    assert abs(r["AP"] - s[0]) < 1e-12 and r["tp"] == s[3] and r["fp"] == s[4]
