#!/usr/bin/env python
"""CrowdHuman evaluation CLI with the argv and record line of the reference's tools/crowdhuman_eval.py:575-595:

    python tools/crowdhuman_eval.py -d test.json -g annotation_val.odgt --remove_empty_gt --visible_flag

Matching runs on the GPU (crowdsam_amd.evaluate / csam_caltech_match); prints AP, MR, Recall, tp, fp and appends
the same comma-separated line to the record file."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdsam_amd import evaluate as ev  # noqa: E402


def _evaluate_predictions_on_crowdhuman(gt_path, dt_path, target_key="boxes", mode=0, remove_empty_gt=False,
                                        visible_flag=False):
    """Same name / argument order / return tuple as the reference helper (crowdhuman_eval.py:550-558)."""
    if mode != 0:
        raise NotImplementedError("only eval mode 0 (body boxes) is reachable from tools/batch_eval.py")
    r = ev.evaluate(gt_path, dt_path, remove_empty_gt, visible_flag)
    return r["AP"], r["MR"], r["recall"], r["tp"], r["fp"]


def main(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate detections in CrowdHuman format (COCO or odgt ground truth).")
    ap.add_argument("-d", "--det_path", type=str)
    ap.add_argument("-g", "--gt_path", type=str, default="")
    ap.add_argument("-o", "--output_path", type=str, default="./record.txt")
    ap.add_argument("-f", "--remove_empty_gt", action="store_true")
    ap.add_argument("-v", "--visible_flag", action="store_true")
    args = ap.parse_args(argv)
    res = _evaluate_predictions_on_crowdhuman(args.gt_path, args.det_path, remove_empty_gt=args.remove_empty_gt,
                                              visible_flag=args.visible_flag)
    names = ["AP", "MR", "Recall", "tp", "fp"]
    for k, v in zip(names, res):
        print(f"{k}: {v}")
    with open(args.output_path, "a") as f:
        f.write(", ".join(f"{k}: {v:.4f}" for k, v in zip(names, res)) + "\n")
    return res


if __name__ == "__main__":
    main()
