#!/usr/bin/env python
"""Single-GPU evaluation harness with the reference's argv and result JSON (reference: tools/test.py:13-89).

    python tools/test.py -c configs/crowdhuman.yaml [--start_idx A --end_idx B] [-r LOCAL_RANK]
                         [-s SAVE_PATH] [-v] [key.sub value ...]

Writes ``[{image_id, num_gt, boxes, scores, categories, rles}, ...]`` to SAVE_PATH.  ``--synthetic N``
(build extension) runs N synthetic crowd frames with seeded weights when no dataset / checkpoints exist.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdsam.model import CrowdSAM  # noqa: E402
from crowdsam.utils import (data_meta, load_config, load_coco_index, load_img_and_annotation, modify_config,  # noqa: E402
                            setup_logger)


def environ_init(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config_file", default="./configs/crowdhuman_mi355x.yaml")
    ap.add_argument("--start_idx", type=int, default=0)
    ap.add_argument("--end_idx", type=int, default=-1)
    ap.add_argument("-r", "--local_rank", type=int, default=0)
    ap.add_argument("-s", "--save_path", default="result.json")
    ap.add_argument("-v", "--visualize", action="store_true")
    ap.add_argument("--synthetic", type=int, default=0, help="run N synthetic frames with seeded weights")
    ap.add_argument("options", nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    config = modify_config(load_config(args.config_file), args.options)
    seed = config["environ"]["seed"]
    np.random.seed(seed)              # the EPS shuffle draws from the global NumPy RNG (trap 7)
    torch.manual_seed(seed)
    logger = setup_logger(config["environ"]["output_dir"])
    return args, config, logger


def run(args, config, logger):
    torch.cuda.set_device(args.local_rank)
    config["environ"]["device"] = f"cuda:{args.local_rank}"
    results = []
    if args.synthetic:
        from crowdsam_amd import synth
        model = CrowdSAM(config, logger, sam_state_dict=synth.make_sam_state_dict(config["model"]["sam_model"]),
                         dino_state_dict=synth.make_dino_state_dict())
        ids = list(range(args.synthetic))
        end = len(ids) if args.end_idx < 0 else args.end_idx
        for i in ids[args.start_idx:end]:
            out = model.generate(synth.synthetic_crowd_frame(i))
            results.append({"image_id": f"synthetic_{i}.jpg", "num_gt": 0, "boxes": out["boxes"].tolist(),
                            "scores": out["scores"].tolist(), "categories": out["categories"].tolist(),
                            "rles": out["rles"]})
    else:
        model = CrowdSAM(config, logger)
        d = config["data"]
        coco = load_coco_index(d["json_file"])
        ids = coco.getImgIds()
        end = len(ids) if args.end_idx < 0 else args.end_idx
        for id_ in ids[args.start_idx:end]:
            image, file_name, gt = load_img_and_annotation(d["dataset_root"], d["dataset"], id_, coco)
            out = model.generate(image)
            results.append({"image_id": file_name, "num_gt": int(len(gt)), "boxes": out["boxes"].tolist(),
                            "scores": out["scores"].tolist(), "categories": out["categories"].tolist(),
                            "rles": out["rles"]})
    with open(args.save_path, "w") as f:
        json.dump(results, f)
    return results


if __name__ == "__main__":
    run(*environ_init())
