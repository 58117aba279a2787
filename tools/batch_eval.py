#!/usr/bin/env python
"""Multi-GPU evaluation: image-sharded, one process per GPU (reference: tools/batch_eval.py:8-106).

The reference spawns ``tools/test.py`` subprocesses and merges ``temp_result_{rank}.json`` files.
Here the same contiguous shards run under torch.distributed and the detections are gathered with one
RCCL all-gather over xGMI (crowdsam_amd.distributed.gather_rows); rank 0 writes the COCO-format
detection json the reference's evaluator consumes (xyxy -> xywh, image_id = file_name[:-4]).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/batch_eval.py -c configs/crowdhuman.yaml -n 8 [--synthetic 64] [key.sub value ...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdsam.utils import load_config, load_coco_index, load_img_and_annotation, modify_config, setup_logger  # noqa: E402
from crowdsam_amd.distributed import detections_to_rows, gather_rows, shard_range  # noqa: E402


def convert_to_coco(det_result, gt_js):
    """Per-image results [{image_id, boxes (xyxy), scores}] + the GT json -> COCO detection dict, as the reference's
    tools/batch_eval.py:31-58: image ids become file_name[:-4], boxes xywh, running annotation ids."""
    images = gt_js["images"]
    for im in images:
        im["id"] = im["file_name"][:-4]
    annots = []
    for k, item in enumerate(det_result):
        image_id = images[k]["id"] if images else item["image_id"]
        for score, box in zip(item["scores"], item["boxes"]):
            x0, y0, x1, y1 = (float(v) for v in box)
            annots.append({"category_id": 1, "bbox": [x0, y0, x1 - x0, y1 - y0], "image_id": image_id, "iscrowd": False,
                           "area": (y1 - y0) * (x1 - x0), "id": len(annots), "score": float(score)})
    return {"images": images, "annotations": annots, "categories": gt_js.get("categories", [])}


def rows_to_results(rows, n_images):
    """Gathered rows [n,6] = (image_index, x0,y0,x1,y1, score) -> the per-image list tools/test.py writes."""
    out = [{"image_id": i, "boxes": [], "scores": []} for i in range(n_images)]
    for r in rows:
        out[int(r[0])]["boxes"].append([float(v) for v in r[1:5]])
        out[int(r[0])]["scores"].append(float(r[5]))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config_file", default="./configs/crowdhuman_mi355x.yaml")
    ap.add_argument("-n", "--num_nodes", type=int, default=1, help="number of GPUs (processes)")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("-o", "--output", default="test.json")
    ap.add_argument("options", nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    config = modify_config(load_config(args.config_file), args.options)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    config["environ"]["device"] = f"cuda:{local}"
    np.random.seed(config["environ"]["seed"])
    logger = setup_logger(config["environ"]["output_dir"], quiet=rank != 0)
    from crowdsam.model import CrowdSAM
    if args.synthetic:
        from crowdsam_amd import synth
        model = CrowdSAM(config, logger, sam_state_dict=synth.make_sam_state_dict(config["model"]["sam_model"]),
                         dino_state_dict=synth.make_dino_state_dict())
        names = [f"synthetic_{i}.jpg" for i in range(args.synthetic)]
        load = lambda i: synth.synthetic_crowd_frame(i)
    else:
        model = CrowdSAM(config, logger)
        d = config["data"]
        coco = load_coco_index(d["json_file"])
        ids = coco.getImgIds()
        names = [coco["images_by_id"][i]["file_name"] for i in ids]
        load = lambda i: load_img_and_annotation(d["dataset_root"], d["dataset"], ids[i], coco)[0]
    start, end = shard_range(len(names), rank, world)
    rows = [np.zeros((0, 6), np.float32)]
    for i in range(start, end):
        out = model.generate(load(i))
        rows.append(detections_to_rows(i, out["boxes"], out["scores"]))
    allrows = gather_rows(np.concatenate(rows))
    if rank == 0:
        gt_js = {"images": [{"file_name": n, "width": 1024, "height": 1024} for n in names], "categories": []} \
            if args.synthetic else json.load(open(config["data"]["json_file"]))
        coco_json = convert_to_coco(rows_to_results(allrows, len(names)), gt_js)
        with open(args.output, "w") as f:
            json.dump(coco_json, f, ensure_ascii=True)
        logger.info("wrote %d detections over %d images to %s", len(allrows), len(names), args.output)
        odgt = config.get("data", {}).get("odgt_file") if not args.synthetic else None
        if odgt and os.path.exists(odgt):
            # the reference shells out to tools/crowdhuman_eval.py (-d test.json -g odgt --remove_empty_gt
            # --visible_flag, :100-103); same numbers here straight from the gathered rows, matching on the GPU
            from crowdsam_amd import evaluate as ev
            r = ev.evaluate(odgt, coco_json, remove_empty_gt=True, visible_flag=True)
            logger.info("AP: %.4f, MR: %.4f, Recall: %.4f, tp: %d, fp: %d", r["AP"], r["MR"], r["recall"], r["tp"], r["fp"])
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
