#!/usr/bin/env python
"""Multi-GPU evaluation: image-sharded, one process per GPU (reference: tools/batch_eval.py:8-106).

    python tools/batch_eval.py -c configs/crowdhuman.yaml -n 8 [key.sub value ...]

Same argv as the reference and the same launch contract: invoked as above it starts the ``-n`` ranks ITSELF
(reference: one ``tools/test.py`` subprocess per GPU, :8-18,76-92).  Here the ranks are one ``torch.distributed``
process group (backend "nccl" == RCCL over xGMI), each rank takes the reference's contiguous index range (:80-89) and
the detections are gathered with ONE variable-length all-gather (crowdsam_amd.distributed.gather_rows) instead of
``temp_result_{rank}.json`` files + ``merge_json`` (:12,20-29); rank 0 writes the COCO-format detection json the
reference's evaluator consumes (``convert_to_coco``, :31-58) and evaluates it (the reference shells out to
tools/crowdhuman_eval.py, :100-102; same numbers here from the device matcher).  Launched under ``torchrun`` /
``python -m torch.distributed.run`` (WORLD_SIZE set) it runs as a worker of that group.

Build extensions: ``--synthetic N`` (seeded weights + synthetic crowd frames, no dataset needed), ``-o`` output json,
``--backend gloo`` (CPU test of the launch + gather path), ``--keep_json`` (the reference deletes test.json).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdsam.utils as utils  # noqa: E402
from crowdsam_amd.distributed import detections_to_rows, gather_rows, shard_range  # noqa: E402


def convert_to_coco(det_result, gt_js):
    """Per-image results [{image_id, boxes (xyxy), scores}] + the GT json -> COCO detection dict, as the reference's
    tools/batch_eval.py:31-58: image ids become file_name[:-4], boxes xywh, running annotation ids."""
    images = gt_js["images"]
    for im in images:
        im["id"] = im["file_name"][:-4]
    annots = []
    for k, item in enumerate(det_result):
        image_id = images[k]["id"] if images != [] else item["image_id"]
        for score, box in zip(item["scores"], item["boxes"]):
            x0, y0, x1, y1 = (float(v) for v in box)
            annots.append({"category_id": 1, "bbox": [x0, y0, x1 - x0, y1 - y0], "image_id": image_id, "iscrowd": False,
                           "area": (y1 - y0) * (x1 - x0), "id": len(annots), "score": float(score)})
    return {"images": images, "annotations": annots, "categories": gt_js["categories"]}


def merge_json(json_files):
    """tools/batch_eval.py:20-29 (kept for users of the file-based flow: concatenates and removes the rank files)."""
    merged = []
    for f in json_files:
        with open(f) as fh:
            merged.extend(json.load(fh))
    for f in json_files:
        os.remove(f)
    return merged


def rows_to_results(rows, n_images):
    """Gathered rows [n,6] = (image_index, x0,y0,x1,y1, score) -> the per-image list tools/test.py writes."""
    out = [{"image_id": i, "boxes": [], "scores": []} for i in range(n_images)]
    for r in rows:
        out[int(r[0])]["boxes"].append([float(v) for v in r[1:5]])
        out[int(r[0])]["scores"].append(float(r[5]))
    return out


def parse(argv=None):
    parser = argparse.ArgumentParser(description="Run the evaluation image-sharded over several GPUs")
    parser.add_argument("-n", "--num_nodes", type=int, default=8, help="Number of nodes (GPUs / processes) to use")
    parser.add_argument("-c", "--config_file", default="./configs/crowdhuman.yaml")
    parser.add_argument("--synthetic", type=int, default=0)
    parser.add_argument("--profile", action="store_true",
                        help="per-stage times (device-synchronised) into <output_dir>/timings_rank<r>.json + roctx ranges")
    parser.add_argument("-o", "--output", default="test.json")
    parser.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    parser.add_argument("--keep_json", action="store_true")
    parser.add_argument("options", nargs=argparse.REMAINDER)
    return parser.parse_args(argv)


def launch_ranks(args, argv):
    """The reference's ``-n N`` contract: this process starts the N ranks (one per GPU) and waits for them."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.num_nodes}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"Running command: {' '.join(cmd)}")
    rc = subprocess.run(cmd).returncode
    if rc != 0:
        raise SystemExit(f"batch_eval: a rank failed (exit code {rc})")     # the reference ignores rank failures (:18)
    print("All processes done")


def worker(args):
    config = utils.modify_config(utils.load_config(args.config_file), args.options)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if rank == 0:
        print(yaml.dump(config, default_flow_style=False, default_style=""))
    gpu = args.backend == "nccl"
    if gpu:
        torch.cuda.set_device(local)
        config["environ"]["device"] = f"cuda:{local}"
    if world > 1:
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", local)} if gpu else {}))
    np.random.seed(config["environ"]["seed"])
    logger = utils.setup_logger(config["environ"]["output_dir"] + "/log", quiet=rank != 0)
    if args.synthetic:
        names = [f"synthetic_{i}.jpg" for i in range(args.synthetic)]
        gt_js = {"images": [{"file_name": n, "width": 1024, "height": 1024} for n in names], "categories": []}
    else:
        gt_js = json.load(open(config["data"]["json_file"]))
    num_imgs = len(gt_js["images"])
    start, end = shard_range(num_imgs, rank, world)           # tools/batch_eval.py:80-89
    rows = [np.zeros((0, 6), np.float32)]
    if gpu:
        from crowdsam.model import CrowdSAM
        if args.synthetic:
            from crowdsam_amd import synth
            model = CrowdSAM(config, logger, sam_state_dict=synth.make_sam_state_dict(config["model"]["sam_model"]),
                             dino_state_dict=synth.make_dino_state_dict())
            load = synth.synthetic_crowd_frame
        else:
            model = CrowdSAM(config, logger)
            d = config["data"]
            load = lambda i: utils.load_img_and_annotation(d["dataset_root"], gt_js, d["dataset"], i)[0]
        import crowdsam.model as cm
        if args.profile:
            cm.profile(True)
        cm.settle_host()                 # long-lived objects out of the garbage collector's way (crowdsam.model.settle_host)
        t_run, kept = time.perf_counter(), 0
        # the rank's shard as ONE stream (CrowdSAM.generate_stream: the next frames' encoders run beside this frame's tail)
        for i, out in zip(range(start, end), cm.settled(model.generate_stream(load(i) for i in range(start, end)))):
            rows.append(detections_to_rows(i, out["boxes"], out["scores"]))
            kept += len(out["boxes"])
        if args.profile:
            from crowdsam_amd import trace
            trace.write_timings(os.path.join(config["environ"]["output_dir"], "timings_rank%d.json" % rank), rank, model,
                                end - start, kept, time.perf_counter() - t_run)
    else:   # --backend gloo: launch / shard / gather plumbing only (one deterministic fake detection per image)
        for i in range(start, end):
            rows.append(detections_to_rows(i, [[i, i, i + 10, i + 20]], [1.0 / (1 + i)]))
    allrows = gather_rows(np.concatenate(rows))
    if rank == 0:
        coco_json = convert_to_coco(rows_to_results(allrows, num_imgs), gt_js)
        json.dump(coco_json, open(args.output, "w"), ensure_ascii=True)
        logger.info("wrote %d detections over %d images to %s", len(allrows), num_imgs, args.output)
        odgt = config.get("data", {}).get("odgt_file") if not args.synthetic else None
        if gpu and odgt and os.path.exists(odgt):
            # tools/batch_eval.py:100-102 (-d test.json -g odgt --remove_empty_gt --visible_flag)
            from crowdsam_amd import evaluate as ev
            r = ev.evaluate(odgt, coco_json, remove_empty_gt=True, visible_flag=True)
            print("AP:{:.4f}, MR:{:.4f}, Recall:{:.4f}, tp:{}, fp:{}".format(r["AP"], r["MR"], r["recall"], r["tp"], r["fp"]))
        if not args.keep_json and args.output == "test.json":
            os.remove("test.json")                            # as the reference (:103)
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.num_nodes > 1:
        return launch_ranks(args, argv)
    return worker(args)


if __name__ == "__main__":
    main()
