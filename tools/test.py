#!/usr/bin/env python
"""Single-GPU evaluation harness: the reference's argv, call sequence and result JSON (reference: tools/test.py:13-89).

    python tools/test.py -c configs/crowdhuman.yaml [--start_idx A --end_idx B] [-r LOCAL_RANK]
                         [-s SAVE_PATH] [-v] [--mode seg|bbox] [key.sub value ...]

Writes ``[{image_id, num_gt, boxes, scores, categories, rles}, ...]`` to SAVE_PATH (default
``<environ.output_dir>/result.json``).  It uses only the surface the reference's own script imports
(``crowdsam.model.CrowdSAM``; ``crowdsam.utils.{load_img_and_annotation, setup_logger, data_meta, load_config,
modify_config, visualize_result, evaluate_boxes}``), so the reference's unmodified tools/test.py runs against this
package as well (tests/test_tools_dropin_*.py replay its statements).  ``--synthetic N`` (build extension) runs N
synthetic crowd frames with seeded weights when no dataset / checkpoints exist.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdsam.model import CrowdSAM  # noqa: E402
from crowdsam.utils import (data_meta, evaluate_boxes, load_config, load_img_and_annotation, modify_config,  # noqa: E402
                            setup_logger, visualize_result)


def envrion_init(argv=None):
    """argparse + YAML + seeds + logger (tools/test.py:13-35; the function name is the reference's spelling)."""
    parser = argparse.ArgumentParser(description="CrowdSAM argparser")
    parser.add_argument("--mode", type=str, choices=["seg", "bbox"], default="seg")
    parser.add_argument("--start_idx", type=int, default=0)
    parser.add_argument("--end_idx", type=int, default=-1)          # -1: all images
    parser.add_argument("-c", "--config_file", type=str, default="./configs/crowdhuman.yaml")
    parser.add_argument("-v", "--visualize", help="visualize the outputs", action="store_true")
    parser.add_argument("-s", "--save_path", help="the path to dump json result", type=str, default="")
    parser.add_argument("-r", "--local_rank", type=int, default=0)
    parser.add_argument("--synthetic", type=int, default=0, help="(build extension) N synthetic frames, seeded weights")
    parser.add_argument("--profile", action="store_true",
                        help="(build extension; the reference has no tracing, SURVEY.md section 5) per-stage times -- "
                             "device-synchronised, so the stages do not overlap -- into <output_dir>/timings_rank<r>.json, and "
                             "roctx ranges for `rocprofv3 --marker-trace --kernel-trace -- python tools/test.py --profile ...`")
    parser.add_argument("options", nargs=argparse.REMAINDER)
    args = parser.parse_args(argv)
    configs = modify_config(load_config(args.config_file), args.options)
    np.random.seed(configs["environ"]["seed"])          # the EPS shuffle draws from the global NumPy RNG (trap 7)
    torch.random.manual_seed(configs["environ"]["seed"])
    os.makedirs(configs["environ"]["output_dir"], exist_ok=True)
    os.makedirs(configs["environ"]["output_dir"] + "/log", exist_ok=True)
    logger = setup_logger(configs["environ"]["output_dir"] + "/log")
    logger.info(args)
    return args, configs, logger


environ_init = envrion_init


def instance_record(image_id, num_gt, result):
    """One entry of the result JSON (tools/test.py:69-72): only the fields present are copied, so an image without
    detections (no 'categories' in the reference's MaskData) does not abort the run."""
    d = {"image_id": image_id, "num_gt": num_gt}
    d.update({k: v.tolist() for k, v in result.items() if k in ["boxes", "scores", "categories"]})
    d.update({k: v for k, v in result.items() if k in ["rles"]})
    return d


def run(args, config, logger):
    dataset_path = config["data"]["dataset_root"]
    n_class, class_names = data_meta[config["data"]["dataset"]][1:]
    if "cuda" in config["environ"]["device"]:
        torch.cuda.set_device(args.local_rank)
        print(f"set device cuda:{args.local_rank}")
        config["environ"]["device"] = f"cuda:{args.local_rank}"
    output_content = []
    import crowdsam.model as _cm
    if getattr(args, "profile", False):
        _cm.profile(True)
    t_run = time.perf_counter()
    if args.synthetic:
        from crowdsam_amd import synth
        model = CrowdSAM(config, logger, sam_state_dict=synth.make_sam_state_dict(config["model"]["sam_model"]),
                         dino_state_dict=synth.make_dino_state_dict())
        _cm.settle_host()
        end_idx = args.synthetic if args.end_idx == -1 else min(args.end_idx, args.synthetic)
        ids = list(range(args.start_idx, end_idx))
        # one frame of look-ahead (CrowdSAM.generate_stream): frame i+1's encoders run beside frame i's tail
        for id_, result in zip(ids, _cm.settled(model.generate_stream(synth.synthetic_crowd_frame(i) for i in ids))):
            output_content.append(instance_record(f"synthetic_{id_}.jpg", 0, result))
    else:
        model = CrowdSAM(config, logger)
        _cm.settle_host()                # long-lived objects out of the garbage collector's way (one 100 ms frame in ~90 otherwise)
        logger.info("load images and annotations from crowdhuman dataset..")
        annots = json.load(open(config["data"]["json_file"]))
        end_idx = len(annots["images"]) if args.end_idx == -1 else min(args.end_idx, len(annots["images"]))
        image_ids = [i for i in range(args.start_idx, end_idx)]
        logger.info(f"total images  to process { len(image_ids)}")
        loaded = (load_img_and_annotation(dataset_path, annots, config["data"]["dataset"], i) for i in image_ids)
        # The look-ahead is this build's extension (a model with the reference's plain generate(image) is called as is):
        # CrowdSAM.generate_stream reads ahead of the frame it returns -- one frame with the shipped EPS configuration, a
        # group of test.encoder_batch frames (one image-batched encoder pass) with a dense sweep -- so the records of the
        # frames in flight wait in a queue
        import collections
        pending = collections.deque()

        def frames():
            for rec in loaded:
                pending.append(rec)
                yield rec[0]

        results = model.generate_stream(frames()) if hasattr(model, "generate_stream") else (model.generate(f) for f in frames())
        for id_, result in zip(image_ids, _cm.settled(results)):
            logger.debug(f"start processing {id_}")
            image, gt_boxes, image_id = pending.popleft()
            output_content.append(instance_record(image_id, len(gt_boxes) - 1, result))     # "- 1" as tools/test.py:69
            logger.debug(f"process for image:{id_} is done")
            if args.visualize:
                save_path = os.path.join(config["environ"]["output_dir"], f"{id_}.jpg")
                result["gt_boxes"] = gt_boxes
                FP_list, FN_list = evaluate_boxes(result["boxes"], result["scores"], gt_boxes, 0.5)[2:]
                visualize_result(image, result, class_names, save_path, conf_thresh=config["vis"]["vis_thresh"],
                                 FP_ind=FP_list, FN_ind=FN_list, vis_masks=args.mode == "seg")
            del result
    if getattr(args, "profile", False):
        from crowdsam_amd import trace
        trace.write_timings(os.path.join(config["environ"]["output_dir"], "timings_rank%d.json" % args.local_rank), args.local_rank,
                            model, len(output_content), sum(len(r.get("boxes", [])) for r in output_content),
                            time.perf_counter() - t_run)
    if args.save_path == "":
        file_path = os.path.join(config["environ"]["output_dir"], "result.json")
        print(f"dump json file to {file_path}")
        json.dump(output_content, open(file_path, "w"), ensure_ascii=True)
    else:
        json.dump(output_content, open(args.save_path, "w"), ensure_ascii=True)
    return output_content


if __name__ == "__main__":
    run(*envrion_init())
