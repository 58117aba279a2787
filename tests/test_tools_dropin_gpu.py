"""Row b on the GPU: the build's tools/test.py run as the reference is run -- a YAML config with checkpoint PATHS and a
dataset directory, nothing injected -- end to end on the HIP path (seeded weights written as checkpoint files)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_tools_test_py_from_checkpoint_files(tmp_path, cuda):
    from crowdsam_amd import synth
    arch = "vit_test128"
    sd = synth.make_sam_state_dict(arch)
    torch.save(sd, tmp_path / "sam.pth")
    torch.save({k[len("mask_decoder."):]: v for k, v in sd.items() if k.startswith("mask_decoder.")}, tmp_path / "adapter.pth")
    torch.save(synth.make_dino_state_dict(depth=2), tmp_path / "dino.pth")
    root = tmp_path / "crowdhuman"
    (root / "Images").mkdir(parents=True)
    images, annots = [], []
    for i, (h, w) in enumerate([(768, 1024), (600, 900), (700, 1366)]):
        name = f"frame{i}.png"
        Image.fromarray(synth.synthetic_crowd_frame(20 + i, max(h, w), 60)[:h, :w]).save(root / "Images" / name)
        images.append({"id": i + 1, "file_name": name, "width": w, "height": h})
        annots += [{"id": len(annots) + j, "image_id": i + 1, "bbox": [10 + 40 * j, 20, 30, 60], "category_id": 1} for j in range(3)]
    (root / "val.json").write_text(json.dumps({"images": images, "annotations": annots, "categories": [{"id": 1, "name": "person"}]}))
    cfg = {"environ": {"seed": 42, "device": "cuda", "output_dir": str(tmp_path / "out")},
           "data": {"dataset": "crowdhuman", "dataset_root": str(root), "json_file": str(root / "val.json")},
           "model": {"dino_repo": "./dinov2", "dino_model": "dinov2_vitl14", "dino_checkpoint": str(tmp_path / "dino.pth"),
                     "dino_depth": 2, "sam_checkpoint": str(tmp_path / "sam.pth"), "sam_model": arch, "sam_arch": "crowdsam",
                     "sam_adapter_checkpoint": str(tmp_path / "adapter.pth"), "n_class": 1, "max_size": 1024, "trainfree": False},
           "test": dict(output_rles=True, crop_n_layers=0, crop_nms_thresh=0.7, crop_overlap_ratio=0.341, pos_sim_thresh=-1.0,
                        apply_box_offsets=False, grid_size=8, max_prompts=64, filter_thresh=0.7, points_per_batch=32,
                        mask_selection="max_iou", max_size=1024, fuse_simmap=False, min_mask_region_area=100,
                        box_nms_thresh=0.65, stability_score_thresh=0.0, stability_score_offset=1, pred_iou_thresh=0.1),
           "vis": {"vis_thresh": 0.0}}
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    out = tmp_path / "res.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "test.py"), "-c", str(tmp_path / "cfg.yaml"), "-s", str(out),
                        "-v", "test.grid_size", "6"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert [x["image_id"] for x in res] == [1, 2, 3] and all(x["num_gt"] == 2 for x in res)
    for x, im in zip(res, images):
        assert len(x["boxes"]) == len(x["scores"]) == len(x["categories"]) == len(x["rles"])
        b = np.array(x["boxes"]).reshape(-1, 4)
        assert (b >= 0).all() and (b[:, [0, 2]] <= im["width"]).all() and (b[:, [1, 3]] <= im["height"]).all()
    assert sum(len(x["boxes"]) for x in res) > 0
    assert os.path.exists(tmp_path / "out" / "0.jpg")            # -v: visualisation written per image

    # --profile (build extension): same detections, plus the per-rank timings record and roctx ranges
    # (no profiler attached here: the calls go to the roctx library and nowhere else)
    out2 = tmp_path / "res_profile.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "test.py"), "-c", str(tmp_path / "cfg.yaml"), "-s", str(out2),
                        "--profile", "test.grid_size", "6"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert json.load(open(out2)) == res
    t = json.load(open(tmp_path / "out" / "timings_rank0.json"))
    assert t["rank"] == 0 and t["images"] == 3 and t["kept_masks"] == sum(len(x["boxes"]) for x in res)
    for stage in ("set_image", "sample_prompts", "eps_sweep"):
        assert t["stage_ms_total"][stage] > 0 and t["stage_ms_per_image"][stage] == pytest.approx(t["stage_ms_total"][stage] / 3)


def test_batch_eval_gpu_worker_streams_its_shard(tmp_path, cuda):
    """tools/batch_eval.py's GPU worker (reference: tools/batch_eval.py:76-103, one process per GPU) on one rank: the shard goes
    through CrowdSAM.generate_stream, the detections equal tools/test.py's for the same synthetic frames, and --profile leaves the
    per-rank timings record."""
    cfg = {"environ": {"seed": 42, "device": "cuda", "output_dir": str(tmp_path / "out")},
           "data": {"dataset": "crowdhuman", "dataset_root": "", "json_file": ""},
           "model": {"dino_repo": "./dinov2", "dino_model": "dinov2_vitl14", "dino_checkpoint": "", "sam_checkpoint": "",
                     "sam_model": "vit_test128", "sam_arch": "crowdsam", "sam_adapter_checkpoint": "", "n_class": 1,
                     "max_size": 1024, "trainfree": False},
           "test": dict(output_rles=True, crop_n_layers=0, crop_nms_thresh=0.7, crop_overlap_ratio=0.341, pos_sim_thresh=-1.0,
                        apply_box_offsets=False, grid_size=8, max_prompts=64, filter_thresh=0.7, points_per_batch=32,
                        mask_selection="max_iou", max_size=1024, fuse_simmap=False, min_mask_region_area=100,
                        box_nms_thresh=0.65, stability_score_thresh=0.0, stability_score_offset=1, pred_iou_thresh=0.1),
           "vis": {"vis_thresh": 0.0}}
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    det = tmp_path / "det.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "batch_eval.py"), "-n", "1", "-c", str(tmp_path / "cfg.yaml"),
                        "--synthetic", "5", "--profile", "-o", str(det)], capture_output=True, text=True, timeout=900,
                       cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = tmp_path / "res.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "test.py"), "-c", str(tmp_path / "cfg.yaml"), "--synthetic", "5",
                        "-s", str(res)], capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    per_image = json.load(open(res))
    coco = json.load(open(det))
    dets = coco["annotations"] if isinstance(coco, dict) and "annotations" in coco else coco
    assert len(dets) == sum(len(x["boxes"]) for x in per_image) > 0
    assert sorted(d["score"] for d in dets) == sorted(float(np.float32(v)) for x in per_image for v in x["scores"])
    t = json.load(open(tmp_path / "out" / "timings_rank0.json"))
    assert t["images"] == 5 and t["kept_masks"] == len(dets) and t["stage_ms_total"]["eps_sweep"] > 0
