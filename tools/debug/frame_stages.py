"""Developer: per-stage times (device-synchronised, CSAM_TIMING) of single frames of the crowded bench -- which stage of which
frame stalls.   python tools/debug/frame_stages.py 9 10 11 12"""
import sys, os
os.environ["CSAM_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG
ids = [int(a) for a in sys.argv[1:]] or [9, 10, 11, 12]
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
         filter_thresh=float("inf"), max_prompts=4096)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
m.box_nms_thresh = m.crop_nms_thresh = 1.0
m.pred_iou_thresh = 0.8890                      # bench.py CROWD_FROZEN
for i in (0, 1):
    m.generate(synth.synthetic_crowd_frame(i, 1024, 150))
for i in ids:
    f = synth.synthetic_crowd_frame(i, 1024, 150)
    m.timings = {}
    out = m.generate(f)
    torch.cuda.synchronize()
    runs = sum(len(r["counts"]) for r in out["rles"]) if "rles" in out._stats else -1
    print("frame %2d: kept %3d candidates %4d rle runs %9d | " % (i, len(out["boxes"]), m.last_candidates, runs)
          + " ".join("%s %.1f" % (k, v) for k, v in m.timings.items()))
