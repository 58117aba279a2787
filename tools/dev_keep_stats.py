import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG as DEFAULT_TEST_CFG
t = dict(DEFAULT_TEST_CFG); t.update(grid_size=64, points_per_batch=256, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"),
                                     max_prompts=4096, stability_score_thresh=0.0, pred_iou_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0,
                                     min_mask_region_area=0)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
m.output_rles = False
import segment_anything_cs.utils.amg as amg
np.random.seed(0)
# avoid RLE of 4096 masks: monkeypatch
import crowdsam.model as cm
cm.mask_to_rle_arrays = lambda masks: [ {"size": list(masks.shape[1:]), "counts": [int(masks.shape[1]*masks.shape[2])]} for _ in range(masks.shape[0]) ]
out = m.generate(synth.synthetic_crowd_frame(0))
s, st, b = out["scores"], out["stability_score"], out["boxes"]
print("n", len(s), "score pct", np.percentile(s, [1, 10, 50, 90, 99]))
print("stability pct", np.percentile(st, [1, 10, 25, 50, 75, 90, 99]))
w = (b[:, 2] - b[:, 0]); h = (b[:, 3] - b[:, 1]); print("box w pct", np.percentile(w, [1, 50, 99]), "h", np.percentile(h, [1, 50, 99]))
