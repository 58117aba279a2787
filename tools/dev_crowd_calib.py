"""Developer probe for bench.py's crowded-frame constants: for the bench's seeded frames, the predicted-IoU cut that lets
~720 candidates through and what box NMS does to the survivors at thresholds between 0.65 and 1.0 (so that the timed
frame can keep ~300 masks WITH the NMS suppression path active)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crowdsam.model import CrowdSAM
from crowdsam.utils import DEFAULT_TEST_CONFIG
from crowdsam_amd import hip, synth

t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
         filter_thresh=float("inf"), max_prompts=4096)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
m.box_nms_thresh = m.crop_nms_thresh = 1.0
m.pred_iou_thresh = 0.0
for fi in (0, 1, 2, 7, 1000, 2003):
    f = synth.synthetic_crowd_frame(fi, 1024, 150)
    m.pred_iou_thresh = 0.0
    m.generate(f)
    n = m.last_candidates
    sc = m._store["score"][:n].float().cpu().numpy()
    srt = np.sort(sc)[::-1]
    print("frame %d: %d candidates past stability at cut 0; score quantiles" % (fi, n),
          [round(float(srt[min(k, n - 1)]), 4) for k in (100, 200, 330, 500, 720, 1000, 1500)])
    for cut in (0.85, 0.87, 0.889, 0.9):
        keep = sc > cut
        b = m._store["boxes"][:n][torch.as_tensor(keep).cuda()].float()
        s = torch.as_tensor(sc[keep]).cuda()
        row = []
        for thr in (0.65, 0.8, 0.9, 0.95, 0.97, 0.98, 0.99, 1.0):
            row.append((thr, int(len(hip.box_nms(b, s, thr)))))
        print("   cut %.3f: %d candidates -> kept by box NMS at threshold:" % (cut, int(keep.sum())), row)
