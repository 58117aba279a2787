"""Developer: the dense-sweep frame with the blob-mask weight set (synth.blob_heads) at the SHIPPED thresholds: how many candidates
pass the stability / score filters, how many survive box NMS and the small-region pass, what the masks look like, how long it takes."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crowdsam.model import CrowdSAM
from crowdsam.utils import DEFAULT_TEST_CONFIG
from crowdsam_amd import synth

kw = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"), max_prompts=4096)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
sd = synth.blob_heads(synth.make_sam_state_dict("vit_l"), **kw)
m = CrowdSAM(cfg, sam_state_dict=sd, dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(6)]
for f in frames[:2]:
    out = m.generate(f)
torch.cuda.synchronize()
n = m.last_candidates
st = m._store
print("candidates after the per-batch filters (pred_iou %.2f, stability %.2f): %d of 4096" % (m.pred_iou_thresh, m.stability_score_thresh, n))
if n:
    sc, stab = st["score"][:n].float().cpu().numpy(), st["stability"][:n].float().cpu().numpy()
    bx = st["boxes"][:n].float().cpu().numpy()
    print("  score quantiles", np.round(np.quantile(sc, [0, .1, .5, .9, 1]), 3), "stability quantiles", np.round(np.quantile(stab, [0, .1, .5, .9, 1]), 3))
    print("  box width quantiles", np.quantile(bx[:, 2] - bx[:, 0], [0, .1, .5, .9, 1]), "height", np.quantile(bx[:, 3] - bx[:, 1], [0, .1, .5, .9, 1]))
print("kept after box NMS %.2f + small regions: %d; box w/h medians %.0f / %.0f" % (m.box_nms_thresh, len(out["boxes"]),
      np.median(out["boxes"][:, 2] - out["boxes"][:, 0]) if len(out["boxes"]) else 0, np.median(out["boxes"][:, 3] - out["boxes"][:, 1]) if len(out["boxes"]) else 0))
torch.cuda.synchronize()
t0 = time.perf_counter()
kept = 0
for out in m.generate_stream(frames[2:] * 3, batch=4):
    kept += len(out["boxes"])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("12 frames through generate_stream(batch=4): %.2f ms/frame, %.1f kept masks per frame" % (1e3 * dt / 12, kept / 12))
