"""Developer timing of the crowded-tail leg of bench.py (box NMS off, ~700 candidates, ~320 kept masks): stage split with
per-stage syncs, then a cProfile of the host side without them."""
import sys, os
os.environ["CSAM_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats, time
import numpy as np, torch
import crowdsam.model as cm
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
         filter_thresh=float("inf"), max_prompts=4096)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(8)]
m.box_nms_thresh = m.crop_nms_thresh = 1.0
m.generate(frames[0])
sc = np.sort(m._store["score"][:m.last_candidates].float().cpu().numpy())[::-1]
m.pred_iou_thresh = float(sc[min(700, len(sc) - 1)])
for f in frames[:3]: m.generate(f)
m.timings = {}
kept = 0
for f in frames[3:]: kept += len(m.generate(f)["boxes"])
torch.cuda.synchronize()
print("with per-stage syncs:", {k: round(v / 5, 2) for k, v in m.timings.items()}, "candidates", m.last_candidates, "kept/img", kept / 5)
cm._TIMING = False
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in frames[3:]: m.generate(f)
torch.cuda.synchronize()
print("ms/img", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for f in frames[3:]: m.generate(f)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumtime").print_stats(28)
