"""Developer timing of the headline dense mode: stage split per image (CSAM_TIMING adds a device sync per stage) and a
cProfile of the host side without those syncs."""
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
np.random.seed(0)
frames = [synth.synthetic_crowd_frame(i) for i in range(8)]
for f in frames[:3]: m.generate(f)
m.timings = {}
for f in frames[3:]: m.generate(f)
torch.cuda.synchronize()
print("with per-stage syncs:", {k: round(v / 5, 2) for k, v in m.timings.items()}, "candidates", m.last_candidates)
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
st.sort_stats("tottime").print_stats(25)
