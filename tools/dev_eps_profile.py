"""Developer: cProfile of CrowdSAM.generate in the shipped EPS configuration (where does the HOST time of a round go?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats, time
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG
GRID = int(sys.argv[1]) if len(sys.argv) > 1 else 192
t = dict(DEFAULT_TEST_CONFIG); t.update(grid_size=GRID, stability_score_thresh=0.25)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(0)
frames = [synth.synthetic_crowd_frame(i) for i in range(8)]
for f in frames[:3]: m.generate(f)
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
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumtime").print_stats(30)
