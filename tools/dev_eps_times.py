"""Developer timing of the shipped EPS mode (32 prompts per batch, one sync per batch): stage split per image."""
import sys, os
os.environ["CSAM_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG
GRID = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t = dict(DEFAULT_TEST_CONFIG); t.update(grid_size=GRID, stability_score_thresh=0.25)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(0)
frames = [synth.synthetic_crowd_frame(i) for i in range(6)]
for f in frames[:2]: m.generate(f)
m.timings = {}
t0 = time.perf_counter()
for f in frames[2:]:
    out = m.generate(f)
torch.cuda.synchronize()
n = len(frames) - 2
print("grid", GRID, "total ms/img", (time.perf_counter() - t0) / n * 1e3, "candidates", m.last_candidates)
print({k: round(v / n, 2) for k, v in m.timings.items()})
