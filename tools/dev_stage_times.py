import sys, os
os.environ["CSAM_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG as DEFAULT_TEST_CFG
t = dict(DEFAULT_TEST_CFG); t.update(grid_size=64, points_per_batch=256, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"),
                                     max_prompts=4096, stability_score_thresh=float(sys.argv[1]) if len(sys.argv) > 1 else 0.25)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(0)
frames = [synth.synthetic_crowd_frame(i) for i in range(4)]
for f in frames[:2]: m.generate(f)
m.timings = {}
t0 = time.perf_counter()
for f in frames[2:]:
    out = m.generate(f)
    t1 = time.perf_counter(); enc = [__import__("segment_anything_cs.utils.amg", fromlist=["x"]).coco_encode_rle] 
print("total ms/img", (time.perf_counter() - t0) / 2 * 1e3, "candidates", m.last_candidates)
print({k: round(v / 2, 2) for k, v in m.timings.items()})
