import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from oracle import pipeline_oracle as po
from oracle.make_goldens import PIPE_CFG, StandInDino, pipeline_image
from tests.test_pipeline_gpu import _config, GpuStandInDino, ARCH
cuda = torch.device("cuda:0")
cfg = dict(PIPE_CFG)
cfg.update(max_size=int(sys.argv[1]) if len(sys.argv) > 1 else 1536, max_prompts=16, min_mask_region_area=0)
img = pipeline_image()
m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
np.random.seed(1)
out = m.generate(img)
D, depth, heads, gidx = synth.SAM_CONFIGS[ARCH]
np.random.seed(1)
o = po.OracleCrowdSAM(synth.make_sam_state_dict(ARCH), (depth, heads, gidx), StandInDino(), cfg, rng=np.random)
with torch.no_grad():
    ref = o.generate(img)
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for k in ("points", "scores", "boxes", "stability_score"):
    print(k, "HIP\n", out[k], "\nORACLE\n", ref[k])
print("cfg", {k: cfg[k] for k in ("pred_iou_thresh", "stability_score_thresh", "box_nms_thresh", "filter_thresh", "grid_size", "points_per_batch", "pos_sim_thresh")})
