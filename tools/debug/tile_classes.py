"""developer: how many 64x64 tiles of the crowded frame's kept masks are empty / full / mixed (per CC pass)."""
import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
         filter_thresh=float("inf"), max_prompts=4096)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(3)]
m.box_nms_thresh = m.crop_nms_thresh = 1.0
m.generate(frames[0])
sc = np.sort(m._store["score"][:m.last_candidates].float().cpu().numpy())[::-1]
m.pred_iou_thresh = float(sc[min(700, len(sc) - 1)])
from crowdsam_amd import hip
orig = hip.small_regions_idx
def spy(store, idx, min_area, out_store=None):
    mk = store[idx.long()]                                  # [n, H, W] uint8 (BEFORE the clean-up)
    n, H, W = mk.shape
    tl = mk.view(n, H // 64, 64, W // 64, 64).permute(0, 1, 3, 2, 4).reshape(n, -1, 4096)
    s_ = tl.sum(-1)
    empty, full = (s_ == 0).float().mean().item(), (s_ == 4096).float().mean().item()
    area = mk.view(n, -1).float().mean(1)
    print("masks %d  tiles: empty %.3f  full %.3f  mixed %.3f | mask area fraction mean %.3f min %.3f max %.3f"
          % (n, empty, full, 1 - empty - full, area.mean().item(), area.min().item(), area.max().item()))
    return orig(store, idx, min_area, out_store)
hip.small_regions_idx = spy
import crowdsam.model as cm
for f in frames[1:]:
    m.generate(f)
