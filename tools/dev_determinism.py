"""Developer check: run the same frame twice and compare the candidate store (scores / boxes keyed by prompt point)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam.utils import DEFAULT_TEST_CONFIG
from crowdsam_amd import synth

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ARCH = "vit_test128"
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=grid, points_per_batch=2048, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"),
         max_prompts=grid * grid, stability_score_thresh=0.25, pred_iou_thresh=0.05)
cfg = {"environ": {"device": "cuda"}, "model": {"sam_model": ARCH, "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict(ARCH), dino_state_dict=synth.make_dino_state_dict(depth=1), dino_depth=1)
img = synth.synthetic_crowd_frame(7, 1024, 150)
runs = []
for r in range(3):
    np.random.seed(0)
    out = m.generate(img)
    n = m.last_candidates
    st = m._store
    pts = st["points"][:n].cpu().numpy(); sc = st["score"][:n].cpu().numpy(); bx = st["boxes"][:n].cpu().numpy()
    key = pts[:, 0].astype(np.int64) * 100000 + pts[:, 1]
    o = np.argsort(key, kind="stable")
    runs.append((key[o], sc[o], bx[o], out["scores"].copy()))
    print("run", r, "candidates", n, "kept", len(out["scores"]), out["scores"][:3])
for r in (1, 2):
    same_keys = np.array_equal(runs[0][0], runs[r][0])
    print("run0 vs run%d: same prompt set %s" % (r, same_keys), end=" ")
    if same_keys:
        d = np.abs(runs[0][1] - runs[r][1])
        print("score diffs: n!=0", int((d != 0).sum()), "max", d.max(), "box diffs", int((runs[0][2] != runs[r][2]).any(1).sum()))
    else:
        print(len(runs[0][0]), len(runs[r][0]))
