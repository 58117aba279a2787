"""debug: candidates / NaNs of the first vs later generate() calls at full composition"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_full_composition_gpu import _model
from oracle.make_goldens import full_frame
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_l"
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = _model(arch, grid, 4096)
img = full_frame(2 if arch == "vit_l" else 4)
p = m.predictor
reset = p.reset_image
def grab():
    if p.is_image_set:
        f = p._feat_tok
        d = p._dtok16
        fg = p._plan.fg_logits()
        print("   feat finite", bool(torch.isfinite(f).all()), float(f.abs().mean()), "| dino finite", bool(torch.isfinite(d).all()),
              float(d.float().abs().mean()), "| fg finite", bool(torch.isfinite(fg).all()), float(fg.abs().mean()),
              "| sim finite", bool(torch.isfinite(m.sim_map).all()))
    reset()
p.reset_image = grab
for i in range(3):
    np.random.seed(0)
    out = m.generate(img)
    print("call", i, "candidates", m.last_candidates, "kept", len(out["boxes"]))
