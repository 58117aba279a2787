import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_full_composition_gpu import _model, _invariants
from oracle.make_goldens import full_frame
m = _model("vit_l", 64, 4096)
try:
    _invariants(m, full_frame(2), torch.device("cuda:0"), 1024, 64)
    print("invariants OK", m.last_candidates)
except AssertionError as e:
    print("ASSERT", str(e)[:300], "last_candidates", m.last_candidates)
