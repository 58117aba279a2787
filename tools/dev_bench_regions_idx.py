"""Microbenchmark: csam_small_regions_idx (the compact ring-forest form the driver uses, in place on store slots) on
person-like masks at 1024 x 1024 (ellipses of a few percent of the frame with holes and specks: the shape real Crowd-SAM
masks have) and on the noise-like masks random weights produce in bench.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crowdsam_amd import hip
dev = torch.device("cuda:0")
H = W = 1024
n = 320
rng = np.random.RandomState(0)
yy, xx = np.mgrid[:H, :W]
people = np.zeros((n, H, W), np.uint8)
for i in range(n):
    cy, cx = rng.randint(150, H - 150), rng.randint(60, W - 60)
    ry, rx = rng.randint(60, 160), rng.randint(25, 70)
    m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
    for _ in range(6):                                              # holes and specks
        y, x = rng.randint(0, H - 8), rng.randint(0, W - 8)
        m[y:y + rng.randint(1, 8), x:x + rng.randint(1, 8)] ^= True
    people[i] = m
g = torch.Generator().manual_seed(0)
noise = (torch.rand(n, H, W, generator=g) > 0.13).to(torch.uint8).numpy()
for name, masks in (("people", people), ("noise(bench-like)", noise)):
    store = torch.as_tensor(masks).to(dev)
    keep = store.clone()
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    tiles = store.view(n, 16, 64, 16, 64).permute(0, 1, 3, 2, 4).reshape(n, 256, 4096).sum(-1)
    frac = ((tiles == 0) | (tiles == 4096)).float().mean().item()
    for _ in range(2):
        store.copy_(keep); hip.small_regions_idx(store, idx, 100)
    ts = []
    for _ in range(5):
        store.copy_(keep); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.small_regions_idx(store, idx, 100); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%-18s %d masks, trivial tiles %.3f: %.3f ms (%.1f us/mask)" % (name, n, frac, min(ts), min(ts) / n * 1e3), flush=True)
