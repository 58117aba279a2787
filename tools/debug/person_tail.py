"""developer: stage times of the tail (small regions, NMS, RLE, strings) on person-shaped masks, windowed vs full-frame clean-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam_amd import hip
from segment_anything_cs.utils.amg import coco_encode_rles, mask_to_rle_arrays
dev = torch.device("cuda:0")
n, H, W = 330, 1024, 1024
rs = np.random.RandomState(7)
store = torch.zeros(n, H, W, dtype=torch.uint8, device=dev)
yy = torch.arange(H, device=dev, dtype=torch.float32)[:, None]
xx = torch.arange(W, device=dev, dtype=torch.float32)[None, :]
for i in range(n):
    cy, cx = rs.uniform(100, H - 100), rs.uniform(50, W - 50)
    ay, ax = rs.uniform(50, 130), rs.uniform(18, 50)
    m = ((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1.0
    for _ in range(3):
        hy, hx = int(cy + rs.uniform(-0.5, 0.5) * ay), int(cx + rs.uniform(-0.4, 0.4) * ax)
        m[hy:hy + 3, hx:hx + 3] = False
    store[i] = m
ref = store.clone()
ra, ca = ref.any(2), ref.any(1)
ah, aw = torch.arange(H, device=dev), torch.arange(W, device=dev)
boxes = torch.stack([torch.where(ca, aw, 1 << 30).amin(1), torch.where(ra, ah, 1 << 30).amin(1), torch.where(ca, aw, -1).amax(1),
                     torch.where(ra, ah, -1).amax(1)], 1).long()
idx = torch.arange(n, dtype=torch.int32, device=dev)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("restore copy            %.3f ms" % t(lambda: store.copy_(ref)))
print("small regions full      %.3f ms (incl. restore)" % t(lambda: (store.copy_(ref), hip.small_regions_idx(store, idx, 100))))
print("small regions windowed  %.3f ms (incl. restore)" % t(lambda: (store.copy_(ref), hip.small_regions_windowed(store, idx, boxes, 100))))
store.copy_(ref)
rl = None
def rle():
    global rl
    rl = mask_to_rle_arrays(store, idx=idx)
print("RLE arrays (count, write, D2H, numpy) %.3f ms" % t(rle))
print("COCO strings            %.3f ms" % t(lambda: coco_encode_rles(rl)))
