"""Microbenchmark: csam_small_regions on blob-like and noise masks (device) vs the host scipy path."""
import time
import numpy as np
import torch
from crowdsam_amd import hip
from segment_anything_cs.utils.amg import remove_small_regions

dev = torch.device("cuda:0")
H, W = 683, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(100, 1, H, W, generator=g)
blobs = (torch.nn.functional.avg_pool2d(x, 31, 1, 15)[:, 0] > 0.02)
noise = torch.rand(4, H, W, generator=g) > 0.5
lo = torch.nn.functional.avg_pool2d(torch.randn(100, 1, 171, 256, generator=g), 9, 1, 4)
smooth = torch.nn.functional.interpolate(lo, (H, W), mode="bilinear", align_corners=False)[:, 0] > 0.05
for name, m in (("smooth100", smooth), ("blobs100", blobs), ("noise4", noise), ("noise1", noise[:1])):
    md = m.to(dev)
    for _ in range(2):
        out = hip.small_regions(md, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = hip.small_regions(md, 100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    k = min(len(m), 4)
    for a in m[:k].numpy():
        a, _ = remove_small_regions(a, 100, "holes")
        remove_small_regions(a, 100, "islands")
    th = (time.perf_counter() - t0) / k * 1e3
    print(f"{name}: device {dt:.3f} ms for {len(m)} masks ({dt / len(m) * 1e3:.1f} us/mask); host {th:.2f} ms/mask", flush=True)
