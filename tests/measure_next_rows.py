"""Measurement for the SURVEY 8(f) rows: device kernel vs the CPU oracle on the same inputs (JSON on stdout).
    python tests/measure_next_rows.py > profiles/r01_next_rows.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crowdsam_amd import evaluate as ev
from crowdsam_amd import hip
from oracle import eval_oracle as eo
from oracle import pipeline_oracle as po

dev = torch.device("cuda:0")
res = {}


def gpu_ms(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# f2: small-region clean-up on 100 smooth 683x1024 masks
g = torch.Generator().manual_seed(0)
lo = torch.nn.functional.avg_pool2d(torch.randn(100, 1, 171, 256, generator=g), 9, 1, 4)
masks = torch.nn.functional.interpolate(lo, (683, 1024), mode="bilinear", align_corners=False)[:, 0] > 0.05
md = masks.to(dev)
t_dev = gpu_ms(lambda: hip.small_regions(md, 100))
t0 = time.perf_counter()
for m in masks[:4].numpy():
    a, _ = po.remove_small_regions(m, 100, "holes")
    po.remove_small_regions(a, 100, "islands")
t_cpu = (time.perf_counter() - t0) / 4 * 1e3
res["small_regions"] = {"workload": "100 masks 683x1024, holes + islands < 100 px", "device_ms": t_dev,
                        "device_us_per_mask": t_dev * 10, "cpu_oracle_ms_per_mask": t_cpu, "cpu_cores": 1}

# f3: mask coverage NMS on 1500 ellipse masks
rs = np.random.RandomState(1)
n, H, W = 1500, 683, 1024
yy, xx = np.mgrid[0:H, 0:W]
mk = np.zeros((n, H, W), bool)
for i in range(n):
    mk[i] = ((yy - rs.uniform(0, H)) / rs.uniform(8, 80)) ** 2 + ((xx - rs.uniform(0, W)) / rs.uniform(8, 80)) ** 2 <= 1
sc = torch.from_numpy(rs.permutation(n).astype(np.float32))
mkd, scd = torch.from_numpy(mk).to(dev), sc.to(dev)
t_dev = gpu_ms(lambda: hip.mask_nms(mkd, scd, 0.5))
t0 = time.perf_counter()
po.mask_iou_nms(None, sc.numpy()[:300], torch.from_numpy(mk[:300]), 0.5)
t_cpu = (time.perf_counter() - t0) * 1e3
res["mask_nms"] = {"workload": "1500 masks 683x1024 -> 150x150 bits, coverage > 0.5", "device_ms": t_dev,
                   "cpu_oracle_ms_300_masks": t_cpu, "cpu_cores": os.cpu_count()}

# f1: Caltech matching, 4370 images x (120 detections, 60 GT)
recs = []
for i in range(4370):
    r = ev.ImageRecord(i, 1600, 1200)
    ng, nd = int(rs.randint(20, 100)), int(rs.randint(60, 180))
    xy, wh = rs.uniform(0, 1400, (ng, 2)), rs.uniform(20, 200, (ng, 2))
    r.gt = np.concatenate([xy, xy + wh, np.where(rs.rand(ng) < 0.1, -1.0, 1.0)[:, None]], 1)
    dxy, dwh = rs.uniform(0, 1400, (nd, 2)), rs.uniform(20, 200, (nd, 2))
    r.dt = np.concatenate([dxy, dxy + dwh, rs.rand(nd, 1)], 1)
    recs.append(r)
t0 = time.perf_counter()
ev.match(recs, 0.5, dev)
t_dev = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
for r in recs[:200]:
    eo.compare_caltech(r.dt, r.gt, 0.5)
t_cpu = (time.perf_counter() - t0) / 200 * 4370 * 1e3
res["caltech_match"] = {"workload": "4370 images (CrowdHuman val size), ~120 detections x ~60 GT each",
                        "device_ms_incl_host_packing": t_dev, "cpu_oracle_ms_extrapolated": t_cpu, "cpu_cores": 1}

# f4: fuse_simmap scores for 500 masks
sim = torch.rand(64, 64, device=dev)
t_dev = gpu_ms(lambda: hip.mask_mean_bilinear(mkd[:500], sim[:43, :64]))
t0 = time.perf_counter()
po.fuse_simmap_scores(torch.from_numpy(mk[:50]), torch.rand(50), sim.cpu()[:43, :64], (H, W))
t_cpu = (time.perf_counter() - t0) * 10 * 1e3
res["fuse_simmap"] = {"workload": "500 masks 683x1024, prior 43x64", "device_ms": t_dev,
                      "cpu_oracle_ms_extrapolated": t_cpu, "cpu_cores": os.cpu_count()}
print(json.dumps(res, indent=1))
