"""ORACLE-side CPU baseline (test infrastructure): time the fp32 PyTorch-CPU restatement of the path on
the host cores of the GPU box, on a BOUNDED sample, and extrapolate to the benchmark workload.

Only bench.py's ``cpu_baseline`` leg calls this.  It is a reported baseline, not the product.
"""
import os
import time

import numpy as np
import torch

from . import pipeline_oracle as po
from . import sam_oracle as so


def measure(sam_sd, dino_sd, arch_cfg, image, n_prompts_full=4096, sample_prompts=8, dino_blocks_sample=2,
            threads=None):
    """Returns dict(value=images/s extrapolated, cores, sample description, stage seconds)."""
    depth, heads, gidx = arch_cfg
    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    img = torch.from_numpy(image).permute(2, 0, 1).float().contiguous()
    with torch.no_grad():
        x = so.preprocess(img)[None]
        t0 = time.time()
        feat = so.image_encoder(sam_sd, x, depth, heads, gidx)
        t_enc = time.time() - t0
        xd = torch.nn.functional.interpolate(x, (1022, 1022), mode="bilinear")
        t0 = time.time()
        so.dinov2_forward(dino_sd, xd, depth=dino_blocks_sample)
        t_dino = (time.time() - t0) * 24.0 / dino_blocks_sample       # linear in depth
        dino_feats = torch.from_numpy(np.random.RandomState(1).standard_normal((1, 73, 73, 1024)).astype(np.float32))
        pe = so.dense_pe(sam_sd)
        pts = torch.from_numpy(np.random.RandomState(2).randint(0, 1024, size=(sample_prompts, 1, 2)).astype(np.float64))
        t0 = time.time()
        sparse = so.embed_points(sam_sd, pts, torch.ones(sample_prompts, 1, dtype=torch.int))
        low, iou, cls = so.mask_decoder(sam_sd, feat, pe, sparse, dino_feats)
        masks = so.postprocess_masks(low, image.shape[:2], image.shape[:2])
        s = torch.clamp(iou, 0.) * cls.squeeze(2).sigmoid()
        ind = s.max(-1)[1]
        sel = masks[torch.arange(sample_prompts), ind]
        po.calculate_stability_score(sel, 0.0, 1.0)
        po.batched_mask_to_box(sel > 0)
        t_dec = (time.time() - t0) / sample_prompts
    total = t_enc + t_dino + n_prompts_full * t_dec
    return dict(value=1.0 / total, unit="images/s", cores=threads, kind="port",
                sample=(f"1 image: SAM encoder full ({t_enc:.1f}s) + DINOv2 {dino_blocks_sample}/24 blocks scaled "
                        f"({t_dino:.1f}s) + {sample_prompts} prompts decoded+post-processed ({t_dec:.3f}s/prompt) "
                        f"extrapolated to {n_prompts_full} prompts"),
                seconds=dict(encoder=t_enc, dino=t_dino, per_prompt=t_dec))
