"""ORACLE-side CPU baseline (test infrastructure): time the fp32 PyTorch-CPU restatement of the path on
the host cores of the GPU box, on a BOUNDED sample, and extrapolate to the benchmark workload.

Only bench.py's ``cpu_baseline`` leg runs this (as a subprocess with a hard time limit).  It is a reported
baseline, not the product.  Sample: SAM encoder = 1 windowed + 1 global block timed and scaled to the
architecture's block counts; DINOv2 = 1 of 24 blocks; decoder + post-processing = a few prompts.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def measure(arch="vit_l", n_prompts_full=4096, sample_prompts=4, log=None):
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle import sam_oracle as so
    say = (lambda *a: print(*a, file=sys.stderr, flush=True)) if log is None else log
    D, depth, heads, gidx = synth.SAM_CONFIGS[arch]
    threads = torch.get_num_threads()
    n_glob = len(gidx)
    # encoder: depth-2 sample (block 0 windowed, block 1 global) + patch embed + neck
    specs = [s for s in synth.sam_param_specs(D, 2, heads, (1,)) if not s[0].startswith("image_encoder.blocks.")
             or s[0].startswith("image_encoder.blocks.0.") or s[0].startswith("image_encoder.blocks.1.")]
    sd = synth.make_state_dict(specs, 0)
    image = synth.synthetic_crowd_frame(0, 1024, 150)
    img = torch.from_numpy(image).permute(2, 0, 1).float().contiguous()
    with torch.no_grad():
        x = so.preprocess(img)[None]
        t0 = time.time()
        tok = torch.nn.functional.conv2d(x, sd["image_encoder.patch_embed.proj.weight"],
                                         sd["image_encoder.patch_embed.proj.bias"], stride=16).permute(0, 2, 3, 1)
        tok = tok + sd["image_encoder.pos_embed"]
        t_embed = time.time() - t0
        t0 = time.time()
        tok = so.encoder_block(sd, "image_encoder.blocks.0.", tok, heads, 14)
        t_win = time.time() - t0
        t0 = time.time()
        tok = so.encoder_block(sd, "image_encoder.blocks.1.", tok, heads, 0)
        t_glob = time.time() - t0
        t0 = time.time()
        f = tok.permute(0, 3, 1, 2)
        f = torch.nn.functional.conv2d(f, sd["image_encoder.neck.0.weight"])
        f = so.layer_norm_2d(sd, "image_encoder.neck.1", f)
        f = torch.nn.functional.conv2d(f, sd["image_encoder.neck.2.weight"], padding=1)
        feat = so.layer_norm_2d(sd, "image_encoder.neck.3", f)
        t_neck = time.time() - t0
        t_enc = t_embed + t_win * (depth - n_glob) + t_glob * n_glob + t_neck
        say(f"cpu_baseline: encoder win {t_win:.2f}s glob {t_glob:.2f}s -> {t_enc:.1f}s")
        dsd = synth.make_state_dict(synth.dino_param_specs(1024, 1), 1)
        xd = torch.nn.functional.interpolate(x, (1022, 1022), mode="bilinear")
        t0 = time.time()
        so.dinov2_forward(dsd, xd, depth=1)
        t_dino = (time.time() - t0) * 24.0
        say(f"cpu_baseline: dino {t_dino:.1f}s")
        dino_feats = torch.from_numpy(np.random.RandomState(1).standard_normal((1, 73, 73, 1024)).astype(np.float32))
        pe = so.dense_pe(sd)
        pts = torch.from_numpy(np.random.RandomState(2).randint(0, 1024, size=(sample_prompts, 1, 2)).astype(np.float64))
        t0 = time.time()
        sparse = so.embed_points(sd, pts, torch.ones(sample_prompts, 1, dtype=torch.int))
        low, iou, cls = so.mask_decoder(sd, feat, pe, sparse, dino_feats)
        masks = so.postprocess_masks(low, image.shape[:2], image.shape[:2])
        s = torch.clamp(iou, 0.) * cls.squeeze(2).sigmoid()
        sel = masks[torch.arange(sample_prompts), s.max(-1)[1]]
        po.calculate_stability_score(sel, 0.0, 1.0)
        po.batched_mask_to_box(sel > 0)
        t_dec = (time.time() - t0) / sample_prompts
        say(f"cpu_baseline: decoder {t_dec:.3f}s/prompt")
    total = t_enc + t_dino + n_prompts_full * t_dec
    return dict(value=1.0 / total, unit="images/s", cores=threads, kind="port",
                sample=(f"1 image on {threads} torch threads: SAM {arch} encoder from 1 windowed ({t_win:.2f}s) + 1 global "
                        f"({t_glob:.2f}s) block scaled to {depth - n_glob}+{n_glob} blocks = {t_enc:.1f}s; DINOv2-L 1/24 blocks "
                        f"scaled = {t_dino:.1f}s; {sample_prompts} prompts decoded+post-processed = {t_dec:.3f}s/prompt, "
                        f"extrapolated to {n_prompts_full} prompts"))


if __name__ == "__main__":
    arch = sys.argv[1] if len(sys.argv) > 1 else "vit_l"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    print(json.dumps(measure(arch, n)), flush=True)
