"""ORACLE-side CPU baseline (test infrastructure): time the fp32 PyTorch-CPU restatement of the path on
the host cores of the GPU box, on a BOUNDED sample, and extrapolate to the benchmark workload.

Only bench.py's ``cpu_baseline`` leg runs this (as a subprocess with a hard time limit).  It is a reported
baseline, not the product.  Sample (all WARM: each stage runs once untimed first): SAM encoder = 1 windowed + 1 global
block timed and scaled to the architecture's block counts; DINOv2 = 1 of 24 blocks; decoder + post-processing = one
batch of 32 prompts, the reference's own points_per_batch.
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


def _warm(fn, reps=1):
    """Run once untimed (page-in, thread pool spin-up, allocator warm-up), then time ``reps`` runs; returns
    (seconds per run, last result)."""
    out = fn()
    t0 = time.time()
    for _ in range(reps):
        out = fn()
    return (time.time() - t0) / reps, out


def measure(arch="vit_l", n_prompts_full=4096, batch=32, log=None):
    """WARM timings (every stage runs once untimed first) of: patch embed, one windowed and one global encoder block
    (scaled to the architecture's block counts), the neck, one DINOv2 block (x24), and ONE decoder batch of ``batch``
    prompts -- the reference's own points_per_batch (configs/crowdhuman.yaml:48) -- incl. postprocess_masks, PWD-Net
    selection, stability and boxes (scaled to the prompt count).  Stage breakdown as SURVEY.md section 8d asks."""
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle import sam_oracle as so
    say = (lambda *a: print(*a, file=sys.stderr, flush=True)) if log is None else log
    D, depth, heads, gidx = synth.SAM_CONFIGS[arch]
    threads = torch.get_num_threads()
    n_glob = len(gidx)
    specs = [s for s in synth.sam_param_specs(D, 2, heads, (1,)) if not s[0].startswith("image_encoder.blocks.")
             or s[0].startswith("image_encoder.blocks.0.") or s[0].startswith("image_encoder.blocks.1.")]
    sd = synth.make_state_dict(specs, 0)
    image = synth.synthetic_crowd_frame(0, 1024, 150)
    img = torch.from_numpy(image).permute(2, 0, 1).float().contiguous()
    F = torch.nn.functional
    with torch.no_grad():
        x = so.preprocess(img)[None]

        def embed():
            t = F.conv2d(x, sd["image_encoder.patch_embed.proj.weight"], sd["image_encoder.patch_embed.proj.bias"],
                         stride=16).permute(0, 2, 3, 1)
            return t + sd["image_encoder.pos_embed"]

        t_embed, tok = _warm(embed)
        t_win, tok1 = _warm(lambda: so.encoder_block(sd, "image_encoder.blocks.0.", tok, heads, 14))
        t_glob, tok2 = _warm(lambda: so.encoder_block(sd, "image_encoder.blocks.1.", tok1, heads, 0))

        def neck():
            f = tok2.permute(0, 3, 1, 2)
            f = F.conv2d(f, sd["image_encoder.neck.0.weight"])
            f = so.layer_norm_2d(sd, "image_encoder.neck.1", f)
            f = F.conv2d(f, sd["image_encoder.neck.2.weight"], padding=1)
            return so.layer_norm_2d(sd, "image_encoder.neck.3", f)

        t_neck, feat = _warm(neck)
        t_enc = t_embed + t_win * (depth - n_glob) + t_glob * n_glob + t_neck
        say(f"cpu_baseline: encoder win {t_win:.2f}s glob {t_glob:.2f}s -> {t_enc:.1f}s")
        dsd = synth.make_state_dict(synth.dino_param_specs(1024, 1), 1)
        xd = F.interpolate(x, (1022, 1022), mode="bilinear")
        t_d1, _ = _warm(lambda: so.dinov2_forward(dsd, xd, depth=1))
        t_dino = t_d1 * 24.0
        say(f"cpu_baseline: dino {t_dino:.1f}s")
        dino_feats = torch.from_numpy(np.random.RandomState(1).standard_normal((1, 73, 73, 1024)).astype(np.float32))
        pe = so.dense_pe(sd)
        pts = torch.from_numpy(np.random.RandomState(2).randint(0, 1024, size=(batch, 1, 2)).astype(np.float64))

        def decode():
            sparse = so.embed_points(sd, pts, torch.ones(batch, 1, dtype=torch.int))
            return so.mask_decoder(sd, feat, pe, sparse, dino_feats)

        t_dec, (low, iou, cls) = _warm(decode)

        def post():
            masks = so.postprocess_masks(low, image.shape[:2], image.shape[:2])
            s = torch.clamp(iou, 0.) * cls.squeeze(2).sigmoid()
            sel = masks[torch.arange(batch), s.max(-1)[1]]
            po.calculate_stability_score(sel, 0.0, 1.0)
            return po.batched_mask_to_box(sel > 0)

        t_post, _ = _warm(post)
        say(f"cpu_baseline: decoder batch of {batch}: {t_dec:.2f}s + post {t_post:.2f}s")
    n_batches = n_prompts_full / batch
    t_sweep, t_posts = n_batches * t_dec, n_batches * t_post
    total = t_enc + t_dino + t_sweep + t_posts
    return dict(value=1.0 / total, unit="images/s", cores=threads, kind="port",
                stages_s={"encoder": t_enc, "dino": t_dino, "decoder_sweep": t_sweep, "post": t_posts},
                sample=(f"warm timings on {threads} torch threads: SAM {arch} encoder = patch embed {t_embed:.2f}s + "
                        f"{depth - n_glob} x windowed block {t_win:.2f}s + {n_glob} x global block {t_glob:.2f}s + neck "
                        f"{t_neck:.2f}s = {t_enc:.1f}s; DINOv2-L = 24 x one block {t_d1:.2f}s = {t_dino:.1f}s; decoder sweep = "
                        f"{n_batches:.0f} batches of {batch} prompts (the reference's points_per_batch) x {t_dec:.2f}s = "
                        f"{t_sweep:.0f}s; postprocess_masks + selection + stability + boxes = {n_batches:.0f} x {t_post:.2f}s = "
                        f"{t_posts:.0f}s; one block of each kind and one batch were RUN, the rest scaled"))


def measure_e2e(arch="vit_l", grid=8, batch=32):
    """ONE image end to end through the oracle driver (oracle/pipeline_oracle.py::OracleCrowdSAM.generate: full-depth SAM
    encoder + full-depth DINOv2-L + FG prior + dense sweep of a grid x grid prompt grid in batches of ``batch`` + filters + NMS
    + RLE), MEASURED, not scaled: the small prompt count (64 at grid 8) is what keeps it inside a bench run (VERDICT r2
    item 9).  Cold run: nothing is executed twice."""
    from crowdsam_amd import synth
    from oracle import pipeline_oracle as po
    from oracle import sam_oracle as so
    D, depth, heads, gidx = synth.SAM_CONFIGS[arch]
    sd = synth.make_sam_state_dict(arch)
    dsd = synth.make_dino_state_dict()
    cfg = dict(po.DEFAULT_TEST_CFG)
    cfg.update(grid_size=grid, points_per_batch=batch, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"),
               max_prompts=grid * grid, stability_score_thresh=0.25, min_mask_region_area=0)
    image = synth.synthetic_crowd_frame(0, 1024, 150)
    np.random.seed(0)
    o = po.OracleCrowdSAM(sd, (depth, heads, gidx), lambda x: so.dinov2_forward(dsd, x, depth=24), cfg, rng=np.random)
    t0 = time.time()
    with torch.no_grad():
        out = o.generate(image)
    dt = time.time() - t0
    return dict(seconds=dt, prompts=grid * grid, batches=-(-grid * grid // batch), kept=int(len(out["boxes"])),
                images_per_sec=1.0 / dt)


if __name__ == "__main__":
    arch = sys.argv[1] if len(sys.argv) > 1 else "vit_l"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    res = measure(arch, n)
    if "--e2e" in sys.argv:
        try:
            e = measure_e2e(arch)
            res["e2e_measured"] = e
            res["sample"] = ("`value` = EXTRAPOLATION to %d prompts: %s.  `e2e_measured` = ONE image run end to end through the "
                             "oracle driver at %d prompts (%d batches of 32), cold, nothing scaled: %.1f s"
                             % (n, res["sample"], e["prompts"], e["batches"], e["seconds"]))
        except Exception as exc:   # noqa: BLE001
            res["e2e_measured"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    print(json.dumps(res), flush=True)
