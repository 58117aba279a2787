"""Developer: every csam_gemm_f16* launch of one crowded bench frame, grouped by (M, N, K, epilogue): calls per image, HIP-event
time, TFLOP/s.  Eager launches with events (graphs and the side streams off), so the times are the kernels' own."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
import torch
from crowdsam_amd import hip, synth
from crowdsam.model import CrowdSAM
from crowdsam.utils import DEFAULT_TEST_CONFIG

hip.GRAPHS_ENABLED = False
import segment_anything_cs.predictor as _pred
_pred._TWO_STREAMS = False
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
         filter_thresh=float("inf"), max_prompts=4096, box_nms_thresh=1.0, crop_nms_thresh=1.0, pred_iou_thresh=0.889)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(4)]
m.generate(frames[0])
torch.cuda.synchronize()
calls = []


def wrap(name):
    orig = getattr(hip, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        if name == "gemm_f16_batched":
            A, lda, sA, W, ldw, sW, C, ldc, sC, M, N, K, batch = a[:13]
            key = (M, N, K, "batched x%d" % batch)
            fl = 2.0 * M * N * K * batch
        else:
            A, W = a[0], a[1]
            M = k.get("M") or A.shape[0]
            out = k.get("out") if name != "gemm_f16_ln" else a[2]
            ep = []
            if k.get("residual") is not None or (name == "gemm_f16_resmod"):
                ep.append("+res")
            if k.get("act", 0):
                ep.append("act")
            if k.get("stats_in") is not None:
                ep.append("ln-in")
            if k.get("stats_out") is not None:
                ep.append("ln-out")
            dt = "f32" if (out is not None and out.dtype == torch.float32) or k.get("out_dtype") == torch.float32 else "f16"
            key = (M, W.shape[0], A.shape[1], dt + "".join(ep))
            fl = 2.0 * M * W.shape[0] * A.shape[1]
        calls.append((key, e0, e1, fl))
        return r
    setattr(hip, name, f)


for n in ("gemm_f16", "gemm_f16_ln", "gemm_f16_resmod", "gemm_f16_batched"):
    if hasattr(hip, n):
        wrap(n)
N = 3
for f in frames[1:1 + N]:
    m.generate(f)
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for key, e0, e1, fl in calls:
    a = acc[key]
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
    a[2] += fl
tot = sum(a[1] for a in acc.values()) / N
print("GEMM launches of one crowded frame (ViT-L + DINOv2-L + 4096 prompts), %d frames averaged: %.2f ms per image" % (N, tot / 1e3))
print("%7s %6s %6s %-16s %6s %9s %9s %8s" % ("M", "N", "K", "epilogue", "calls", "us/call", "us/image", "TFLOP/s"))
for key, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%7d %6d %6d %-16s %6.1f %9.1f %9.1f %8.0f" % (key[0], key[1], key[2], key[3], a[0] / N, a[1] / a[0], a[1] / N, a[2] / a[1] / 1e6))
