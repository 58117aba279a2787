"""Developer: SAM ViT-L encoder and DINOv2-L, ms per image at B images per pass (hipGraph replay)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crowdsam_amd import synth
from crowdsam_amd.encoder import EncoderPlan
from crowdsam_amd.dino import DinoPlan, N_PATCH

dev = torch.device("cuda:0")
D, depth, heads, gidx = synth.SAM_CONFIGS["vit_l"]
enc = EncoderPlan(synth.make_sam_state_dict("vit_l"), "image_encoder.", D, depth, heads, gidx, dev)
dino = DinoPlan(synth.make_dino_state_dict(), dev)
frames = [torch.from_numpy(synth.synthetic_crowd_frame(i, 1024, 100)).permute(2, 0, 1).float().contiguous().to(dev) for i in range(8)]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (1, 2, 3, 4, 6, 8):
    imgs = frames[:B]
    outs = [torch.zeros(5376, 1024, dtype=torch.float16, device=dev) for _ in range(B)]
    te = timeit(lambda: enc.forward_batch_static(imgs))
    td = timeit(lambda: dino.forward_batch_static(imgs, [o[:N_PATCH] for o in outs]))
    print("B=%d  SAM ViT-L %.3f ms/image (%.0f TFLOP/s of 2985.7 GFLOP)   DINOv2-L %.3f ms/image   sum %.3f"
          % (B, te / B, 2985.7e9 / (te / B * 1e-3) / 1e12, td / B, (te + td) / B), flush=True)
