"""ViT-H (head_dim 80, generic attention route) encoder-only timing on synthetic weights."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from crowdsam_amd import synth
from crowdsam_amd.encoder import EncoderPlan
dev = torch.device("cuda:0")
D, depth, heads, gidx = synth.SAM_CONFIGS["vit_h"]
specs = [s for s in synth.sam_param_specs(D, depth, heads, gidx) if s[0].startswith("image_encoder.")]
sd = synth.make_state_dict(specs, 5)
plan = EncoderPlan(sd, "image_encoder.", D, depth, heads, gidx, dev)
img = torch.rand(3, 1024, 1024, device=dev) * 255
for _ in range(2):
    plan.forward_static(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    plan.forward_static(img)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"vit_h encoder {ms:.2f} ms/image, {plan.flops() / ms / 1e9:.1f} TF/s required-flops rate")
