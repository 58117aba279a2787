"""Developer benchmark: SAM ViT-L encoder forward on synthetic weights (ms, TFLOP/s)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import synth
from crowdsam_amd.encoder import EncoderPlan

arch = sys.argv[1] if len(sys.argv) > 1 else "vit_l"
D, depth, heads, gidx = synth.SAM_CONFIGS[arch]
t0 = time.time()
specs = [s for s in synth.sam_param_specs(D, depth, heads, gidx) if s[0].startswith("image_encoder.")]
sd = synth.make_state_dict(specs, 0)
print("weights", time.time() - t0, flush=True)
plan = EncoderPlan(sd, "image_encoder.", D, depth, heads, gidx, torch.device("cuda"))
img = torch.rand(3, 1024, 1024, device="cuda") * 255
for _ in range(3):
    plan.forward(img)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    plan.forward(img)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{arch}: {ms:.3f} ms/image  {plan.flops()/ms/1e9:.1f} TFLOP/s (required flops {plan.flops()/1e9:.1f} G)")
