"""Developer: one decoder batch of B prompts -- hipGraph replay time vs the sum of its kernels' durations (gaps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import synth, hip
from crowdsam_amd.decoder import DecoderPlan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
specs = [s for s in synth.sam_param_specs(128, 4, 2, (1, 3))]
sd = synth.make_state_dict(specs, 0)
plan = DecoderPlan(sd, torch.device("cuda"), 1, B)
feat = torch.randn(4096, 256, device="cuda")
dtok = torch.zeros(5376, 1024, dtype=torch.float16, device="cuda"); dtok[:5329] = torch.randn(5329, 1024, device="cuda").half()
plan.set_image(feat, dtok)
coords = torch.rand(B, 2, device="cuda") * 1023
for _ in range(3): plan.run_batch(coords)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): plan.run_batch(coords)
e1.record(); torch.cuda.synchronize()
print(f"B={B}: graph replay {e0.elapsed_time(e1) / n * 1e3:.1f} us per batch")
