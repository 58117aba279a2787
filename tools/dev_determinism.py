"""Developer check: bitwise repeatability of the encoder, DINOv2 and one decoder batch (intermediate tensors)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crowdsam_amd import synth, hip
from crowdsam_amd.decoder import DecoderPlan

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
specs = [s for s in synth.sam_param_specs(128, 4, 2, (1, 3))]
sd = synth.make_state_dict(specs, 0)
plan = DecoderPlan(sd, torch.device("cuda"), 1, B)
torch.manual_seed(0)
feat = torch.randn(4096, 256, device="cuda")
dtok = torch.zeros(5376, 1024, dtype=torch.float16, device="cuda"); dtok[:5329] = torch.randn(5329, 1024, device="cuda").half()
plan.set_image(feat, dtok)
coords = torch.rand(B, 2, device="cuda") * 1023
ref = None
names = ["masks", "iou", "cls", "keysA", "keysB", "t2i_o", "hyper", "stats", "wadj", "pooled", "queries"]
for r in range(6):
    plan.run_batch(coords)
    torch.cuda.synchronize()
    cur = {n: plan.ws[n].clone() for n in names}
    if ref is None:
        ref = cur
        continue
    bad = []
    for n in names:
        a, b = ref[n], cur[n]
        neq = (a.view(-1).view(torch.int16 if a.element_size() == 2 else torch.int32) != b.view(-1).view(torch.int16 if a.element_size() == 2 else torch.int32))
        k = int(neq.sum())
        if k:
            bad.append((n, k, a.numel()))
    print("run", r, "differs:", bad if bad else "none", flush=True)
