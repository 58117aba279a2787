"""GPU parity of the DINOv2 HIP forward against the CPU oracle restatement (depth-2 ViT-L/14 width)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dino_forward_vs_oracle(cuda):
    from crowdsam_amd import synth
    from crowdsam_amd.dino import DinoPlan
    from oracle import sam_oracle as so
    depth = 2
    sd = synth.make_state_dict(synth.dino_param_specs(1024, depth), 1)
    plan = DinoPlan(sd, cuda, depth=depth)
    img = synth.synthetic_crowd_frame(5, 1024, 40)[:768]          # 768 x 1024
    img_t = torch.from_numpy(img).permute(2, 0, 1).float().contiguous()
    y = plan.forward(img_t.to(cuda)).float().cpu()
    with torch.no_grad():
        x = so.preprocess(img_t)[None]
        xd = torch.nn.functional.interpolate(x, (1022, 1022), mode="bilinear")
        ref = so.dinov2_forward(sd, xd, depth=depth)[0]
    err = (y - ref).abs()
    assert err.max().item() < 6e-2 and err.mean().item() < 4e-3, (err.max().item(), err.mean().item())
