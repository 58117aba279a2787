"""GPU parity of the DINOv2 HIP forward: depth 2 against the CPU oracle restatement, and FULL depth (24 blocks) against
an independent implementation of the same published architecture -- transformers.models.dinov2 run in the authoring
container on the seeded weights (tests/golden/dino_hf_vitl14.npz, oracle/make_goldens.py::golden_dino_hf).  The
reference's own dinov2/ submodule is empty and unpinned, so DINOv2 stays "parity unpinned" w.r.t. the reference."""
import os

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


def test_dino_full_depth_vs_transformers_golden(cuda):
    from crowdsam_amd import synth
    from crowdsam_amd.dino import DinoPlan
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dino_hf_vitl14.npz"))
    # which pos-embed interpolation form the independent implementation agrees with (recorded by the generator)
    print("oracle vs HF: size= form", g["err_vs_oracle_size"], " +0.1 scale_factor form", g["err_vs_oracle_offset"])
    assert g["err_vs_oracle_size"][0] < 2e-3            # the restatement (size= form) == transformers, fp32 round-off
    sd = synth.make_dino_state_dict()
    img = synth.synthetic_crowd_frame(5, 1024, 40)[:768]
    img_t = torch.from_numpy(img).permute(2, 0, 1).float().contiguous()
    ref = g["sample"]
    scale = np.abs(ref).mean()
    plan = DinoPlan(sd, cuda, depth=24, pos_offset=None)
    y = plan.forward(img_t.to(cuda)).float().cpu().numpy()
    err = np.abs(y[::7, ::8] - ref)
    print("DINOv2-L x24 vs transformers: mean|ref| %.3f max err %.4f mean err %.5f" % (scale, err.max(), err.mean()))
    assert err.mean() < 0.01 * scale and err.max() < 0.12 * scale, (err.mean(), err.max(), scale)
